// Where do the waves of a "3 workgroups per CU" launch really run, and when?  (Round 6: every "cycles per instruction per SIMD" of
// tools/ubench_issue.hip / ubench_mulmod.hip assumes 3 waves on every SIMD for the whole kernel; the kernels' event time is 1.4-1.5 x
// the waves' own s_memrealtime span, so either the waves do not run at the same time or they are not spread 3 per SIMD.)
// Every wave records s_memrealtime at its first and last instruction, HW_REG_HW_ID (wave / SIMD / CU / SH / SE) and HW_REG_XCC_ID; the
// host prints the histogram of waves per SIMD, the number of SIMDs used, and the timeline (start and end percentiles in microseconds
// from the first wave's start) for the multiply-add loop of ubench_issue (mode 0) at 1, 2, 3 and 6 workgroups per CU.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/ubench_placement.hip -o tools/ubench_placement
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <map>
#include <string>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
constexpr int ITERS = 1000;
struct Rec { unsigned long long r0, r1, t0, t1; uint32_t hw, xcc; };

#define MAD8 asm volatile( \
    "v_mad_u64_u32 %0, vcc, %8, %9, %0\n v_mad_u64_u32 %1, vcc, %8, %9, %1\n v_mad_u64_u32 %2, vcc, %8, %9, %2\n v_mad_u64_u32 %3, vcc, %8, %9, %3\n" \
    "v_mad_u64_u32 %4, vcc, %8, %9, %4\n v_mad_u64_u32 %5, vcc, %8, %9, %5\n v_mad_u64_u32 %6, vcc, %8, %9, %6\n v_mad_u64_u32 %7, vcc, %8, %9, %7\n" \
    : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(x), "v"(y) : "vcc")

__global__ void __launch_bounds__(256, 3) k_mad(uint32_t* out, Rec* rec, uint32_t seed) {
  uint32_t x = seed + threadIdx.x, y = seed * 3u + blockIdx.x;
  uint64_t a0 = x, a1 = y, a2 = 3, a3 = 4, a4 = 5, a5 = 6, a6 = 7, a7 = 8;
  const unsigned long long r0 = __builtin_amdgcn_s_memrealtime(), t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < ITERS; ++it) {
#pragma unroll
    for (int r = 0; r < 8; ++r) MAD8;
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  if (threadIdx.x % 64 == 0) {
    Rec& q = rec[(blockIdx.x * blockDim.x + threadIdx.x) / 64];
    q.r0 = r0; q.r1 = r1; q.t0 = t0; q.t1 = t1;
    q.hw = __builtin_amdgcn_s_getreg(4 | (31 << 11));          // HW_REG_HW_ID
    q.xcc = __builtin_amdgcn_s_getreg(20 | (31 << 11));        // HW_REG_XCC_ID
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = (uint32_t)(a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7);
}

// `ubench_placement masked`: the same launch (3 workgroups per CU of the mask) on CU-masked streams of 8 / 32 / 128 CUs, mask bits
// spread over the whole mask ("spread", what ubench_issue's argv[3] builds) or the first bits ("first"): does the mask restrict the
// placement, i.e. do three waves really share every SIMD of the masked CUs?
int main(int argc, char** argv) {
  CK(hipSetDevice(0));
  hipDeviceProp_t prop;
  CK(hipGetDeviceProperties(&prop, 0));
  const int all_cus = prop.multiProcessorCount;
  const bool masked = argc >= 2 && std::string(argv[1]) == "masked";
  struct Case { int cus, per_cu; int layout; };             // layout 0: no mask, 1: spread bits, 2: first bits
  std::vector<Case> cases;
  if (!masked) for (int per_cu : {1, 2, 3, 4, 6}) cases.push_back({all_cus, per_cu, 0});
  else for (int layout : {1, 2}) for (int c : {8, 32, 128, 256}) cases.push_back({c, 3, layout});
  for (const Case& cs_ : cases) {
    const int cus = cs_.cus, per_cu = cs_.per_cu;
    hipStream_t stream = nullptr;
    if (cs_.layout) {
      std::vector<uint32_t> mask((all_cus + 31) / 32, 0u);
      for (int i = 0; i < cus; ++i) {
        const int bit = cs_.layout == 1 ? (int)((long long)i * all_cus / cus) : i;
        mask[bit / 32] |= 1u << (bit % 32);
      }
      CK(hipExtStreamCreateWithCUMask(&stream, (uint32_t)mask.size(), mask.data()));
      std::vector<uint32_t> back(mask.size(), 0u);
      hipError_t ge = hipExtStreamGetCUMask(stream, (uint32_t)back.size(), back.data());
      int set = 0; for (uint32_t w : back) set += __builtin_popcount(w);
      printf("CU mask of %d CUs, %s bits; hipExtStreamGetCUMask: %s, %d bits set\n", cus, cs_.layout == 1 ? "spread" : "first", hipGetErrorString(ge), set);
    }
    const int blocks = cus * per_cu, waves = blocks * 4;
    uint32_t* dout; Rec* drec;
    CK(hipMalloc(&dout, (size_t)blocks * 256 * 4)); CK(hipMalloc(&drec, (size_t)waves * sizeof(Rec)));
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    float ms = 0;
    for (int rep = 0; rep < 3; ++rep) {
      CK(hipEventRecord(a, stream));
      hipLaunchKernelGGL(k_mad, dim3(blocks), dim3(256), 0, stream, dout, drec, 777u + rep);
      CK(hipEventRecord(b, stream)); CK(hipEventSynchronize(b)); CK(hipEventElapsedTime(&ms, a, b));
    }
    std::vector<Rec> h(waves);
    CK(hipMemcpy(h.data(), drec, waves * sizeof(Rec), hipMemcpyDeviceToHost));
    unsigned long long first = ~0ull, last = 0;
    std::map<uint64_t, int> per_simd, per_cu_count;
    std::vector<double> starts, ends, spans, cyc;
    for (auto& r : h) { first = std::min(first, r.r0); last = std::max(last, r.r1); }
    for (auto& r : h) {
      const uint32_t simd = (r.hw >> 4) & 3, cu = (r.hw >> 8) & 15, sh = (r.hw >> 12) & 1, se = (r.hw >> 13) & 7, xcc = r.xcc & 15;
      const uint64_t cu_key = ((uint64_t)xcc << 16) | (se << 8) | (sh << 4) | cu;
      per_simd[(cu_key << 4) | simd] += 1; per_cu_count[cu_key] += 1;
      starts.push_back((r.r0 - first) * 0.01); ends.push_back((r.r1 - first) * 0.01); spans.push_back((r.r1 - r.r0) * 0.01); cyc.push_back((double)(r.t1 - r.t0));
    }
    std::map<int, int> hist;
    for (auto& kv : per_simd) hist[kv.second] += 1;
    auto pct = [](std::vector<double> v, double q) { std::sort(v.begin(), v.end()); return v[std::min(v.size() - 1, (size_t)(q * v.size()))]; };
    double cs = 0; for (double v : cyc) cs += v;
    printf("%d workgroups per CU (%d waves): kernel %.1f us (events); waves on %zu CUs / %zu SIMDs; waves per SIMD histogram:", per_cu, waves, ms * 1e3, per_cu_count.size(), per_simd.size());
    for (auto& kv : hist) printf(" %dx%d", kv.first, kv.second);
    printf("\n    starts us p0 %.1f p50 %.1f p90 %.1f p100 %.1f | ends us p0 %.1f p50 %.1f p100 %.1f | wave span us p5 %.1f p50 %.1f p95 %.1f | s_memtime cycles per wave mean %.0f -> %.3f cycles per mad per wave (longest wave %.3f, shortest %.3f)\n",
           pct(starts, 0), pct(starts, 0.5), pct(starts, 0.9), pct(starts, 1.0), pct(ends, 0), pct(ends, 0.5), pct(ends, 1.0), pct(spans, 0.05), pct(spans, 0.5), pct(spans, 0.95),
           cs / waves, cs / waves / (64.0 * ITERS), pct(cyc, 1.0) / (64.0 * ITERS), pct(cyc, 0) / (64.0 * ITERS));
    CK(hipFree(dout)); CK(hipFree(drec));
    if (stream) CK(hipStreamDestroy(stream));
  }
  return 0;
}
