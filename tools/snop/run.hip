// Loads the code objects tools/snop/build.py made and times them: per variant and occupancy the time of k_two_chains / k_three_chains,
// whether the results equal the as-compiled kernel's, and cycles per product per SIMD at the nominal 2.4 GHz.
//   tools/snop/run [dir]       (on the GPU box; dir defaults to the directory of the executable)
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <string>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
constexpr int NL = 9, ITERS = 300;

int main(int argc, char** argv) {
  std::string dir = argc > 1 ? argv[1] : std::string(argv[0]).substr(0, std::string(argv[0]).find_last_of('/'));
  hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
  std::vector<uint32_t> h(4096 * NL);
  uint64_t s = 0x9E3779B97F4A7C15ull;
  for (auto& v : h) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; v = (uint32_t)s; }
  uint32_t *din = nullptr, *dout = nullptr;
  const size_t maxthreads = (size_t)prop.multiProcessorCount * 3 * 256;
  CK(hipMalloc(&din, h.size() * 4)); CK(hipMalloc(&dout, maxthreads * 6 * NL * 4));
  CK(hipMemcpy(din, h.data(), h.size() * 4, hipMemcpyHostToDevice));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  printf("%s, %d CUs; every kernel: 6 sequences x %d Montgomery products per thread (162 v_mad_u64_u32 each)\n", prop.gcnArchName, prop.multiProcessorCount, ITERS);
  for (int waves : {3, 2}) {
    const int blocks = prop.multiProcessorCount * waves;            // one 256-thread block per CU per wave slot: `waves` waves on every SIMD
    const size_t nthreads = (size_t)blocks * 256;
    std::vector<uint32_t> ref[2], got(nthreads * 6 * NL);
    for (const char* variant : {"as_compiled", "nops_removed", "nops_added", "as_compiled"}) {
      hipModule_t mod;
      const std::string path = dir + "/w" + std::to_string(waves) + "_" + variant + ".hsaco";
      if (hipModuleLoad(&mod, path.c_str()) != hipSuccess) { printf("cannot load %s\n", path.c_str()); return 2; }
      int ki = 0;
      for (const char* kname : {"k_two_chains", "k_three_chains"}) {
        hipFunction_t fn; CK(hipModuleGetFunction(&fn, mod, kname));
        void* args[] = {&din, &dout};
        float best = 1e9f;
        for (int rep = 0; rep < 4; ++rep) {
          CK(hipEventRecord(e0));
          CK(hipModuleLaunchKernel(fn, blocks, 1, 1, 256, 1, 1, 0, nullptr, args, nullptr));
          CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
          float ms; CK(hipEventElapsedTime(&ms, e0, e1));
          if (rep > 0 && ms < best) best = ms;
        }
        CK(hipMemcpy(got.data(), dout, got.size() * 4, hipMemcpyDeviceToHost));
        if (ref[ki].empty()) ref[ki] = got;
        const double prods = (double)nthreads * 6 * ITERS;
        printf("%d waves/SIMD  %-13s %-15s %8.3f ms  %7.1f cycles per product per SIMD @2.4 GHz   %s\n", waves, variant, kname, best,
               best * 1e-3 * 2.4e9 * prop.multiProcessorCount * 4 / (prods / 64), ref[ki] == got ? "results identical" : "RESULTS DIFFER");
        ++ki;
      }
      CK(hipModuleUnload(mod));
    }
  }
  return 0;
}
