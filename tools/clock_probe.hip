// The shader clock at 20 us resolution WHILE the library's streams run (round 6; dev tool, not part of the product).
// hwmon's sclk is a 49 Hz average and rocprofv3's GRBM_GUI_ACTIVE / duration needs a PMC pass that runs the kernels one after another;
// this is the direct reading: ONE wave on its own high-priority stream samples s_memtime (ticks at sclk) against s_memrealtime (the
// constant 100 MHz counter) every `interval_us` for `samples` intervals, beside whatever the process runs on its other streams
// (a probe wave needs 8 VGPRs: it fits beside three 160-register accumulation waves on a SIMD).  tools/clock_timeline.py drives it.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared tools/clock_probe.hip -o tools/libclock_probe.so
#include <hip/hip_runtime.h>
#include <cstdint>
#include <vector>

__global__ void __launch_bounds__(64) k_clock_probe(unsigned long long* out, uint32_t samples, uint32_t interval_ticks) {
  if (threadIdx.x != 0) return;
  unsigned long long next = __builtin_amdgcn_s_memrealtime() + interval_ticks;
  for (uint32_t i = 0; i < samples; ++i) {
    unsigned long long r;
    do { __builtin_amdgcn_s_sleep(8); r = __builtin_amdgcn_s_memrealtime(); } while (r < next);
    out[2 * i] = r;
    out[2 * i + 1] = __builtin_amdgcn_s_memtime();
    next += interval_ticks;
  }
}

static hipStream_t g_stream = nullptr;
static unsigned long long* g_dev = nullptr;
static uint32_t g_samples = 0;

// start: enqueue the probe (returns at once); finish: wait for it and copy the (realtime, sclk ticks) pairs out
extern "C" int clock_probe_start(uint32_t samples, uint32_t interval_us) {
  if (!g_stream) {
    int least = 0, greatest = 0;
    if (hipDeviceGetStreamPriorityRange(&least, &greatest) != hipSuccess) return -1;
    if (hipStreamCreateWithPriority(&g_stream, hipStreamNonBlocking, greatest) != hipSuccess) return -2;
  }
  if (g_dev) { (void)hipFree(g_dev); g_dev = nullptr; }
  if (hipMalloc(&g_dev, (size_t)samples * 16) != hipSuccess) return -3;
  g_samples = samples;
  hipLaunchKernelGGL(k_clock_probe, dim3(1), dim3(64), 0, g_stream, g_dev, samples, interval_us * 100u);
  return hipGetLastError() == hipSuccess ? 0 : -4;
}
extern "C" int clock_probe_finish(unsigned long long* out_pairs) {
  if (!g_stream || !g_dev) return -1;
  if (hipStreamSynchronize(g_stream) != hipSuccess) return -2;
  if (hipMemcpy(out_pairs, g_dev, (size_t)g_samples * 16, hipMemcpyDeviceToHost) != hipSuccess) return -3;
  (void)hipFree(g_dev); g_dev = nullptr;
  return 0;
}
