// Host check of msm_kernels.h's heavy_slice (the partition k_heavy_combine and k_heavy_finish both rely on): for every bucket span
// tf .. tl the non-empty slices are disjoint, consecutive, cover every chunk exactly once, and their first chunks are distinct slots.
#include <cstdio>
#include "../../go-snark-study_amd/csrc/msm_kernels.h"
using namespace gs;
int main() {
  long checked = 0;
  for (uint32_t tf : {0u, 1u, 7u, 1000u, 123456u})
    for (uint32_t n = kHeavySpan + 2; n < 9000; n += (n < 400 ? 1 : 37)) {
      const uint32_t tl = tf + n - 1;
      uint32_t next = tf;
      for (uint32_t s = 0; s < (uint32_t)kHeavySlices; ++s) {
        uint32_t c0, c1;
        heavy_slice(tf, tl, s, c0, c1);
        if (c0 > tl) continue;                                   // empty slice: nothing may follow it either
        if (c0 != next || c1 <= c0 || c1 > tl + 1) { printf("FAIL tf=%u n=%u slice %u: [%u, %u) after %u\n", tf, n, s, c0, c1, next); return 1; }
        next = c1;
      }
      if (next != tl + 1) { printf("FAIL tf=%u n=%u: slices end at %u, bucket at %u\n", tf, n, next, tl + 1); return 1; }
      ++checked;
    }
  printf("OK %ld spans\n", checked);
  return 0;
}
