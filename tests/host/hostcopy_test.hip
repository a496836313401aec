// Stress test of the host-side copy pool behind the uploads from pageable caller memory (go-snark-study_amd/csrc/hostcopy.h):
// thousands of jobs of varying size and alignment, every byte checked.  Host code only (no device call is made).
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <vector>

#include "../../go-snark-study_amd/csrc/hostcopy.h"

int main(int argc, char** argv) {
  const int iterations = argc > 1 ? atoi(argv[1]) : 6000;
  std::vector<uint8_t> src(9u << 20), dst(9u << 20);
  for (size_t i = 0; i < src.size(); ++i) src[i] = (uint8_t)((i * 2654435761u) >> 13);
  uint64_t x = 88172645463325252ull;
  for (int it = 0; it < iterations; ++it) {
    x ^= x << 13; x ^= x >> 7; x ^= x << 17;
    const size_t n = (x % 3 == 0) ? (256u << 10) + x % 5000 : (size_t)(x % (8u << 20)) + 1;     // around the pool's threshold and up to 8 MiB
    const size_t off = (size_t)((x >> 40) % (1u << 20));
    memset(dst.data() + off, 0, n);
    gs::HostCopyPool::get().copy(dst.data() + off, src.data() + off, n);
    if (memcmp(dst.data() + off, src.data() + off, n) != 0) { printf("MISMATCH iteration %d n=%zu off=%zu\n", it, n, off); return 1; }
  }
  printf("OK %d\n", iterations);
  return 0;
}
