/* What the LAST METRE of the boundary costs: go/gosnarkhip/pack.go turns the reference's []*big.Int -- a slice of
 * pointers to {neg bool; abs []Word} headers whose word arrays live elsewhere on the heap -- into n x 4 limbs.  No Go toolchain exists in
 * the image, so this is the same memory walk in C: n heap objects {flag, words*, len, cap} allocated in shuffled order, each with its own
 * 4-word array, packed by 1 / 2 / 4 / 8 threads over contiguous chunks exactly as ScalarsInto does (sign check, length check, copy of
 * <= 4 words, zero fill).  It prices the pointer chasing, which is what dominates; Go adds bounds checks and its write barrier-free copy.
 *   gcc -O2 -pthread tools/pack_cost.c -o /tmp/pack_cost && /tmp/pack_cost 20 */
#define _POSIX_C_SOURCE 200809L
#include <pthread.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

typedef struct { int neg; uint64_t* abs; size_t len, cap; } bigint;
typedef struct { bigint** vals; uint64_t* dst; size_t lo, hi; int bad; } job;

static double now_ms(void) { struct timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return ts.tv_sec * 1e3 + ts.tv_nsec * 1e-6; }
static void* pack(void* arg) {
  job* j = (job*)arg;
  for (size_t i = j->lo; i < j->hi; ++i) {
    const bigint* v = j->vals[i];
    if (!v || v->neg || v->len > 4) { j->bad = 1; return NULL; }
    uint64_t* d = j->dst + 4 * i;
    d[0] = d[1] = d[2] = d[3] = 0;
    for (size_t k = 0; k < v->len; ++k) d[k] = v->abs[k];
  }
  return NULL;
}

int main(int argc, char** argv) {
  const size_t n = (size_t)1 << (argc > 1 ? atoi(argv[1]) : 20);
  bigint** vals = (bigint**)malloc(n * sizeof *vals);
  size_t* order = (size_t*)malloc(n * sizeof *order);
  uint64_t* dst = (uint64_t*)malloc(n * 32);
  uint64_t s = 88172645463325252ull;
  for (size_t i = 0; i < n; ++i) order[i] = i;
  for (size_t i = n - 1; i > 0; --i) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; size_t k = s % (i + 1), t = order[i]; order[i] = order[k]; order[k] = t; }
  for (size_t i = 0; i < n; ++i) {                    /* allocation order != slice order: what a witness computed gate by gate looks like */
    bigint* b = (bigint*)malloc(sizeof *b);
    b->neg = 0; b->len = b->cap = 4; b->abs = (uint64_t*)malloc(32);
    for (int k = 0; k < 4; ++k) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; b->abs[k] = s; }
    vals[order[i]] = b;
  }
  memset(dst, 1, n * 32);
  for (int threads = 1; threads <= 8; threads *= 2) {
    double best = 1e30;
    for (int rep = 0; rep < 5; ++rep) {
      pthread_t th[8]; job jb[8];
      const double t0 = now_ms();
      for (int t = 0; t < threads; ++t) { jb[t] = (job){vals, dst, n * t / threads, n * (t + 1) / threads, 0}; pthread_create(&th[t], NULL, pack, &jb[t]); }
      for (int t = 0; t < threads; ++t) pthread_join(th[t], NULL);
      const double dt = now_ms() - t0;
      if (dt < best) best = dt;
    }
    printf("pack %zu heap big integers -> limbs, %d thread(s): %.2f ms (%.1f ns per element); x3 for w + px of a 2^20 proof: %.1f ms\n", n, threads, best, best * 1e6 / (double)n, 3 * best);
  }
  /* the same with the values already in one flat array (what a limb-native caller hands to the *Limbs entry points): nothing to do */
  const double t0 = now_ms();
  uint64_t* flat = (uint64_t*)malloc(n * 32);
  memcpy(flat, dst, n * 32);
  printf("for comparison, one memcpy of the packed %zu MiB: %.2f ms\n", n * 32 >> 20, now_ms() - t0);
  return 0;
}
