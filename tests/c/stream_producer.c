/* The ingest ceiling of the C ABI (VERDICT r5 next #1 iii, weak #10): what a Go caller -- goroutines on OS threads, no Python, no torch --
 * can reach.  A plain-C process builds the sqchain(n) circuit of SURVEY 8d, runs the device trusted setup (gs_groth16_setup), computes
 * `nwit` DISTINCT satisfying witnesses in host memory with its own Fr arithmetic, and then streams proofs the way
 * go/groth16hip.Prover does: P producer threads call gs_groth16_prove_witness_host_begin (route 0: w alone, against the resident sparse
 * R1CS) or gs_groth16_prove_host_begin (route 1: w and px) with another witness every time, at most three tickets in flight on the
 * device (one producer keeps three, two producers one and two; a _begin beyond the device's three answers GS_ERR_BUSY: the producer
 * collects its own oldest ticket, or yields), and collect with
 * gs_groth16_prove_end.  Every collected proof is compared byte for byte with the proof of ITS witness from the first lap, where each
 * one was made by the blocking entry point and checked by gs_groth16_verify for its own public input (and rejected for its neighbour's).
 *
 *   stream_producer <log2n> <nwit> <seconds> [policy = 1]
 * prints, per (route, producers in {1, 2}): proofs, ms per proof, constraints/s, and how long _begin / _end calls took (mean / max):
 * _begin holds the device context's lock while it stages the witness (csrc/hostcopy.h), so with two producers the other one's calls wait
 * for it (the maxima show by how much); _end waits for the device OUTSIDE the lock (runtime.h, wait_ticket_unlocked).  Ends with OK. */
#define _POSIX_C_SOURCE 200809L
#include <pthread.h>
#include <sched.h>
#include <time.h>

#include "instance.h"

typedef unsigned __int128 u128;
static const uint64_t FR[4] = {0x43e1f593f0000001ull, 0x2833e84879b97091ull, 0xb85045b68181585dull, 0x30644e72e131a029ull};
static const uint64_t FR_INV = 0xc2e1f593efffffffull;                                   /* -r^-1 mod 2^64 */
static const uint64_t FR_R2[4] = {0x1bb8e645ae216da7ull, 0x53fe3ab1e35c59e3ull, 0x8c49833d53bb8085ull, 0x0216d0b17f4e44a5ull};   /* 2^512 mod r */

static int geq(const uint64_t a[4], const uint64_t b[4]) {
  for (int i = 3; i >= 0; --i) if (a[i] != b[i]) return a[i] > b[i];
  return 1;
}
static void sub_r(uint64_t a[4]) {
  u128 br = 0;
  for (int i = 0; i < 4; ++i) { u128 d = (u128)a[i] - FR[i] - br; a[i] = (uint64_t)d; br = (d >> 64) & 1; }
}
/* Montgomery product a b / 2^256 mod r (CIOS), inputs < r */
static void mont_mul(uint64_t out[4], const uint64_t a[4], const uint64_t b[4]) {
  uint64_t t[6] = {0, 0, 0, 0, 0, 0};
  for (int i = 0; i < 4; ++i) {
    u128 c = 0;
    for (int j = 0; j < 4; ++j) { c += (u128)a[j] * b[i] + t[j]; t[j] = (uint64_t)c; c >>= 64; }
    c += t[4]; t[4] = (uint64_t)c; t[5] = (uint64_t)(c >> 64);
    const uint64_t m = t[0] * FR_INV;
    c = ((u128)m * FR[0] + t[0]) >> 64;
    for (int j = 1; j < 4; ++j) { c += (u128)m * FR[j] + t[j]; t[j - 1] = (uint64_t)c; c >>= 64; }
    c += t[4]; t[3] = (uint64_t)c; t[4] = t[5] + (uint64_t)(c >> 64);
  }
  if (t[4] || geq(t, FR)) sub_r(t);
  memcpy(out, t, 32);
}
static void add_mod(uint64_t a[4], const uint64_t b[4]) {
  u128 c = 0;
  for (int i = 0; i < 4; ++i) { c += (u128)a[i] + b[i]; a[i] = (uint64_t)c; c >>= 64; }
  if (c || geq(a, FR)) sub_r(a);
}
/* the satisfying assignment of sqchain(n, x): [one, s_1 = x, s_{k+1} = s_k^2 + k] (go-snark-study_amd/synth.py, sqchain_witness) */
static void sqchain_witness(uint64_t* w, size_t n, const uint64_t x[4]) {
  const uint64_t one[4] = {1, 0, 0, 0};
  uint64_t s[4], k[4] = {0, 0, 0, 0}, one_m[4];
  mont_mul(one_m, one, FR_R2);
  mont_mul(s, x, FR_R2);
  memcpy(w, one, 32);
  memcpy(w + 4, x, 32);
  for (size_t i = 1; i < n; ++i) {
    add_mod(k, one_m);                        /* k = i in Montgomery form */
    mont_mul(s, s, s);
    add_mod(s, k);
    mont_mul(w + 4 * (i + 1), s, one);
  }
}

static double now_ms(void) {
  struct timespec ts;
  clock_gettime(CLOCK_MONOTONIC, &ts);
  return ts.tv_sec * 1e3 + ts.tv_nsec * 1e-6;
}
static uint64_t splitmix(uint64_t* s) {
  uint64_t z = (*s += 0x9e3779b97f4a7c15ull);
  z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull;
  z = (z ^ (z >> 27)) * 0x94d049bb133111ebull;
  return z ^ (z >> 31);
}
static void field_elem(uint64_t out[4], uint64_t* seed) {
  do { for (int i = 0; i < 4; ++i) out[i] = splitmix(seed); out[3] &= 0x3fffffffffffffffull; } while (geq(out, FR));
}

typedef struct {
  gs_handle key, r1cs;
  size_t m, npx, nwit;
  int route, first, stride, cap;         /* this producer proves witnesses first, first + stride, ...; cap: tickets it keeps in flight */
  uint64_t **w, **px, *rs, *want;        /* want: 32 words + 3 flags per witness */
  double until;
  long done, begins, ends, mismatches;
  double begin_ms, begin_max, end_ms, end_max;
  int status;
} producer;

static void* produce(void* arg) {
  producer* p = (producer*)arg;
  int dev = 0;
  uint64_t fifo_t[3], got[32];
  size_t fifo_k[3];
  int depth = 0, inf[3];
  size_t k = (size_t)p->first;
  if (gs_set_device(dev) != GS_OK) { p->status = 1; return NULL; }
  for (;;) {
    const int more = now_ms() < p->until;
    if (!more && depth == 0) break;
    int began = 0;
    if (more && depth < p->cap) {
      uint64_t t = 0;
      const double t0 = now_ms();
      const int rc = p->route == 0 ? gs_groth16_prove_witness_host_begin(p->key, p->r1cs, p->w[k], p->m, p->rs, p->rs + 4, &t)
                                   : gs_groth16_prove_host_begin(p->key, p->w[k], p->m, p->px[k], p->npx, p->rs, p->rs + 4, &t);
      const double dt = now_ms() - t0;
      if (rc == GS_OK) {
        p->begins += 1; p->begin_ms += dt; if (dt > p->begin_max) p->begin_max = dt;
        fifo_t[depth] = t; fifo_k[depth] = k; depth += 1; began = 1;
        k += (size_t)p->stride; if (k >= p->nwit) k = (size_t)p->first;
      } else if (rc != GS_ERR_BUSY) { printf("FAIL _begin: %d %s\n", rc, gs_last_error()); p->status = 2; return NULL; }
    }
    if (began && depth < p->cap && more) continue;        /* fill the pipeline before collecting */
    if (depth == 0) { sched_yield(); continue; }           /* all three slots belong to the other producer */
    const double t0 = now_ms();
    const int rc = gs_groth16_prove_end(fifo_t[0], got, inf);
    const double dt = now_ms() - t0;
    if (rc != GS_OK) { printf("FAIL _end: %d %s\n", rc, gs_last_error()); p->status = 3; return NULL; }
    p->ends += 1; p->end_ms += dt; if (dt > p->end_max) p->end_max = dt;
    const uint64_t* want = p->want + 35 * fifo_k[0];
    if (memcmp(got, want, 256) != 0 || (uint64_t)inf[0] != want[32] || (uint64_t)inf[1] != want[33] || (uint64_t)inf[2] != want[34]) p->mismatches += 1;
    p->done += 1;
    depth -= 1;
    for (int i = 0; i < depth; ++i) { fifo_t[i] = fifo_t[i + 1]; fifo_k[i] = fifo_k[i + 1]; }
  }
  return NULL;
}

int main(int argc, char** argv) {
  if (argc < 4) { printf("usage: stream_producer <log2n> <nwit> <seconds> [table policy 0 auto | 1 always | 2 never]\n"); return 9; }
  const size_t n = (size_t)1 << atoi(argv[1]), m = n + 1, nwit = (size_t)atoi(argv[2]), npx = 2 * n - 1;
  const double seconds = atof(argv[3]);
  const int policy = argc > 4 ? atoi(argv[4]) : 1;
  if (n < 4 || nwit < 2 || nwit > 64) return 9;
  int dev = 0;
  CHECK(gs_init(&dev, 1));
  CHECK(gs_set_table_policy(policy));
  /* sqchain(n): A = B: row k -> s_{k+1} (last row: one); C: row k -> {one: -(k+1), s_{k+2}: 1} (last row: {one: 1}) */
  uint32_t* rp = (uint32_t*)malloc((n + 1) * 4), *acol = (uint32_t*)malloc(n * 4), *crp = (uint32_t*)malloc((n + 1) * 4), *ccol = (uint32_t*)malloc(2 * n * 4);
  uint64_t* aval = (uint64_t*)calloc(n * 4, 8), *cval = (uint64_t*)calloc(2 * n * 4, 8);
  if (!rp || !acol || !crp || !ccol || !aval || !cval) return 8;
  for (size_t k = 0; k < n; ++k) { rp[k] = (uint32_t)k; acol[k] = (uint32_t)(k + 1 < n ? k + 1 : 0); aval[4 * k] = 1; }
  rp[n] = (uint32_t)n; acol[n - 1] = 0;
  for (size_t k = 0; k + 1 < n; ++k) {
    crp[k] = (uint32_t)(2 * k);
    ccol[2 * k] = 0; ccol[2 * k + 1] = (uint32_t)(k + 2);
    uint64_t* v = cval + 8 * k;                      /* r - (k + 1) */
    const uint64_t kk = (uint64_t)(k + 1);
    u128 br = 0;
    for (int i = 0; i < 4; ++i) { u128 d = (u128)FR[i] - (i == 0 ? kk : 0) - br; v[i] = (uint64_t)d; br = (d >> 64) & 1; }
    v[4] = 1;
  }
  crp[n - 1] = (uint32_t)(2 * (n - 1)); crp[n] = (uint32_t)(2 * (n - 1) + 1);
  ccol[2 * (n - 1)] = 0; cval[8 * (n - 1)] = 1;
  uint64_t seed = 0x5EED0006ull, toxic[20], rs[8], vk[12 + 72 + 24];
  for (int i = 0; i < 5; ++i) field_elem(toxic + 4 * i, &seed);
  field_elem(rs, &seed); field_elem(rs + 4, &seed);
  gs_handle key, r1cs;
  double t0 = now_ms();
  CHECK(gs_groth16_setup(n, m, 1, rp, acol, aval, rp, acol, aval, crp, ccol, cval, toxic, &key, vk));
  CHECK(gs_r1cs_upload(n, m, rp, acol, aval, rp, acol, aval, crp, ccol, cval, &r1cs));
  printf("sqchain n = %zu: device trusted setup + R1CS upload %.0f ms\n", n, now_ms() - t0);
  /* the witnesses (host arithmetic) and their px (device: gs_r1cs_px, downloaded -- route 1 hands both to every call) */
  uint64_t** w = (uint64_t**)malloc(nwit * sizeof *w), **px = (uint64_t**)malloc(nwit * sizeof *px), *xs = (uint64_t*)malloc(nwit * 32);
  uint64_t* want = (uint64_t*)calloc(nwit * 35, 8);
  if (!w || !px || !xs || !want) return 8;
  t0 = now_ms();
  for (size_t k = 0; k < nwit; ++k) {
    w[k] = (uint64_t*)malloc(m * 32); px[k] = (uint64_t*)malloc(npx * 32);
    if (!w[k] || !px[k]) return 8;
    field_elem(xs + 4 * k, &seed);
    sqchain_witness(w[k], n, xs + 4 * k);
    gs_handle hw, hpx = 0;
    CHECK(gs_scalars_upload(w[k], m, &hw));
    CHECK(gs_r1cs_px(r1cs, hw, &hpx));
    CHECK(gs_scalars_download(hpx, px[k], npx));
    CHECK(gs_free(hw)); CHECK(gs_free(hpx));
  }
  printf("%zu witnesses + their px: %.0f ms\n", nwit, now_ms() - t0);
  if (policy == 1) CHECK(gs_build_tables(key, 0));
  /* first lap: the blocking entry point per witness, the verifier per proof, and both routes agree */
  for (size_t k = 0; k < nwit; ++k) {
    uint64_t* p = want + 35 * k, other[32], jac[48];
    int inf[3], inf2[3], ok = 0, ok_wrong = 1;
    CHECK(gs_groth16_prove_witness_host(key, r1cs, w[k], m, rs, rs + 4, p, inf));
    for (int i = 0; i < 3; ++i) p[32 + i] = (uint64_t)inf[i];
    CHECK(gs_groth16_prove(key, w[k], m, px[k], npx, rs, rs + 4, other, inf2));
    if (memcmp(p, other, 256) != 0 || memcmp(inf, inf2, sizeof inf) != 0) { printf("FAIL: witness route and px route differ for witness %zu\n", k); return 2; }
    proof_to_jacobian(p, inf, jac);
    CHECK(gs_groth16_verify(vk, vk + 12, vk + 36, vk + 60, vk + 84, 2, xs + 4 * k, 1, jac, jac + 12, jac + 36, &ok));
    CHECK(gs_groth16_verify(vk, vk + 12, vk + 36, vk + 60, vk + 84, 2, xs + 4 * ((k + 1) % nwit), 1, jac, jac + 12, jac + 36, &ok_wrong));
    if (!ok || ok_wrong) { printf("FAIL: verifier on witness %zu: own input %d, neighbour's %d\n", k, ok, ok_wrong); return 3; }
  }
  printf("first lap: %zu proofs (blocking, both routes equal) accepted by gs_groth16_verify for their own public input only\n", nwit);
  uint64_t a0, f0, a1, f1;
  for (int route = 0; route < 2; ++route) {
    for (int P = 1; P <= 2; ++P) {
      producer pr[2];
      pthread_t th[2];
      for (int warm = 1; warm >= 0; --warm) {                       /* a short warm-up pass (slot buffers, clocks), then the timed one */
        memset(pr, 0, sizeof pr);
        const double start = now_ms();
        for (int j = 0; j < P; ++j) {
          pr[j].key = key; pr[j].r1cs = r1cs; pr[j].m = m; pr[j].npx = npx; pr[j].nwit = nwit; pr[j].route = route;
          pr[j].first = j; pr[j].stride = P; pr[j].cap = (3 + j) / P;      /* the device has three slots: 3 | 1 + 2 */ pr[j].w = w; pr[j].px = px; pr[j].rs = rs; pr[j].want = want;
          pr[j].until = start + (warm ? (seconds < 1 ? seconds : 1.0) : seconds) * 1e3;
        }
        if (!warm) CHECK(gs_alloc_counters(&a0, &f0));
        for (int j = 0; j < P; ++j) if (pthread_create(&th[j], NULL, produce, &pr[j]) != 0) return 7;
        for (int j = 0; j < P; ++j) pthread_join(th[j], NULL);
        const double wall = now_ms() - start;
        if (warm) continue;
        CHECK(gs_alloc_counters(&a1, &f1));
        long done = 0, bad = 0, nb = 0, ne = 0;
        double bm = 0, bx = 0, em = 0, ex = 0;
        for (int j = 0; j < P; ++j) {
          if (pr[j].status) return 4;
          done += pr[j].done; bad += pr[j].mismatches; nb += pr[j].begins; ne += pr[j].ends; bm += pr[j].begin_ms; em += pr[j].end_ms;
          if (pr[j].begin_max > bx) bx = pr[j].begin_max;
          if (pr[j].end_max > ex) ex = pr[j].end_max;
        }
        printf("route %s producers %d: %ld proofs (%ld + %ld) in %.0f ms = %.3f ms per proof = %.1f M constraints/s | _begin mean %.2f max %.2f ms, _end mean %.2f max %.2f ms | "
               "%llu hipMalloc %llu hipFree | %ld mismatches\n", route == 0 ? "witness_host" : "px_host", P, done, pr[0].done, P > 1 ? pr[1].done : 0L, wall,
               wall / (double)(done ? done : 1), (double)n * (double)done / wall / 1e3, bm / (double)(nb ? nb : 1), bx, em / (double)(ne ? ne : 1), ex,
               (unsigned long long)(a1 - a0), (unsigned long long)(f1 - f0), bad);
        if (bad || done == 0) { printf("FAIL: %ld of %ld streamed proofs differ from the first lap's\n", bad, done); return 5; }
      }
    }
  }
  CHECK(gs_free(r1cs)); CHECK(gs_free(key));
  gs_shutdown();
  printf("OK\n");
  return 0;
}
