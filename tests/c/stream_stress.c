/* Concurrency stress of one logical device (round 6: gs_*_end waits for the device outside the context's lock, blocking entry points work in a
 * free ticket slot).  Threads of four kinds hammer ONE resident key for `seconds`:
 *   producers   host-buffer tickets (witness route and w + px alternating), one ticket each, collected by the thread that began them;
 *   blockers    gs_groth16_prove_witness_host / gs_groth16_prove (blocking: they borrow a free ticket slot, or the fourth set);
 *   cancellers  begin a ticket and abandon it with gs_ticket_cancel;
 *   bystanders  gs_scalars_upload + gs_r1cs_px + gs_scalars_download + gs_free, gs_memory_query, gs_handle_bytes.
 * Every proof that comes back is compared byte for byte with the proof of its witness made before the threads started; every px that
 * comes back with the px of its witness.  argv: log2n nwit seconds producers blockers cancellers bystanders.  Ends with OK. */
#define _POSIX_C_SOURCE 200809L
#include <pthread.h>
#include <sched.h>
#include <time.h>

#include "instance.h"

typedef unsigned __int128 u128;
static const uint64_t FR[4] = {0x43e1f593f0000001ull, 0x2833e84879b97091ull, 0xb85045b68181585dull, 0x30644e72e131a029ull};
static const uint64_t FR_INV = 0xc2e1f593efffffffull;
static const uint64_t FR_R2[4] = {0x1bb8e645ae216da7ull, 0x53fe3ab1e35c59e3ull, 0x8c49833d53bb8085ull, 0x0216d0b17f4e44a5ull};
static int geq(const uint64_t a[4], const uint64_t b[4]) { for (int i = 3; i >= 0; --i) if (a[i] != b[i]) return a[i] > b[i]; return 1; }
static void sub_r(uint64_t a[4]) { u128 br = 0; for (int i = 0; i < 4; ++i) { u128 d = (u128)a[i] - FR[i] - br; a[i] = (uint64_t)d; br = (d >> 64) & 1; } }
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
static void add_mod(uint64_t a[4], const uint64_t b[4]) { u128 c = 0; for (int i = 0; i < 4; ++i) { c += (u128)a[i] + b[i]; a[i] = (uint64_t)c; c >>= 64; } if (c || geq(a, FR)) sub_r(a); }
static void sqchain_witness(uint64_t* w, size_t n, const uint64_t x[4]) {
  const uint64_t one[4] = {1, 0, 0, 0};
  uint64_t s[4], k[4] = {0, 0, 0, 0}, one_m[4];
  mont_mul(one_m, one, FR_R2); mont_mul(s, x, FR_R2);
  memcpy(w, one, 32); memcpy(w + 4, x, 32);
  for (size_t i = 1; i < n; ++i) { add_mod(k, one_m); mont_mul(s, s, s); add_mod(s, k); mont_mul(w + 4 * (i + 1), s, one); }
}
static double now_ms(void) { struct timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return ts.tv_sec * 1e3 + ts.tv_nsec * 1e-6; }
static uint64_t splitmix(uint64_t* s) { uint64_t z = (*s += 0x9e3779b97f4a7c15ull); z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull; z = (z ^ (z >> 27)) * 0x94d049bb133111ebull; return z ^ (z >> 31); }
static void field_elem(uint64_t out[4], uint64_t* seed) { do { for (int i = 0; i < 4; ++i) out[i] = splitmix(seed); out[3] &= 0x3fffffffffffffffull; } while (geq(out, FR)); }

typedef struct {
  int kind, id;
  gs_handle key, r1cs;
  size_t m, npx, nwit;
  uint64_t **w, **px, *rs, *want;
  double until;
  long ops, busy, bad;
  int status;
} worker;

static int proof_ok(const worker* p, size_t k, const uint64_t* got, const int* inf) {
  const uint64_t* want = p->want + 35 * k;
  return memcmp(got, want, 256) == 0 && (uint64_t)inf[0] == want[32] && (uint64_t)inf[1] == want[33] && (uint64_t)inf[2] == want[34];
}
static void* work(void* arg) {
  worker* p = (worker*)arg;
  uint64_t seed = 0x1234 + (uint64_t)p->id * 977, got[32];
  int inf[3];
  if (gs_set_device(0) != GS_OK) { p->status = 1; return NULL; }
  uint64_t* back = p->kind == 3 ? (uint64_t*)malloc(p->npx * 32) : NULL;
  while (now_ms() < p->until) {
    const size_t k = splitmix(&seed) % p->nwit;
    const int with_px = (int)(splitmix(&seed) & 1);
    if (p->kind == 0 || p->kind == 2) {                       /* producer / canceller */
      uint64_t t = 0;
      const int rc = with_px ? gs_groth16_prove_host_begin(p->key, p->w[k], p->m, p->px[k], p->npx, p->rs, p->rs + 4, &t)
                             : gs_groth16_prove_witness_host_begin(p->key, p->r1cs, p->w[k], p->m, p->rs, p->rs + 4, &t);
      if (rc == GS_ERR_BUSY) { p->busy += 1; sched_yield(); continue; }
      if (rc != GS_OK) { printf("FAIL begin: %d %s\n", rc, gs_last_error()); p->status = 2; return NULL; }
      if (p->kind == 2) { if (gs_ticket_cancel(t) != GS_OK) { printf("FAIL cancel: %s\n", gs_last_error()); p->status = 3; return NULL; } }
      else {
        if (gs_groth16_prove_end(t, got, inf) != GS_OK) { printf("FAIL end: %s\n", gs_last_error()); p->status = 4; return NULL; }
        if (!proof_ok(p, k, got, inf)) p->bad += 1;
      }
    } else if (p->kind == 1) {                                /* blocker */
      const int rc = with_px ? gs_groth16_prove(p->key, p->w[k], p->m, p->px[k], p->npx, p->rs, p->rs + 4, got, inf)
                             : gs_groth16_prove_witness_host(p->key, p->r1cs, p->w[k], p->m, p->rs, p->rs + 4, got, inf);
      if (rc != GS_OK) { printf("FAIL blocking: %d %s\n", rc, gs_last_error()); p->status = 5; return NULL; }
      if (!proof_ok(p, k, got, inf)) p->bad += 1;
    } else {                                                  /* bystander */
      gs_handle hw, hpx = 0;
      gs_memory mem;
      uint64_t ob = 0, tb = 0;
      if (gs_scalars_upload(p->w[k], p->m, &hw) != GS_OK || gs_r1cs_px(p->r1cs, hw, &hpx) != GS_OK || gs_scalars_download(hpx, back, p->npx) != GS_OK ||
          gs_free(hw) != GS_OK || gs_free(hpx) != GS_OK || gs_memory_query(&mem) != GS_OK || gs_handle_bytes(p->key, &ob, &tb) != GS_OK) {
        printf("FAIL bystander: %s\n", gs_last_error()); p->status = 6; return NULL;
      }
      if (memcmp(back, p->px[k], p->npx * 32) != 0) p->bad += 1;
    }
    p->ops += 1;
  }
  free(back);
  return NULL;
}

int main(int argc, char** argv) {
  if (argc < 8) { printf("usage: stream_stress <log2n> <nwit> <seconds> <producers> <blockers> <cancellers> <bystanders>\n"); return 9; }
  const size_t n = (size_t)1 << atoi(argv[1]), m = n + 1, nwit = (size_t)atoi(argv[2]), npx = 2 * n - 1;
  const double seconds = atof(argv[3]);
  const int counts[4] = {atoi(argv[4]), atoi(argv[5]), atoi(argv[6]), atoi(argv[7])};
  int dev = 0, total = counts[0] + counts[1] + counts[2] + counts[3];
  if (n < 4 || nwit < 2 || nwit > 64 || total < 1 || total > 32) return 9;
  CHECK(gs_init(&dev, 1));
  uint32_t* rp = (uint32_t*)malloc((n + 1) * 4), *acol = (uint32_t*)malloc(n * 4), *crp = (uint32_t*)malloc((n + 1) * 4), *ccol = (uint32_t*)malloc(2 * n * 4);
  uint64_t* aval = (uint64_t*)calloc(n * 4, 8), *cval = (uint64_t*)calloc(2 * n * 4, 8);
  if (!rp || !acol || !crp || !ccol || !aval || !cval) return 8;
  for (size_t k = 0; k < n; ++k) { rp[k] = (uint32_t)k; acol[k] = (uint32_t)(k + 1 < n ? k + 1 : 0); aval[4 * k] = 1; }
  rp[n] = (uint32_t)n;
  for (size_t k = 0; k + 1 < n; ++k) {
    crp[k] = (uint32_t)(2 * k); ccol[2 * k] = 0; ccol[2 * k + 1] = (uint32_t)(k + 2);
    uint64_t* v = cval + 8 * k;
    u128 br = 0;
    for (int i = 0; i < 4; ++i) { u128 d = (u128)FR[i] - (i == 0 ? (uint64_t)(k + 1) : 0) - br; v[i] = (uint64_t)d; br = (d >> 64) & 1; }
    v[4] = 1;
  }
  crp[n - 1] = (uint32_t)(2 * (n - 1)); crp[n] = (uint32_t)(2 * (n - 1) + 1); ccol[2 * (n - 1)] = 0; cval[8 * (n - 1)] = 1;
  uint64_t seed = 0x5EED0007ull, toxic[20], rs[8];
  for (int i = 0; i < 5; ++i) field_elem(toxic + 4 * i, &seed);
  field_elem(rs, &seed); field_elem(rs + 4, &seed);
  gs_handle key, r1cs;
  CHECK(gs_groth16_setup(n, m, 1, rp, acol, aval, rp, acol, aval, crp, ccol, cval, toxic, &key, NULL));
  CHECK(gs_r1cs_upload(n, m, rp, acol, aval, rp, acol, aval, crp, ccol, cval, &r1cs));
  uint64_t** w = (uint64_t**)malloc(nwit * sizeof *w), **px = (uint64_t**)malloc(nwit * sizeof *px), *want = (uint64_t*)calloc(nwit * 35, 8);
  for (size_t k = 0; k < nwit; ++k) {
    uint64_t x[4];
    int inf[3];
    w[k] = (uint64_t*)malloc(m * 32); px[k] = (uint64_t*)malloc(npx * 32);
    field_elem(x, &seed);
    sqchain_witness(w[k], n, x);
    gs_handle hw, hpx = 0;
    CHECK(gs_scalars_upload(w[k], m, &hw)); CHECK(gs_r1cs_px(r1cs, hw, &hpx)); CHECK(gs_scalars_download(hpx, px[k], npx));
    CHECK(gs_free(hw)); CHECK(gs_free(hpx));
    CHECK(gs_groth16_prove(key, w[k], m, px[k], npx, rs, rs + 4, want + 35 * k, inf));
    for (int i = 0; i < 3; ++i) want[35 * k + 32 + i] = (uint64_t)inf[i];
  }
  worker ws[32];
  pthread_t th[32];
  memset(ws, 0, sizeof ws);
  const double until = now_ms() + seconds * 1e3;
  int id = 0;
  for (int kind = 0; kind < 4; ++kind)
    for (int j = 0; j < counts[kind]; ++j, ++id) {
      ws[id].kind = kind; ws[id].id = id; ws[id].key = key; ws[id].r1cs = r1cs; ws[id].m = m; ws[id].npx = npx; ws[id].nwit = nwit;
      ws[id].w = w; ws[id].px = px; ws[id].rs = rs; ws[id].want = want; ws[id].until = until;
    }
  for (int i = 0; i < total; ++i) if (pthread_create(&th[i], NULL, work, &ws[i]) != 0) return 7;
  for (int i = 0; i < total; ++i) pthread_join(th[i], NULL);
  const char* names[4] = {"producers", "blockers", "cancellers", "bystanders"};
  long bad = 0, ops[4] = {0, 0, 0, 0}, fewest[4] = {-1, -1, -1, -1}, busy = 0;
  for (int i = 0; i < total; ++i) {
    if (ws[i].status) return 4;
    bad += ws[i].bad; ops[ws[i].kind] += ws[i].ops; busy += ws[i].busy;
    if (fewest[ws[i].kind] < 0 || ws[i].ops < fewest[ws[i].kind]) fewest[ws[i].kind] = ws[i].ops;
  }
  for (int kind = 0; kind < 4; ++kind) printf("%s %d: %ld operations (the slowest thread: %ld)\n", names[kind], counts[kind], ops[kind], fewest[kind] < 0 ? 0 : fewest[kind]);
  printf("GS_ERR_BUSY answers: %ld; results that differ from the single-threaded ones: %ld\n", busy, bad);
  if (bad) { printf("FAIL\n"); return 5; }
  /* nobody may be locked out: the context's lock is first come, first served (runtime.h, FairMutex) */
  for (int kind = 0; kind < 4; ++kind)
    if (counts[kind] && seconds >= 2.0 && fewest[kind] < 5) { printf("FAIL: a thread of the %s was starved (%ld operations in %.0f s)\n", names[kind], fewest[kind], seconds); return 10; }
  /* nothing may be left in flight: three fresh tickets fit */
  uint64_t t[3], got[32];
  int inf[3];
  for (int i = 0; i < 3; ++i) CHECK(gs_groth16_prove_witness_host_begin(key, r1cs, w[0], m, rs, rs + 4, &t[i]));
  for (int i = 0; i < 3; ++i) { CHECK(gs_groth16_prove_end(t[i], got, inf)); if (memcmp(got, want, 256) != 0) { printf("FAIL: after the stress\n"); return 6; } }
  CHECK(gs_free(r1cs)); CHECK(gs_free(key));
  gs_shutdown();
  printf("OK\n");
  return 0;
}
