/* CPU ORACLE (test infrastructure, NOT product code) -- C restatement of the reference's naive
 * prover arithmetic for sizes the pure-Python oracle (oracle/ref_py.py) cannot reach.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library.
 *
 * Follows, operation for operation (so that the raw JACOBIAN output is bit-identical to the
 * reference's, which tests/test_oracle_c.py pins against ref_py and the reference's wasm goldens):
 *   fields/fq.go:32-98          Fq.Add/Sub/Mul/Square/...      -> fq_* (4x64-bit Montgomery limbs
 *                               instead of math/big; same residues)
 *   fields/fq2.go:37-133        Fq2.Add/Sub/Mul/Square          -> fq2_*
 *   bn128/g1.go:32-89           G1.Add  (add-2007-bl, no P==Q branch)   -> g1_add
 *   bn128/g1.go:101-138         G1.Double (dbl-2009-l)                  -> g1_double
 *   bn128/g1.go:140-155         G1.MulScalar (MSB-first double-and-add) -> g1_mul_scalar
 *   bn128/g2.go:32-181          the same over Fq2                       -> g2_*
 *   groth16/groth16.go:243-250  acc = Add(acc, MulScalar(base_i, w_i))  -> oracle_g1_msm_naive
 *   r1csqap/r1csqap.go:57-115   PolynomialField.Mul/Div/Add/Sub         -> oracle_poly_*
 *   r1csqap/r1csqap.go:129-158  NewPolZeroAt / LagrangeInterpolation    -> oracle_lagrange
 *
 * Data at the boundary: standard-form little-endian 4 x uint64 per field element, points as the
 * reference's Jacobian triples.  Build: make -C oracle  (gcc -O2, -lpthread).
 */
#include <pthread.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

typedef unsigned __int128 u128;
typedef struct { uint64_t v[4]; } fe;            /* Montgomery form, canonical [0, p) */
typedef struct { const uint64_t p[4]; uint64_t inv; fe one; fe r2; } field;

static const uint64_t Q_[4] = {0x3c208c16d87cfd47ull, 0x97816a916871ca8dull, 0xb85045b68181585dull, 0x30644e72e131a029ull};
static const uint64_t R_[4] = {0x43e1f593f0000001ull, 0x2833e84879b97091ull, 0xb85045b68181585dull, 0x30644e72e131a029ull};
/* SURVEY.md App. D */
static const field FQ = {{0x3c208c16d87cfd47ull, 0x97816a916871ca8dull, 0xb85045b68181585dull, 0x30644e72e131a029ull},
                         0x87d20782e4866389ull,
                         {{0xd35d438dc58f0d9dull, 0x0a78eb28f5c70b3dull, 0x666ea36f7879462cull, 0x0e0a77c19a07df2full}},
                         {{0xf32cfc5b538afa89ull, 0xb5e71911d44501fbull, 0x47ab1eff0a417ff6ull, 0x06d89f71cab8351full}}};
static const field FR = {{0x43e1f593f0000001ull, 0x2833e84879b97091ull, 0xb85045b68181585dull, 0x30644e72e131a029ull},
                         0xc2e1f593efffffffull,
                         {{0xac96341c4ffffffbull, 0x36fc76959f60cd29ull, 0x666ea36f7879462eull, 0x0e0a77c19a07df2full}},
                         {{0x1bb8e645ae216da7ull, 0x53fe3ab1e35c59e3ull, 0x8c49833d53bb8085ull, 0x0216d0b17f4e44a5ull}}};

static int ge_p(const uint64_t a[4], const uint64_t p[4]) {
  for (int i = 3; i >= 0; --i) { if (a[i] > p[i]) return 1; if (a[i] < p[i]) return 0; }
  return 1;
}
static void sub_p(uint64_t a[4], const uint64_t p[4]) {
  u128 b = 0;
  for (int i = 0; i < 4; ++i) { u128 d = (u128)a[i] - p[i] - (uint64_t)b; a[i] = (uint64_t)d; b = (d >> 64) & 1; }
}
static int fe_is_zero(const fe* a) { return (a->v[0] | a->v[1] | a->v[2] | a->v[3]) == 0; }

static void fe_add(const field* f, fe* r, const fe* a, const fe* b) {          /* fq.go:32-35 */
  u128 c = 0; uint64_t t[4];
  for (int i = 0; i < 4; ++i) { c += (u128)a->v[i] + b->v[i]; t[i] = (uint64_t)c; c >>= 64; }
  if (c || ge_p(t, f->p)) sub_p(t, f->p);
  memcpy(r->v, t, 32);
}
static void fe_sub(const field* f, fe* r, const fe* a, const fe* b) {          /* fq.go:44-47 */
  u128 bo = 0; uint64_t t[4];
  for (int i = 0; i < 4; ++i) { u128 d = (u128)a->v[i] - b->v[i] - (uint64_t)bo; t[i] = (uint64_t)d; bo = (d >> 64) & 1; }
  if (bo) { u128 c = 0; for (int i = 0; i < 4; ++i) { c += (u128)t[i] + f->p[i]; t[i] = (uint64_t)c; c >>= 64; } }
  memcpy(r->v, t, 32);
}
static void fe_mul(const field* f, fe* r, const fe* a, const fe* b) {          /* fq.go:56-59 (CIOS) */
  uint64_t t[6] = {0, 0, 0, 0, 0, 0};
  for (int i = 0; i < 4; ++i) {
    u128 c = 0;
    for (int j = 0; j < 4; ++j) { c += (u128)a->v[j] * b->v[i] + t[j]; t[j] = (uint64_t)c; c >>= 64; }
    c += t[4]; t[4] = (uint64_t)c; t[5] = (uint64_t)(c >> 64);
    uint64_t m = t[0] * f->inv;
    c = (u128)m * f->p[0] + t[0]; c >>= 64;
    for (int j = 1; j < 4; ++j) { c += (u128)m * f->p[j] + t[j]; t[j - 1] = (uint64_t)c; c >>= 64; }
    c += t[4]; t[3] = (uint64_t)c; t[4] = t[5] + (uint64_t)(c >> 64);
  }
  if (t[4] || ge_p(t, f->p)) sub_p(t, f->p);
  memcpy(r->v, t, 32);
}
static void fe_sqr(const field* f, fe* r, const fe* a) { fe_mul(f, r, a, a); }  /* fq.go:95-98 */
static void fe_dbl(const field* f, fe* r, const fe* a) { fe_add(f, r, a, a); }  /* fq.go:38-41 */
static void fe_neg(const field* f, fe* r, const fe* a) { fe z = {{0, 0, 0, 0}}; fe_sub(f, r, &z, a); }
static void fe_from_std(const field* f, fe* r, const uint64_t w[4]) {
  fe t; memcpy(t.v, w, 32);
  while (ge_p(t.v, f->p)) sub_p(t.v, f->p);
  fe_mul(f, r, &t, &f->r2);
}
static void fe_to_std(const field* f, uint64_t w[4], const fe* a) {
  fe one = {{1, 0, 0, 0}}, t;
  fe_mul(f, &t, a, &one);
  memcpy(w, t.v, 32);
}
static void fe_inv(const field* f, fe* r, const fe* a) {                       /* fq.go:66-67 via Fermat */
  uint64_t e[4]; memcpy(e, f->p, 32); e[0] -= 2;
  fe acc = f->one, base = *a;
  for (int i = 0; i < 256; ++i) {
    if ((e[i >> 6] >> (i & 63)) & 1) fe_mul(f, &acc, &acc, &base);
    fe_mul(f, &base, &base, &base);
  }
  *r = acc;
}

/* ---- Fq2 (fields/fq2.go) ------------------------------------------------------------------ */
typedef struct { fe c0, c1; } fe2;
static void fq2_add(fe2* r, const fe2* a, const fe2* b) { fe_add(&FQ, &r->c0, &a->c0, &b->c0); fe_add(&FQ, &r->c1, &a->c1, &b->c1); }
static void fq2_sub(fe2* r, const fe2* a, const fe2* b) { fe_sub(&FQ, &r->c0, &a->c0, &b->c0); fe_sub(&FQ, &r->c1, &a->c1, &b->c1); }
static void fq2_mul(fe2* r, const fe2* a, const fe2* b) {                      /* fq2.go:63-76, u^2 = -1 */
  fe v0, v1, s0, s1, t;
  fe_mul(&FQ, &v0, &a->c0, &b->c0); fe_mul(&FQ, &v1, &a->c1, &b->c1);
  fe_add(&FQ, &s0, &a->c0, &a->c1); fe_add(&FQ, &s1, &b->c0, &b->c1);
  fe_mul(&FQ, &t, &s0, &s1);
  fe_sub(&FQ, &r->c0, &v0, &v1);
  fe_sub(&FQ, &t, &t, &v0); fe_sub(&FQ, &r->c1, &t, &v1);
}
static void fq2_sqr(fe2* r, const fe2* a) { fe2 t = *a; fq2_mul(r, &t, &t); }   /* fq2.go:118-133 (same value) */
static int fq2_is_zero(const fe2* a) { return fe_is_zero(&a->c0) && fe_is_zero(&a->c1); }

/* ---- generic Jacobian formulas, instantiated for Fq (G1) and Fq2 (G2) by macro ------------- */
#define DEFINE_CURVE(P, E, ADD, SUB, MUL, SQR, ISZ)                                                 \
  typedef struct { E x, y, z; } P##_pt;                                                           \
  static void P##_add(P##_pt* r, const P##_pt* p1, const P##_pt* p2) { /* g1.go:32-89 / g2.go:32-89 */ \
    if (ISZ(&p1->z)) { *r = *p2; return; }                                                        \
    if (ISZ(&p2->z)) { *r = *p1; return; }                                                        \
    E z1z1, z2z2, u1, u2, t0, s1, t1, s2, h, t2, i, j, t3, rr, v, t4, t5, t6, x3, t7, t8, t9, t10, y3, t11, t12, t13, t14, z3; \
    SQR(&z1z1, &p1->z); SQR(&z2z2, &p2->z);                                                       \
    MUL(&u1, &p1->x, &z2z2); MUL(&u2, &p2->x, &z1z1);                                             \
    MUL(&t0, &p2->z, &z2z2); MUL(&s1, &p1->y, &t0);                                               \
    MUL(&t1, &p1->z, &z1z1); MUL(&s2, &p2->y, &t1);                                               \
    SUB(&h, &u2, &u1); ADD(&t2, &h, &h); SQR(&i, &t2); MUL(&j, &h, &i);                           \
    SUB(&t3, &s2, &s1); ADD(&rr, &t3, &t3); MUL(&v, &u1, &i);                                     \
    SQR(&t4, &rr); ADD(&t5, &v, &v); SUB(&t6, &t4, &j); SUB(&x3, &t6, &t5);                       \
    SUB(&t7, &v, &x3); MUL(&t8, &s1, &j); ADD(&t9, &t8, &t8); MUL(&t10, &rr, &t7);                \
    SUB(&y3, &t10, &t9);                                                                          \
    ADD(&t11, &p1->z, &p2->z); SQR(&t12, &t11); SUB(&t13, &t12, &z1z1); SUB(&t14, &t13, &z2z2);   \
    MUL(&z3, &t14, &h);                                                                           \
    r->x = x3; r->y = y3; r->z = z3;                                                              \
  }                                                                                               \
  static void P##_double(P##_pt* r, const P##_pt* p) {        /* g1.go:101-138 / g2.go:103-140 */  \
    if (ISZ(&p->z)) { *r = *p; return; }                                                          \
    E a, b, c, t0, t1, t2, t3, d, e, f, t4, x3, t5, twoC, fourC, t6, t7, y3, t8, z3;              \
    SQR(&a, &p->x); SQR(&b, &p->y); SQR(&c, &b);                                                  \
    ADD(&t0, &p->x, &b); SQR(&t1, &t0); SUB(&t2, &t1, &a); SUB(&t3, &t2, &c);                     \
    ADD(&d, &t3, &t3); ADD(&e, &a, &a); ADD(&e, &e, &a); SQR(&f, &e);                             \
    ADD(&t4, &d, &d); SUB(&x3, &f, &t4); SUB(&t5, &d, &x3);                                       \
    ADD(&twoC, &c, &c); ADD(&fourC, &twoC, &twoC); ADD(&t6, &fourC, &fourC);                      \
    MUL(&t7, &e, &t5); SUB(&y3, &t7, &t6);                                                        \
    MUL(&t8, &p->y, &p->z); ADD(&z3, &t8, &t8);                                                   \
    r->x = x3; r->y = y3; r->z = z3;                                                              \
  }                                                                                               \
  static void P##_mul_scalar(P##_pt* r, const P##_pt* p, const uint64_t k[4]) { /* g1.go:140-155 */ \
    P##_pt q; memset(&q, 0, sizeof q);                                                            \
    int top = -1;                                                                                 \
    for (int i = 255; i >= 0; --i) if ((k[i >> 6] >> (i & 63)) & 1) { top = i; break; }           \
    for (int i = top; i >= 0; --i) {                                                              \
      P##_double(&q, &q);                                                                         \
      if ((k[i >> 6] >> (i & 63)) & 1) P##_add(&q, &q, p);                                        \
    }                                                                                             \
    *r = q;                                                                                       \
  }

#define FQ_ADD(r, a, b) fe_add(&FQ, r, a, b)
#define FQ_SUB(r, a, b) fe_sub(&FQ, r, a, b)
#define FQ_MUL(r, a, b) fe_mul(&FQ, r, a, b)
#define FQ_SQR(r, a) fe_sqr(&FQ, r, a)
DEFINE_CURVE(g1, fe, FQ_ADD, FQ_SUB, FQ_MUL, FQ_SQR, fe_is_zero)
DEFINE_CURVE(g2, fe2, fq2_add, fq2_sub, fq2_mul, fq2_sqr, fq2_is_zero)

static void g1_load(g1_pt* p, const uint64_t* w) { fe_from_std(&FQ, &p->x, w); fe_from_std(&FQ, &p->y, w + 4); fe_from_std(&FQ, &p->z, w + 8); }
static void g1_store(uint64_t* w, const g1_pt* p) { fe_to_std(&FQ, w, &p->x); fe_to_std(&FQ, w + 4, &p->y); fe_to_std(&FQ, w + 8, &p->z); }
static void g2_load(g2_pt* p, const uint64_t* w) {
  fe_from_std(&FQ, &p->x.c0, w); fe_from_std(&FQ, &p->x.c1, w + 4); fe_from_std(&FQ, &p->y.c0, w + 8);
  fe_from_std(&FQ, &p->y.c1, w + 12); fe_from_std(&FQ, &p->z.c0, w + 16); fe_from_std(&FQ, &p->z.c1, w + 20);
}
static void g2_store(uint64_t* w, const g2_pt* p) {
  fe_to_std(&FQ, w, &p->x.c0); fe_to_std(&FQ, w + 4, &p->x.c1); fe_to_std(&FQ, w + 8, &p->y.c0);
  fe_to_std(&FQ, w + 12, &p->y.c1); fe_to_std(&FQ, w + 16, &p->z.c0); fe_to_std(&FQ, w + 20, &p->z.c1);
}

/* ---- the reference's prover loop: acc = Add(acc, MulScalar(base_i, k_i)), i ascending ------- */
/* groth16.go:243-250,269-271 / snark.go:265-286.  Scalars are used as given (no reduction mod r),
 * like big.Int in g1.go:145-147.  Output: raw Jacobian, bit-identical to the reference. */
void oracle_g1_msm_naive(const uint64_t* pts, const uint64_t* scalars, size_t n, uint64_t out[12]) {
  g1_pt acc; memset(&acc, 0, sizeof acc);
  for (size_t i = 0; i < n; ++i) {
    g1_pt p, t; g1_load(&p, pts + 12 * i);
    g1_mul_scalar(&t, &p, scalars + 4 * i);
    g1_add(&acc, &acc, &t);
  }
  g1_store(out, &acc);
}
void oracle_g2_msm_naive(const uint64_t* pts, const uint64_t* scalars, size_t n, uint64_t out[24]) {
  g2_pt acc; memset(&acc, 0, sizeof acc);
  for (size_t i = 0; i < n; ++i) {
    g2_pt p, t; g2_load(&p, pts + 24 * i);
    g2_mul_scalar(&t, &p, scalars + 4 * i);
    g2_add(&acc, &acc, &t);
  }
  g2_store(out, &acc);
}
void oracle_g1_add(const uint64_t a[12], const uint64_t b[12], uint64_t out[12]) {
  g1_pt x, y, r; g1_load(&x, a); g1_load(&y, b); g1_add(&r, &x, &y); g1_store(out, &r);
}
void oracle_g2_add(const uint64_t a[24], const uint64_t b[24], uint64_t out[24]) {
  g2_pt x, y, r; g2_load(&x, a); g2_load(&y, b); g2_add(&r, &x, &y); g2_store(out, &r);
}
void oracle_g1_mul_scalar(const uint64_t p[12], const uint64_t k[4], uint64_t out[12]) {
  g1_pt x, r; g1_load(&x, p); g1_mul_scalar(&r, &x, k); g1_store(out, &r);
}
void oracle_g2_mul_scalar(const uint64_t p[24], const uint64_t k[4], uint64_t out[24]) {
  g2_pt x, r; g2_load(&x, p); g2_mul_scalar(&r, &x, k); g2_store(out, &r);
}

/* Multi-threaded variant for the "all host cores" CPU baseline: term range split into nthreads
 * contiguous slices, each the literal loop; partial sums added in slice order.  The Jacobian
 * representative differs from the single-thread one (different add order), the point does not. */
typedef struct { const uint64_t* pts; const uint64_t* sc; size_t n; int g2; uint64_t out[24]; } slice_job;
static void* slice_run(void* arg) {
  slice_job* j = (slice_job*)arg;
  if (j->g2) oracle_g2_msm_naive(j->pts, j->sc, j->n, j->out); else oracle_g1_msm_naive(j->pts, j->sc, j->n, j->out);
  return NULL;
}
void oracle_msm_naive_mt(const uint64_t* pts, const uint64_t* scalars, size_t n, int g2, int nthreads, uint64_t* out) {
  if (nthreads < 1) nthreads = 1;
  slice_job* jobs = (slice_job*)calloc((size_t)nthreads, sizeof(slice_job));
  pthread_t* th = (pthread_t*)calloc((size_t)nthreads, sizeof(pthread_t));
  const size_t pw = g2 ? 24 : 12;
  for (int t = 0; t < nthreads; ++t) {
    size_t b = n * (size_t)t / (size_t)nthreads, e = n * (size_t)(t + 1) / (size_t)nthreads;
    jobs[t].pts = pts + pw * b; jobs[t].sc = scalars + 4 * b; jobs[t].n = e - b; jobs[t].g2 = g2;
    pthread_create(&th[t], NULL, slice_run, &jobs[t]);
  }
  for (int t = 0; t < nthreads; ++t) pthread_join(th[t], NULL);
  if (g2) { g2_pt acc, p; memset(&acc, 0, sizeof acc); for (int t = 0; t < nthreads; ++t) { g2_load(&p, jobs[t].out); g2_add(&acc, &acc, &p); } g2_store(out, &acc); }
  else { g1_pt acc, p; memset(&acc, 0, sizeof acc); for (int t = 0; t < nthreads; ++t) { g1_load(&p, jobs[t].out); g1_add(&acc, &acc, &p); } g1_store(out, &acc); }
  free(jobs); free(th);
}

/* out[i] = MulScalar(base, k_i): the encryption loops of the trusted setups (groth16.go:139-175, snark.go:152-230) over nthreads
 * slices; used by oracle/gen_golden_large.py to rebuild on the CPU the key points the synthetic instances define as k_i * G. */
typedef struct { const uint64_t* base; const uint64_t* sc; size_t n; int g2; uint64_t* out; } fb_job;
static void* fb_run(void* arg) {
  fb_job* j = (fb_job*)arg;
  for (size_t i = 0; i < j->n; ++i) {
    if (j->g2) oracle_g2_mul_scalar(j->base, j->sc + 4 * i, j->out + 24 * i);
    else oracle_g1_mul_scalar(j->base, j->sc + 4 * i, j->out + 12 * i);
  }
  return NULL;
}
void oracle_mul_scalar_batch_mt(const uint64_t* base, const uint64_t* scalars, size_t n, int g2, int nthreads, uint64_t* out) {
  if (nthreads < 1) nthreads = 1;
  fb_job* jobs = (fb_job*)calloc((size_t)nthreads, sizeof(fb_job));
  pthread_t* th = (pthread_t*)calloc((size_t)nthreads, sizeof(pthread_t));
  const size_t pw = g2 ? 24 : 12;
  for (int t = 0; t < nthreads; ++t) {
    size_t b = n * (size_t)t / (size_t)nthreads, e = n * (size_t)(t + 1) / (size_t)nthreads;
    jobs[t].base = base; jobs[t].sc = scalars + 4 * b; jobs[t].n = e - b; jobs[t].g2 = g2; jobs[t].out = out + pw * b;
    pthread_create(&th[t], NULL, fb_run, &jobs[t]);
  }
  for (int t = 0; t < nthreads; ++t) pthread_join(th[t], NULL);
  free(jobs); free(th);
}

/* affine normal form (g1.go:157-170 / g2.go:183-200); returns 1 for infinity */
int oracle_g1_affine(const uint64_t jac[12], uint64_t out[8]) {
  g1_pt p; g1_load(&p, jac); memset(out, 0, 64);
  if (fe_is_zero(&p.z)) return 1;
  fe zi, zi2, zi3, x, y;
  fe_inv(&FQ, &zi, &p.z); fe_sqr(&FQ, &zi2, &zi); fe_mul(&FQ, &x, &p.x, &zi2);
  fe_mul(&FQ, &zi3, &zi2, &zi); fe_mul(&FQ, &y, &p.y, &zi3);
  fe_to_std(&FQ, out, &x); fe_to_std(&FQ, out + 4, &y);
  return 0;
}
int oracle_g2_affine(const uint64_t jac[24], uint64_t out[16]) {
  g2_pt p; g2_load(&p, jac); memset(out, 0, 128);
  if (fq2_is_zero(&p.z)) return 1;
  /* fq2.go:99-110 inverse via the norm */
  fe t0, t1, n, ni; fe2 zi, zi2, zi3, x, y;
  fe_sqr(&FQ, &t0, &p.z.c0); fe_sqr(&FQ, &t1, &p.z.c1); fe_add(&FQ, &n, &t0, &t1); fe_inv(&FQ, &ni, &n);
  fe_mul(&FQ, &zi.c0, &p.z.c0, &ni); fe_mul(&FQ, &t0, &p.z.c1, &ni); fe_neg(&FQ, &zi.c1, &t0);
  fq2_sqr(&zi2, &zi); fq2_mul(&x, &p.x, &zi2); fq2_mul(&zi3, &zi2, &zi); fq2_mul(&y, &p.y, &zi3);
  fe_to_std(&FQ, out, &x.c0); fe_to_std(&FQ, out + 4, &x.c1); fe_to_std(&FQ, out + 8, &y.c0); fe_to_std(&FQ, out + 12, &y.c1);
  return 0;
}

/* ---- r1csqap/r1csqap.go over Fr ---------------------------------------------------------------- */
static fe* load_poly(const uint64_t* w, size_t n) {
  fe* p = (fe*)malloc((n ? n : 1) * sizeof(fe));
  for (size_t i = 0; i < n; ++i) fe_from_std(&FR, &p[i], w + 4 * i);
  return p;
}
static void store_poly(uint64_t* w, const fe* p, size_t n) { for (size_t i = 0; i < n; ++i) fe_to_std(&FR, w + 4 * i, &p[i]); }

/* PolynomialField.Mul, r1csqap.go:57-67 */
void oracle_poly_mul(const uint64_t* a, size_t na, const uint64_t* b, size_t nb, uint64_t* out) {
  fe *pa = load_poly(a, na), *pb = load_poly(b, nb);
  size_t nr = na + nb - 1;
  fe* r = (fe*)calloc(nr, sizeof(fe));
  for (size_t i = 0; i < na; ++i)
    for (size_t j = 0; j < nb; ++j) { fe t; fe_mul(&FR, &t, &pa[i], &pb[j]); fe_add(&FR, &r[i + j], &r[i + j], &t); }
  store_poly(out, r, nr);
  free(pa); free(pb); free(r);
}
/* PolynomialField.Div, r1csqap.go:70-84: long division.  The reference multiplies b by a dense
 * zero-padded monomial each step (:78-80, O(n^2) per step); only the non-zero products are
 * formed here -- identical values, O(n^2) total. */
void oracle_poly_div(const uint64_t* a, size_t na, const uint64_t* b, size_t nb, uint64_t* quo, uint64_t* rem) {
  fe *rm = load_poly(a, na), *pb = load_poly(b, nb);
  size_t nq = na - nb + 1;
  fe* q = (fe*)calloc(nq, sizeof(fe));
  fe lead_inv; fe_inv(&FR, &lead_inv, &pb[nb - 1]);
  size_t len = na;
  while (len >= nb) {
    fe l; fe_mul(&FR, &l, &rm[len - 1], &lead_inv);         /* :75 F.Div(rem[last], b[last]) */
    size_t pos = len - nb;
    q[pos] = l;
    for (size_t j = 0; j < nb; ++j) { fe t; fe_mul(&FR, &t, &pb[j], &l); fe_sub(&FR, &rm[pos + j], &rm[pos + j], &t); }
    len -= 1;                                               /* :81 rem = aux2[:len-1] */
  }
  store_poly(quo, q, nq);
  if (rem) store_poly(rem, rm, nb - 1);
  free(rm); free(pb); free(q);
}
/* LagrangeInterpolation on nodes 1..n, r1csqap.go:129-158, with the mathematically exact
 * denominator (the reference's Go-int `fac` wraps for n >= 22).  O(n^2): master polynomial
 * N(x) = prod (x - k) once, then per node synthetic division N / (x - j) scaled by v_j / N'(j). */
void oracle_lagrange(const uint64_t* values, size_t n, uint64_t* coeffs) {
  fe* v = load_poly(values, n);
  fe* N = (fe*)calloc(n + 1, sizeof(fe));
  N[0] = FR.one;
  size_t deg = 0;
  for (size_t k = 1; k <= n; ++k) {                          /* N *= (x - k) */
    uint64_t kw[4] = {k, 0, 0, 0}; fe kf; fe_from_std(&FR, &kf, kw);
    N[deg + 1] = N[deg];
    for (size_t i = deg; i >= 1; --i) { fe t; fe_mul(&FR, &t, &N[i], &kf); fe_sub(&FR, &N[i], &N[i - 1], &t); }
    { fe t; fe_mul(&FR, &t, &N[0], &kf); fe z = {{0, 0, 0, 0}}; fe_sub(&FR, &N[0], &z, &t); }
    deg += 1;
  }
  fe* out = (fe*)calloc(n ? n : 1, sizeof(fe));
  fe* qd = (fe*)calloc(n ? n : 1, sizeof(fe));
  for (size_t j = 1; j <= n; ++j) {
    if (fe_is_zero(&v[j - 1])) continue;
    uint64_t jw[4] = {j, 0, 0, 0}; fe jf; fe_from_std(&FR, &jf, jw);
    /* qd = N / (x - j) by synthetic division (degree n-1) */
    qd[n - 1] = N[n];
    for (size_t i = n - 1; i >= 1; --i) { fe t; fe_mul(&FR, &t, &qd[i], &jf); fe_add(&FR, &qd[i - 1], &N[i], &t); }
    /* denominator = qd(j) = prod_{k != j} (j - k) */
    fe den = qd[n - 1];
    for (size_t i = n - 1; i >= 1; --i) { fe t; fe_mul(&FR, &t, &den, &jf); fe_add(&FR, &den, &t, &qd[i - 1]); }
    fe di, sc; fe_inv(&FR, &di, &den); fe_mul(&FR, &sc, &v[j - 1], &di);
    for (size_t i = 0; i < n; ++i) { fe t; fe_mul(&FR, &t, &qd[i], &sc); fe_add(&FR, &out[i], &out[i], &t); }
  }
  store_poly(coeffs, out, n);
  free(v); free(N); free(out); free(qd);
}
/* PolynomialField.Eval, r1csqap.go:118-126 (Horner gives the same residue as the Exp-per-term sum) */
void oracle_poly_eval(const uint64_t* v, size_t n, const uint64_t x[4], uint64_t out[4]) {
  fe* p = load_poly(v, n); fe xf, acc = {{0, 0, 0, 0}};
  fe_from_std(&FR, &xf, x);
  for (size_t i = n; i >= 1; --i) { fe t; fe_mul(&FR, &t, &acc, &xf); fe_add(&FR, &acc, &t, &p[i - 1]); }
  fe_to_std(&FR, out, &acc);
  free(p);
}
/* element-wise helpers for test generators: out = a * b mod r, out = a^-1 */
void oracle_fr_mul(const uint64_t a[4], const uint64_t b[4], uint64_t out[4]) {
  fe x, y, z; fe_from_std(&FR, &x, a); fe_from_std(&FR, &y, b); fe_mul(&FR, &z, &x, &y); fe_to_std(&FR, out, &z);
}
void oracle_fr_inv(const uint64_t a[4], uint64_t out[4]) {
  fe x, z; fe_from_std(&FR, &x, a); fe_inv(&FR, &z, &x); fe_to_std(&FR, out, &z);
}
const uint64_t* oracle_modulus_q(void) { return Q_; }
const uint64_t* oracle_modulus_r(void) { return R_; }
