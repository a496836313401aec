/* Flat-limb instance files written by tests/c_util.py (little-endian u64 words, no padding). */
#ifndef GS_TEST_INSTANCE_H
#define GS_TEST_INSTANCE_H
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "gosnark_hip.h"

#define CHECK(x) do { int _s = (x); if (_s != 0) { printf("FAIL %s: %d %s\n", #x, _s, gs_last_error()); return 1; } } while (0)

static __attribute__((unused)) uint64_t* gs_take(uint64_t** p, size_t words) { uint64_t* r = *p; *p += words; return r; }

static __attribute__((unused)) uint64_t* gs_read_file(const char* path, size_t* words) {
  FILE* f = fopen(path, "rb");
  if (!f) return NULL;
  fseek(f, 0, SEEK_END);
  long bytes = ftell(f);
  fseek(f, 0, SEEK_SET);
  uint64_t* buf = (uint64_t*)malloc((size_t)bytes + 8);
  if (!buf || fread(buf, 1, (size_t)bytes, f) != (size_t)bytes) { fclose(f); free(buf); return NULL; }
  fclose(f);
  if (words) *words = (size_t)bytes / 8;
  return buf;
}

/* groth16.Pk / Vk / witness / px / (r, s) / public signals of one instance (c_util.write_groth_instance) */
typedef struct {
  size_t m, npx, nz, nptd, nic, npublic;
  uint64_t *at, *b1, *b2, *cd, *pt, *abd, *bd2, *z, *w, *px, *rs, *vka, *vk2, *ic, *pub;
} groth_instance;

static __attribute__((unused)) int read_groth_instance(const char* path, groth_instance* g) {
  uint64_t* p = gs_read_file(path, NULL);
  if (!p) return 1;
  g->m = p[0]; g->npx = p[1]; g->nz = p[2]; g->nptd = p[3]; g->nic = p[4]; g->npublic = p[5]; p += 6;
  g->at = gs_take(&p, g->m * 12); g->b1 = gs_take(&p, g->m * 12); g->b2 = gs_take(&p, g->m * 24); g->cd = gs_take(&p, g->m * 12);
  g->pt = gs_take(&p, g->nptd * 12); g->abd = gs_take(&p, 36); g->bd2 = gs_take(&p, 48); g->z = gs_take(&p, g->nz * 4);
  g->w = gs_take(&p, g->m * 4); g->px = gs_take(&p, g->npx * 4); g->rs = gs_take(&p, 8);
  g->vka = gs_take(&p, 12); g->vk2 = gs_take(&p, 72); g->ic = gs_take(&p, g->nic * 12); g->pub = gs_take(&p, (g->nic - 1) * 4);
  return 0;
}

/* upload the five arrays and assemble the resident key on the calling thread's current logical device */
static __attribute__((unused)) int upload_groth_pk(const groth_instance* g, gs_handle* pk) {
  gs_handle hat, hb1, hb2, hcd, hpt;
  CHECK(gs_g1_upload(g->at, g->m, &hat)); CHECK(gs_g1_upload(g->b1, g->m, &hb1)); CHECK(gs_g2_upload(g->b2, g->m, &hb2));
  CHECK(gs_g1_upload(g->cd, g->m, &hcd)); CHECK(gs_g1_upload(g->pt, g->nptd, &hpt));
  CHECK(gs_groth16_pk_create(hat, hb1, hb2, hcd, hpt, g->abd, g->abd + 12, g->abd + 24, g->bd2, g->bd2 + 24, g->z, g->nz, g->m, g->npublic, pk));
  CHECK(gs_free(hat)); CHECK(gs_free(hb1)); CHECK(gs_free(hb2)); CHECK(gs_free(hcd)); CHECK(gs_free(hpt));
  return 0;
}

/* snark.Pk / Vk / witness / px / public signals of one Pinocchio instance (c_util.write_pinocchio_instance) */
typedef struct {
  size_t m, npx, nz, ng1t, nic, npublic;
  uint64_t *a, *ap, *b, *bp, *c, *cp, *kp, *g1t, *z, *w, *px;
  uint64_t *vka, *vkb, *vkc, *g1kbg, *g2kbg, *g2kg, *vkz, *ic, *pub;
} pinocchio_instance;

static __attribute__((unused)) int read_pinocchio_instance(const char* path, pinocchio_instance* g) {
  uint64_t* p = gs_read_file(path, NULL);
  if (!p) return 1;
  g->m = p[0]; g->npx = p[1]; g->nz = p[2]; g->ng1t = p[3]; g->nic = p[4]; g->npublic = p[5]; p += 6;
  g->a = gs_take(&p, g->m * 12); g->ap = gs_take(&p, g->m * 12); g->b = gs_take(&p, g->m * 24); g->bp = gs_take(&p, g->m * 12);
  g->c = gs_take(&p, g->m * 12); g->cp = gs_take(&p, g->m * 12); g->kp = gs_take(&p, g->m * 12); g->g1t = gs_take(&p, g->ng1t * 12);
  g->z = gs_take(&p, g->nz * 4); g->w = gs_take(&p, g->m * 4); g->px = gs_take(&p, g->npx * 4);
  g->vka = gs_take(&p, 24); g->vkb = gs_take(&p, 12); g->vkc = gs_take(&p, 24); g->g1kbg = gs_take(&p, 12); g->g2kbg = gs_take(&p, 24);
  g->g2kg = gs_take(&p, 24); g->vkz = gs_take(&p, 24); g->ic = gs_take(&p, g->nic * 12); g->pub = gs_take(&p, (g->nic - 1) * 4);
  return 0;
}

/* a sparse R1CS (three CSR matrices, indices stored as u64 words in the file) + toxic values (c_util.write_r1cs) */
typedef struct {
  size_t n, m, npublic, ntoxic;
  uint32_t *rowptr[3], *col[3];
  uint64_t *val[3], *toxic;
} r1cs_instance;

static __attribute__((unused)) uint32_t* gs_narrow(uint64_t** p, size_t count) {
  uint32_t* out = (uint32_t*)malloc((count + 1) * sizeof(uint32_t));
  for (size_t i = 0; i < count; ++i) out[i] = (uint32_t)(*p)[i];
  *p += count;
  return out;
}

static __attribute__((unused)) int read_r1cs_instance(const char* path, r1cs_instance* g) {
  uint64_t* p = gs_read_file(path, NULL);
  if (!p) return 1;
  g->n = p[0]; g->m = p[1]; g->npublic = p[2]; g->ntoxic = p[3]; p += 4;
  for (int k = 0; k < 3; ++k) {
    size_t nnz = *p++;
    g->rowptr[k] = gs_narrow(&p, g->n + 1);
    g->col[k] = gs_narrow(&p, nnz);
    g->val[k] = gs_take(&p, nnz * 4);
  }
  g->toxic = gs_take(&p, g->ntoxic * 4);
  return 0;
}

static __attribute__((unused)) int write_words(const char* path, const uint64_t* w, size_t n) {
  FILE* f = fopen(path, "wb");
  if (!f) return 1;
  const size_t done = fwrite(w, 8, n, f);
  fclose(f);
  return done == n ? 0 : 1;
}

/* proof words (affine PiA | PiB | PiC) -> the Jacobian triples [x, y, 1] the verifier takes */
static __attribute__((unused)) void proof_to_jacobian(const uint64_t proof[32], const int inf[3], uint64_t jac[48]) {
  memset(jac, 0, 48 * 8);
  if (!inf[0]) { memcpy(jac, proof, 64); jac[8] = 1; }
  if (!inf[1]) { memcpy(jac + 12, proof + 8, 128); jac[28] = 1; }
  if (!inf[2]) { memcpy(jac + 36, proof + 24, 64); jac[44] = 1; }
}
#endif
