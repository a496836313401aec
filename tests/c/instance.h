/* Flat-limb instance files written by tests/c_util.py (little-endian u64 words, no padding). */
#ifndef GS_TEST_INSTANCE_H
#define GS_TEST_INSTANCE_H
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "gosnark_hip.h"

#define CHECK(x) do { int _s = (x); if (_s != 0) { printf("FAIL %s: %d %s\n", #x, _s, gs_last_error()); return 1; } } while (0)

static uint64_t* gs_take(uint64_t** p, size_t words) { uint64_t* r = *p; *p += words; return r; }

static uint64_t* gs_read_file(const char* path, size_t* words) {
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

static int read_groth_instance(const char* path, groth_instance* g) {
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
static int upload_groth_pk(const groth_instance* g, gs_handle* pk) {
  gs_handle hat, hb1, hb2, hcd, hpt;
  CHECK(gs_g1_upload(g->at, g->m, &hat)); CHECK(gs_g1_upload(g->b1, g->m, &hb1)); CHECK(gs_g2_upload(g->b2, g->m, &hb2));
  CHECK(gs_g1_upload(g->cd, g->m, &hcd)); CHECK(gs_g1_upload(g->pt, g->nptd, &hpt));
  CHECK(gs_groth16_pk_create(hat, hb1, hb2, hcd, hpt, g->abd, g->abd + 12, g->abd + 24, g->bd2, g->bd2 + 24, g->z, g->nz, g->m, g->npublic, pk));
  CHECK(gs_free(hat)); CHECK(gs_free(hb1)); CHECK(gs_free(hb2)); CHECK(gs_free(hcd)); CHECK(gs_free(hpt));
  return 0;
}

/* proof words (affine PiA | PiB | PiC) -> the Jacobian triples [x, y, 1] the verifier takes */
static void proof_to_jacobian(const uint64_t proof[32], const int inf[3], uint64_t jac[48]) {
  memset(jac, 0, 48 * 8);
  if (!inf[0]) { memcpy(jac, proof, 64); jac[8] = 1; }
  if (!inf[1]) { memcpy(jac + 12, proof + 8, 128); jac[28] = 1; }
  if (!inf[2]) { memcpy(jac + 36, proof + 24, 64); jac[44] = 1; }
}
#endif
