/* groth16hip.GenerateProofsFromWitness / snarkhip.GenerateProofsFromWitness (go/gosnarkhip/witness.go), as C: the circuit's sparse
 * R1CS is uploaded once (gs_r1cs_upload), then each proof needs only the witness -- CombinePolynomials + Div (r1csqap.go:191-216)
 * happen on the device (gs_groth16_prove_witness / gs_pinocchio_prove_witness, H(x) straight from the constraint values).
 * Checked here: px from gs_r1cs_px equals the px the reference computed (instance file), and both witness-route proofs equal
 * the proofs made from that px.  argv: r1cs file, groth16 instance, pinocchio instance, output (32 + 72 proof words). */
#include "instance.h"

int main(int argc, char** argv) {
  if (argc != 5) return 9;
  r1cs_instance q;
  groth_instance g;
  pinocchio_instance p;
  if (read_r1cs_instance(argv[1], &q) || read_groth_instance(argv[2], &g) || read_pinocchio_instance(argv[3], &p)) return 8;
  if (q.m != g.m || q.m != p.m || g.npx != 2 * q.n - 1) { printf("FAIL: instance shapes\n"); return 7; }
  int dev = 0, inf[3], inf_w[3], pinf[8], pinf_w[8];
  gs_handle gk, pk, r1cs, w, px = 0, h[8];
  uint64_t want[32], got[32 + 72], pwant[72];
  uint64_t* pxback = (uint64_t*)malloc(g.npx * 32);
  CHECK(gs_init(&dev, 1));
  if (upload_groth_pk(&g, &gk)) return 3;
  CHECK(gs_g1_upload(p.a, p.m, &h[0])); CHECK(gs_g1_upload(p.ap, p.m, &h[1])); CHECK(gs_g2_upload(p.b, p.m, &h[2]));
  CHECK(gs_g1_upload(p.bp, p.m, &h[3])); CHECK(gs_g1_upload(p.c, p.m, &h[4])); CHECK(gs_g1_upload(p.cp, p.m, &h[5]));
  CHECK(gs_g1_upload(p.kp, p.m, &h[6])); CHECK(gs_g1_upload(p.g1t, p.ng1t, &h[7]));
  CHECK(gs_pinocchio_pk_create(h[0], h[1], h[2], h[3], h[4], h[5], h[6], h[7], p.z, p.nz, p.m, p.npublic, &pk));
  for (int i = 0; i < 8; ++i) CHECK(gs_free(h[i]));
  CHECK(gs_r1cs_upload(q.n, q.m, q.rowptr[0], q.col[0], q.val[0], q.rowptr[1], q.col[1], q.val[1], q.rowptr[2], q.col[2], q.val[2], &r1cs));
  CHECK(gs_scalars_upload(g.w, g.m, &w));
  /* R1CS + witness -> px on the device == the reference's CombinePolynomials output */
  CHECK(gs_r1cs_px(r1cs, w, &px));
  CHECK(gs_scalars_download(px, pxback, g.npx));
  if (memcmp(pxback, g.px, g.npx * 32) != 0) { printf("FAIL: px differs from the reference's\n"); return 4; }
  /* Groth16 */
  CHECK(gs_groth16_prove_resident(gk, w, px, g.rs, g.rs + 4, want, inf));
  CHECK(gs_groth16_prove_witness(gk, r1cs, w, g.rs, g.rs + 4, got, inf_w));
  if (memcmp(want, got, sizeof want) != 0 || memcmp(inf, inf_w, sizeof inf) != 0) { printf("FAIL: groth16 witness route\n"); return 5; }
  /* Pinocchio (same circuit, same witness) */
  if (memcmp(p.w, g.w, g.m * 32) != 0) { printf("FAIL: the two instances do not share the witness\n"); return 6; }
  CHECK(gs_pinocchio_prove_resident(pk, w, px, pwant, pinf));
  CHECK(gs_pinocchio_prove_witness(pk, r1cs, w, got + 32, pinf_w));
  if (memcmp(pwant, got + 32, sizeof pwant) != 0 || memcmp(pinf, pinf_w, sizeof pinf) != 0) { printf("FAIL: pinocchio witness route\n"); return 10; }
  if (write_words(argv[4], got, 32 + 72)) return 11;
  CHECK(gs_free(px)); CHECK(gs_free(w)); CHECK(gs_free(r1cs)); CHECK(gs_free(gk)); CHECK(gs_free(pk));
  gs_shutdown();
  printf("OK\n");
  return 0;
}
