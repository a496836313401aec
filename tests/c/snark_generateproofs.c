/* snarkhip.GenerateProofs + VerifyProof (go/snarkhip/snarkhip.go) = gosnarkhip.NewPinocchioKey + (*PinocchioKey).Prove +
 * PinocchioVerify, as C:  8 x gs_g*_upload -> gs_pinocchio_pk_create -> 8 x gs_free -> gs_pinocchio_prove -> gs_pinocchio_verify.
 * Replaces snark.GenerateProofs (snark.go:254-289) / snark.VerifyProof (snark.go:292-368).
 * argv: instance file, output file (72 proof words | inf[8] | ok, failed_check for the right input | ok, failed_check for a wrong one). */
#include "instance.h"

int main(int argc, char** argv) {
  if (argc != 3) return 9;
  pinocchio_instance g;
  if (read_pinocchio_instance(argv[1], &g)) return 8;
  int dev = 0, inf[8], ok = 0, failed = -1, ok2 = 1, failed2 = -1;
  gs_handle h[8], pk;
  uint64_t out[84], proof[108], wrong[4] = {34, 0, 0, 0};
  CHECK(gs_init(&dev, 1));
  CHECK(gs_set_device(0));
  CHECK(gs_g1_upload(g.a, g.m, &h[0])); CHECK(gs_g1_upload(g.ap, g.m, &h[1])); CHECK(gs_g2_upload(g.b, g.m, &h[2]));
  CHECK(gs_g1_upload(g.bp, g.m, &h[3])); CHECK(gs_g1_upload(g.c, g.m, &h[4])); CHECK(gs_g1_upload(g.cp, g.m, &h[5]));
  CHECK(gs_g1_upload(g.kp, g.m, &h[6])); CHECK(gs_g1_upload(g.g1t, g.ng1t, &h[7]));
  CHECK(gs_pinocchio_pk_create(h[0], h[1], h[2], h[3], h[4], h[5], h[6], h[7], g.z, g.nz, g.m, g.npublic, &pk));
  for (int i = 0; i < 8; ++i) CHECK(gs_free(h[i]));
  CHECK(gs_pinocchio_prove(pk, g.w, g.m, g.px, g.npx, out, inf));
  /* out = PiA | PiAp | PiB (16) | PiBp | PiC | PiCp | PiH | PiKp (affine) -> the verifier's Jacobian layout
     PiA, PiAp (12 each) | PiB (24) | PiBp, PiC, PiCp, PiH, PiKp (12 each) */
  memset(proof, 0, sizeof proof);
  {
    const int src[8] = {0, 8, 16, 32, 40, 48, 56, 64}, dst[8] = {0, 12, 24, 48, 60, 72, 84, 96};
    for (int k = 0; k < 8; ++k) {
      if (inf[k]) continue;
      if (k == 2) { memcpy(proof + dst[k], out + src[k], 128); proof[dst[k] + 16] = 1; }
      else { memcpy(proof + dst[k], out + src[k], 64); proof[dst[k] + 8] = 1; }
    }
  }
  CHECK(gs_pinocchio_verify(g.vka, g.vkb, g.vkc, g.g1kbg, g.g2kbg, g.g2kg, g.vkz, g.ic, g.nic, g.pub, g.nic - 1, proof, &ok, &failed));
  CHECK(gs_pinocchio_verify(g.vka, g.vkb, g.vkc, g.g1kbg, g.g2kbg, g.g2kg, g.vkz, g.ic, g.nic, wrong, g.nic - 1, proof, &ok2, &failed2));
  for (int i = 0; i < 8; ++i) out[72 + i] = (uint64_t)inf[i];
  out[80] = (uint64_t)ok; out[81] = (uint64_t)failed; out[82] = (uint64_t)ok2; out[83] = (uint64_t)failed2;
  if (write_words(argv[2], out, 84)) return 4;
  CHECK(gs_free(pk));
  gs_shutdown();
  printf("OK\n");
  return 0;
}
