/* groth16hip.GenerateProofsWithRS (go/groth16hip/groth16hip.go) = gosnarkhip.NewGroth16Key + (*Groth16Key).Prove, as C:
 *   5 x gs_g1_upload / gs_g2_upload -> gs_groth16_pk_create -> 5 x gs_free -> gs_groth16_prove (-> gs_groth16_verify, the
 *   sequence of groth16hip.VerifyProof).  Replaces groth16.GenerateProofs (groth16/groth16.go:225-278).
 * argv: instance file, output file (32 proof words | inf[3], verdict as words). */
#include "instance.h"

int main(int argc, char** argv) {
  if (argc != 3) return 9;
  groth_instance g;
  if (read_groth_instance(argv[1], &g)) return 8;
  int dev = 0, inf[3], ok = 0, bad = 1;
  gs_handle pk;
  uint64_t out[36], jac[48], wrong[4] = {34, 0, 0, 0};
  CHECK(gs_init(&dev, 1));
  CHECK(gs_set_device(0));
  if (upload_groth_pk(&g, &pk)) return 3;
  CHECK(gs_groth16_prove(pk, g.w, g.m, g.px, g.npx, g.rs, g.rs + 4, out, inf));
  proof_to_jacobian(out, inf, jac);
  CHECK(gs_groth16_verify(g.vka, g.vk2, g.vk2 + 24, g.vk2 + 48, g.ic, g.nic, g.pub, g.nic - 1, jac, jac + 12, jac + 36, &ok));
  CHECK(gs_groth16_verify(g.vka, g.vk2, g.vk2 + 24, g.vk2 + 48, g.ic, g.nic, wrong, g.nic - 1, jac, jac + 12, jac + 36, &bad));
  for (int i = 0; i < 3; ++i) out[32 + i] = (uint64_t)inf[i];
  out[35] = (uint64_t)(ok == 1 && bad == 0);
  if (write_words(argv[2], out, 36)) return 4;
  CHECK(gs_free(pk));
  gs_shutdown();
  printf("OK %s\n", gs_version());
  return 0;
}
