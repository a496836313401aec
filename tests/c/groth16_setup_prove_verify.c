/* groth16hip.GenerateTrustedSetup -> GenerateProofs -> VerifyProof (go/groth16hip/groth16hip.go), as C:
 *   gs_groth16_setup (sparse R1CS + the five toxic values; groth16/groth16.go:94-222) -> gs_groth16_pk_export 0..6 (what
 *   (*Groth16Key).Export reads back into groth16.Setup.Pk) -> gs_groth16_prove on the RESIDENT key (no upload: the key the
 *   setup built stays cached) -> gs_groth16_verify with the vk the setup returned.
 * argv: r1cs file, instance file (w, px, r, s, public), output file:
 *   proof 32 | inf 3 | ok 1 | vk (84 + 12 (npublic + 1)) | At, BACGamma1 (m x 12 each) | BACGamma2 (m x 24) | BACDelta (m x 12) |
 *   PowersTauDelta (nz x 12) | singles 84 | Z (nz x 4) */
#include "instance.h"

int main(int argc, char** argv) {
  if (argc != 4) return 9;
  r1cs_instance r;
  groth_instance g;
  if (read_r1cs_instance(argv[1], &r) || read_groth_instance(argv[2], &g) || r.ntoxic != 5) return 8;
  const size_t m = r.m, nz = m - 1, nvk = 84 + 12 * (r.npublic + 1);
  const size_t total = 36 + nvk + m * 12 * 3 + m * 24 + nz * 12 + 84 + nz * 4;
  uint64_t* out = (uint64_t*)calloc(total, 8);
  uint64_t *vk = out + 36, *at = vk + nvk, *b1 = at + m * 12, *b2 = b1 + m * 12, *cd = b2 + m * 24, *pt = cd + m * 12, *single = pt + nz * 12,
           *z = single + 84;
  int dev = 0, inf[3], ok = 0;
  uint64_t jac[48];
  gs_handle pk;
  CHECK(gs_init(&dev, 1));
  CHECK(gs_set_device(0));
  CHECK(gs_groth16_setup(r.n, r.m, r.npublic, r.rowptr[0], r.col[0], r.val[0], r.rowptr[1], r.col[1], r.val[1], r.rowptr[2], r.col[2], r.val[2],
                         r.toxic, &pk, vk));
  CHECK(gs_groth16_pk_export(pk, 0, at, m)); CHECK(gs_groth16_pk_export(pk, 1, b1, m)); CHECK(gs_groth16_pk_export(pk, 2, b2, m));
  CHECK(gs_groth16_pk_export(pk, 3, cd, m)); CHECK(gs_groth16_pk_export(pk, 4, pt, nz)); CHECK(gs_groth16_pk_export(pk, 5, single, 5));
  CHECK(gs_groth16_pk_export(pk, 6, z, nz));
  CHECK(gs_groth16_prove(pk, g.w, g.m, g.px, g.npx, g.rs, g.rs + 4, out, inf));
  proof_to_jacobian(out, inf, jac);
  CHECK(gs_groth16_verify(vk, vk + 12, vk + 36, vk + 60, vk + 84, r.npublic + 1, g.pub, r.npublic, jac, jac + 12, jac + 36, &ok));
  for (int i = 0; i < 3; ++i) out[32 + i] = (uint64_t)inf[i];
  out[35] = (uint64_t)ok;
  if (write_words(argv[3], out, total)) return 4;
  CHECK(gs_free(pk));
  gs_shutdown();
  printf("OK\n");
  return 0;
}
