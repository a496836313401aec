/* snarkhip.GenerateTrustedSetup -> GenerateProofs -> VerifyProof (go/snarkhip/snarkhip.go), as C:
 *   gs_pinocchio_setup (sparse R1CS + eight toxic values; snark.go:98-251) -> gs_pinocchio_pk_export 0..8 -> gs_pinocchio_prove on
 *   the resident key -> gs_pinocchio_verify with the vk the setup returned;
 * then the sharded forms of go/gosnarkhip/pinocchio_multi.go at ONE shard / ONE rank, each of which must reproduce that proof:
 *   (*PinocchioKey).Shard = gs_pinocchio_pk_shard; WitnessValues = gs_pinocchio_witness_values on the uploaded R1CS (the key comes
 *   from the device setup, so it carries its evaluation-basis array); ProvePartialsValues + PinocchioCombine; ProveSharded /
 *   ProveShardedValues = gs_comm_unique_id + gs_comm_init_rank(id, 1, 0) + gs_pinocchio_prove_sharded[_values].
 * argv: r1cs file, instance file (w, px, public), output file:
 *   proof 72 | inf 8 | ok, failed | vk (144 + 12 (npublic + 1)) | A, Ap (m x 12) | B (m x 24) | Bp, C, Cp, Kp (m x 12) | G1T (nz x 12) | Z (nz x 4) */
#include "instance.h"

int main(int argc, char** argv) {
  if (argc != 4) return 9;
  r1cs_instance r;
  pinocchio_instance g;
  if (read_r1cs_instance(argv[1], &r) || read_pinocchio_instance(argv[2], &g) || r.ntoxic != 8) return 8;
  const size_t m = r.m, nz = m - 1, nvk = 144 + 12 * (r.npublic + 1);
  const size_t total = 82 + nvk + m * 12 * 6 + m * 24 + nz * 12 + nz * 4;
  uint64_t* out = (uint64_t*)calloc(total, 8);
  uint64_t* vk = out + 82;
  uint64_t* arr[9];
  const size_t words[9] = {m * 12, m * 12, m * 24, m * 12, m * 12, m * 12, m * 12, nz * 12, nz * 4};
  const size_t count[9] = {m, m, m, m, m, m, m, nz, nz};
  arr[0] = vk + nvk;
  for (int k = 1; k < 9; ++k) arr[k] = arr[k - 1] + words[k - 1];
  int dev = 0, inf[8], ok = 0, failed = -1;
  uint64_t proof[108];
  gs_handle pk;
  CHECK(gs_init(&dev, 1));
  CHECK(gs_set_device(0));
  CHECK(gs_pinocchio_setup(r.n, r.m, r.npublic, r.rowptr[0], r.col[0], r.val[0], r.rowptr[1], r.col[1], r.val[1], r.rowptr[2], r.col[2], r.val[2],
                           r.toxic, &pk, vk));
  for (int k = 0; k < 9; ++k) CHECK(gs_pinocchio_pk_export(pk, k, arr[k], count[k]));
  CHECK(gs_pinocchio_prove(pk, g.w, g.m, g.px, g.npx, out, inf));
  memset(proof, 0, sizeof proof);
  {
    const int src[8] = {0, 8, 16, 32, 40, 48, 56, 64}, dst[8] = {0, 12, 24, 48, 60, 72, 84, 96};
    for (int k = 0; k < 8; ++k) {
      if (inf[k]) continue;
      if (k == 2) { memcpy(proof + dst[k], out + src[k], 128); proof[dst[k] + 16] = 1; }
      else { memcpy(proof + dst[k], out + src[k], 64); proof[dst[k] + 8] = 1; }
    }
  }
  CHECK(gs_pinocchio_verify(vk, vk + 24, vk + 36, vk + 60, vk + 72, vk + 96, vk + 120, vk + 144, r.npublic + 1, g.pub, r.npublic, proof, &ok, &failed));
  {
    gs_handle slice = 0, q = 0, wv = 0, hv = 0, pxh = 0;
    uint32_t bad = 1;
    uint64_t part[72], comb[72], sh[72], shv[72];
    int pinf[8], cinf[8], sinf[8], svinf[8];
    uint8_t id[128];
    CHECK(gs_pinocchio_pk_shard(pk, 0, 1, &slice));
    CHECK(gs_r1cs_upload(r.n, r.m, r.rowptr[0], r.col[0], r.val[0], r.rowptr[1], r.col[1], r.val[1], r.rowptr[2], r.col[2], r.val[2], &q));
    CHECK(gs_scalars_upload(g.w, g.m, &wv));
    CHECK(gs_scalars_upload(g.px, g.npx, &pxh));
    CHECK(gs_pinocchio_witness_values(slice, q, wv, &hv, &bad));
    if (bad != 0) { printf("FAIL: the fixture witness violates %u constraints\n", bad); return 5; }
    CHECK(gs_pinocchio_prove_partials_values(slice, wv, hv, 0, 1, part, pinf));
    CHECK(gs_pinocchio_combine(part, pinf, 1, comb, cinf));
    CHECK(gs_comm_unique_id(id));
    CHECK(gs_comm_init_rank(id, 1, 0));
    CHECK(gs_pinocchio_prove_sharded(slice, wv, pxh, sh, sinf));
    CHECK(gs_pinocchio_prove_sharded_values(slice, wv, hv, shv, svinf));
    gs_comm_destroy();
    if (memcmp(comb, out, 72 * 8) || memcmp(sh, out, 72 * 8) || memcmp(shv, out, 72 * 8) || memcmp(cinf, inf, sizeof inf) ||
        memcmp(sinf, inf, sizeof inf) || memcmp(svinf, inf, sizeof inf)) {
      printf("FAIL: a sharded form (values route / rank mode at one rank) differs from gs_pinocchio_prove\n");
      return 6;
    }
    CHECK(gs_free(hv)); CHECK(gs_free(pxh)); CHECK(gs_free(wv)); CHECK(gs_free(q)); CHECK(gs_free(slice));
  }
  for (int i = 0; i < 8; ++i) out[72 + i] = (uint64_t)inf[i];
  out[80] = (uint64_t)ok; out[81] = (uint64_t)failed;
  if (write_words(argv[3], out, total)) return 4;
  CHECK(gs_free(pk));
  gs_shutdown();
  printf("OK\n");
  return 0;
}
