/* One Groth16 proof over THREE logical devices (GPU 0 listed three times), from a plain C process:
 *   gs_init({0,0,0}) -> key on device 0 -> gs_groth16_pk_shard_to / gs_scalars_clone -> gs_comm_init_local ->
 *   gs_groth16_prove_multi == gs_groth16_prove on the full key.
 * The call sequence of go/groth16hip.GenerateProofsMulti. */
#include "instance.h"

int main(int argc, char** argv) {
  if (argc != 2) return 9;
  groth_instance g;
  if (read_groth_instance(argv[1], &g)) return 8;
  enum { N = 3 };
  int devs[N] = {0, 0, 0}, inf[3], inf2[3], ok = 0, used = -1, nranks = 0, rank = -1, local = 0;
  uint64_t proof[32], multi[32], jac[48], collectives = 0;
  gs_handle full, w0, px0, pk[N], w[N], px[N];
  CHECK(gs_init(devs, N));
  if (gs_device_count() != N) { printf("FAIL: %d logical devices\n", gs_device_count()); return 2; }
  CHECK(gs_set_device(0));
  if (upload_groth_pk(&g, &full)) return 3;
  CHECK(gs_scalars_upload(g.w, g.m, &w0));
  CHECK(gs_scalars_upload(g.px, g.npx, &px0));
  CHECK(gs_groth16_prove(full, g.w, g.m, g.px, g.npx, g.rs, g.rs + 4, proof, inf));
  for (int d = 0; d < N; ++d) {
    CHECK(gs_groth16_pk_shard_to(full, (size_t)d, N, d, &pk[d]));
    CHECK(gs_scalars_clone(w0, 0, g.m, d, &w[d]));
    CHECK(gs_scalars_clone(px0, 0, g.npx, d, &px[d]));
    if (gs_handle_device(pk[d]) != d || gs_handle_device(w[d]) != d) { printf("FAIL: slice %d is not on its device\n", d); return 4; }
  }
  CHECK(gs_comm_init_local());
  CHECK(gs_groth16_prove_multi(pk, w, px, N, g.rs, g.rs + 4, multi, inf2, &used));
  CHECK(gs_comm_info(&nranks, &rank, &local, &collectives));
  if (memcmp(proof, multi, sizeof proof) != 0 || memcmp(inf, inf2, sizeof inf) != 0) { printf("FAIL: the sharded proof differs\n"); return 5; }
  proof_to_jacobian(multi, inf2, jac);
  CHECK(gs_groth16_verify(g.vka, g.vk2, g.vk2 + 24, g.vk2 + 48, g.ic, g.nic, g.pub, g.nic - 1, jac, jac + 12, jac + 36, &ok));
  if (!ok) { printf("FAIL: the verifier rejects the sharded proof\n"); return 6; }
  printf("used_rccl=%d nranks=%d local=%d collectives=%llu\n", used, nranks, local, (unsigned long long)collectives);
  gs_comm_destroy();
  gs_shutdown();
  printf("OK\n");
  return 0;
}
