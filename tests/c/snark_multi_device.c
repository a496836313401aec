/* One Pinocchio proof over THREE logical devices (GPU 0 listed three times), from a plain C process -- the call sequences of
 * go/gosnarkhip/pinocchio_multi.go:
 *   PinocchioProveMulti   gs_pinocchio_pk_shard_to / gs_scalars_clone per device -> gs_comm_init_local -> gs_pinocchio_prove_multi
 *   ProvePartials + PinocchioCombine   gs_pinocchio_prove_partials per slice -> gs_pinocchio_combine  (what N processes exchange)
 *   PinocchioProveBatch   gs_pinocchio_prove_batch over full replicas
 * each == gs_pinocchio_prove on the full key (snark.go:254-289), which gs_pinocchio_verify accepts.
 * argv: instance file (c_util.write_pinocchio_instance). */
#include "instance.h"

int main(int argc, char** argv) {
  if (argc != 2) return 9;
  pinocchio_instance g;
  if (read_pinocchio_instance(argv[1], &g)) return 8;
  enum { N = 3 };
  int devs[N] = {0, 0, 0}, inf[8], inf2[8], inf3[8], pinf[N * 8], binf[N * 8], used = -1, nranks = 0, rank = -1, local = 0;
  uint64_t want[72], multi[72], combined[72], parts[N * 72], batch[N * 72], collectives = 0;
  gs_handle h[8], full, w0, px0, pk[N], rep[N], w[N], px[N];
  CHECK(gs_init(devs, N));
  CHECK(gs_set_device(0));
  CHECK(gs_g1_upload(g.a, g.m, &h[0])); CHECK(gs_g1_upload(g.ap, g.m, &h[1])); CHECK(gs_g2_upload(g.b, g.m, &h[2]));
  CHECK(gs_g1_upload(g.bp, g.m, &h[3])); CHECK(gs_g1_upload(g.c, g.m, &h[4])); CHECK(gs_g1_upload(g.cp, g.m, &h[5]));
  CHECK(gs_g1_upload(g.kp, g.m, &h[6])); CHECK(gs_g1_upload(g.g1t, g.ng1t, &h[7]));
  CHECK(gs_pinocchio_pk_create(h[0], h[1], h[2], h[3], h[4], h[5], h[6], h[7], g.z, g.nz, g.m, g.npublic, &full));
  for (int i = 0; i < 8; ++i) CHECK(gs_free(h[i]));
  CHECK(gs_scalars_upload(g.w, g.m, &w0));
  CHECK(gs_scalars_upload(g.px, g.npx, &px0));
  CHECK(gs_pinocchio_prove(full, g.w, g.m, g.px, g.npx, want, inf));
  for (int d = 0; d < N; ++d) {
    CHECK(gs_pinocchio_pk_shard_to(full, (size_t)d, N, d, &pk[d]));
    CHECK(gs_pinocchio_pk_shard_to(full, 0, 1, d, &rep[d]));
    CHECK(gs_scalars_clone(w0, 0, g.m, d, &w[d]));
    CHECK(gs_scalars_clone(px0, 0, g.npx, d, &px[d]));
    if (gs_handle_device(pk[d]) != d || gs_handle_device(w[d]) != d) { printf("FAIL: slice %d is not on its device\n", d); return 4; }
  }
  CHECK(gs_comm_init_local());
  CHECK(gs_pinocchio_prove_multi(pk, w, px, N, multi, inf2, &used));
  CHECK(gs_comm_info(&nranks, &rank, &local, &collectives));
  if (memcmp(want, multi, sizeof want) != 0 || memcmp(inf, inf2, sizeof inf) != 0) { printf("FAIL: the sharded proof differs\n"); return 5; }
  for (int d = 0; d < N; ++d) CHECK(gs_pinocchio_prove_partials(pk[d], w[d], px[d], (size_t)d, N, parts + 72 * d, pinf + 8 * d));
  CHECK(gs_pinocchio_combine(parts, pinf, N, combined, inf3));
  if (memcmp(want, combined, sizeof want) != 0 || memcmp(inf, inf3, sizeof inf) != 0) { printf("FAIL: the combined records differ\n"); return 6; }
  CHECK(gs_pinocchio_prove_batch(rep, N, w, px, N, batch, binf));
  for (int d = 0; d < N; ++d)
    if (memcmp(want, batch + 72 * d, sizeof want) != 0 || memcmp(inf, binf + 8 * d, sizeof inf) != 0) { printf("FAIL: batch proof %d differs\n", d); return 7; }
  {
    uint64_t proof[108];
    int ok = 0, failed = -1;
    const int src[8] = {0, 8, 16, 32, 40, 48, 56, 64}, dst[8] = {0, 12, 24, 48, 60, 72, 84, 96};
    memset(proof, 0, sizeof proof);
    for (int k = 0; k < 8; ++k) {
      if (inf2[k]) continue;
      if (k == 2) { memcpy(proof + dst[k], multi + src[k], 128); proof[dst[k] + 16] = 1; }
      else { memcpy(proof + dst[k], multi + src[k], 64); proof[dst[k] + 8] = 1; }
    }
    CHECK(gs_pinocchio_verify(g.vka, g.vkb, g.vkc, g.g1kbg, g.g2kbg, g.g2kg, g.vkz, g.ic, g.nic, g.pub, g.nic - 1, proof, &ok, &failed));
    if (!ok) { printf("FAIL: the verifier rejects the sharded proof (check %d)\n", failed); return 10; }
  }
  printf("used_rccl=%d nranks=%d local=%d collectives=%llu\n", used, nranks, local, (unsigned long long)collectives);
  gs_comm_destroy();
  gs_shutdown();
  printf("OK\n");
  return 0;
}
