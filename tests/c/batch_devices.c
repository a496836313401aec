/* gosnarkhip.ProveBatch (go/gosnarkhip/multi.go), as C: two logical devices (GPU 0 twice), one full key each
 * (gs_groth16_pk_shard_to with shard 0 of 1 = a replica), five proofs with their own (r, s) round-robin over the devices through
 * gs_groth16_prove_batch; every proof equals the blocking gs_groth16_prove_resident.  BASELINE configs[4] in miniature. */
#include "instance.h"

int main(int argc, char** argv) {
  if (argc != 2) return 9;
  groth_instance g;
  if (read_groth_instance(argv[1], &g)) return 8;
  enum { N = 2, P = 5 };
  int devs[N] = {0, 0}, inf[3 * P], inf1[3];
  gs_handle key[N], w0, px0, w[P], px[P];
  uint64_t r[4 * P], s[4 * P], proofs[32 * P], one[32];
  CHECK(gs_init(devs, N));
  CHECK(gs_set_device(0));
  if (upload_groth_pk(&g, &key[0])) return 3;
  CHECK(gs_groth16_pk_shard_to(key[0], 0, 1, 1, &key[1]));
  CHECK(gs_scalars_upload(g.w, g.m, &w0));
  CHECK(gs_scalars_upload(g.px, g.npx, &px0));
  memset(r, 0, sizeof r); memset(s, 0, sizeof s);
  for (int i = 0; i < P; ++i) {
    r[4 * i] = g.rs[0] + 1000003u * (uint64_t)i; r[4 * i + 1] = g.rs[1];
    s[4 * i] = g.rs[4] ^ (0x9E3779B97F4A7C15ull * (uint64_t)(i + 1)); s[4 * i + 2] = (uint64_t)i;
    CHECK(gs_scalars_clone(w0, 0, g.m, i % N, &w[i]));
    CHECK(gs_scalars_clone(px0, 0, g.npx, i % N, &px[i]));
  }
  CHECK(gs_groth16_prove_batch(key, N, w, px, P, r, s, proofs, inf));
  for (int i = 0; i < P; ++i) {
    CHECK(gs_groth16_prove_resident(key[0], w0, px0, r + 4 * i, s + 4 * i, one, inf1));
    if (memcmp(one, proofs + 32 * i, sizeof one) != 0 || memcmp(inf1, inf + 3 * i, sizeof inf1) != 0) { printf("FAIL: proof %d differs\n", i); return 5; }
  }
  gs_shutdown();
  printf("OK\n");
  return 0;
}
