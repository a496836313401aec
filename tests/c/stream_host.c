/* go/gosnarkhip/stream.go as C: the reference's two call shapes at speed (round 5).
 *  (1) a NEW witness (and px) in host memory with every call: host-buffer tickets gs_groth16_prove_host_begin /
 *      gs_groth16_prove_witness_host_begin (+ the blocking gs_groth16_prove_witness_host, + gs_scalars_update in place, + the
 *      Pinocchio twins); after the first lap over the three slots no hipMalloc / hipFree (gs_alloc_counters);
 *  (2) prove ONCE per key load: the default table policy sums a fresh key table-free; gs_build_tables warms it, gs_set_table_policy
 *      / gs_release_tables go back -- the proof never changes.
 * argv: r1cs file, groth16 instance, pinocchio instance, output (32 + 72 proof words, compared with the reference's proofs). */
#include "instance.h"

static int same(const uint64_t* a, const uint64_t* b, size_t words) { return memcmp(a, b, words * 8) == 0; }

int main(int argc, char** argv) {
  if (argc != 5) return 9;
  r1cs_instance q;
  groth_instance g;
  pinocchio_instance p;
  if (read_r1cs_instance(argv[1], &q) || read_groth_instance(argv[2], &g) || read_pinocchio_instance(argv[3], &p)) return 8;
  int dev = 0, inf[3], inf2[3], pinf[8], pinf2[8];
  gs_handle gk, pk, r1cs, w, h[8];
  uint64_t want[32], got[32], out[32 + 72], pwant[72], pgot[72], t[3], a0, f0, a1, f1, obj = 0, tab = 1;
  size_t tb = 0, mb = 0;
  CHECK(gs_init(&dev, 1));
  CHECK(gs_abi_sizes(&tb, &mb));
  if (tb != sizeof(gs_timing) || mb != sizeof(gs_memory)) { printf("FAIL: struct sizes %zu %zu\n", tb, mb); return 2; }
  if (upload_groth_pk(&g, &gk)) return 3;
  CHECK(gs_g1_upload(p.a, p.m, &h[0])); CHECK(gs_g1_upload(p.ap, p.m, &h[1])); CHECK(gs_g2_upload(p.b, p.m, &h[2]));
  CHECK(gs_g1_upload(p.bp, p.m, &h[3])); CHECK(gs_g1_upload(p.c, p.m, &h[4])); CHECK(gs_g1_upload(p.cp, p.m, &h[5]));
  CHECK(gs_g1_upload(p.kp, p.m, &h[6])); CHECK(gs_g1_upload(p.g1t, p.ng1t, &h[7]));
  CHECK(gs_pinocchio_pk_create(h[0], h[1], h[2], h[3], h[4], h[5], h[6], h[7], p.z, p.nz, p.m, p.npublic, &pk));
  for (int i = 0; i < 8; ++i) CHECK(gs_free(h[i]));
  CHECK(gs_r1cs_upload(q.n, q.m, q.rowptr[0], q.col[0], q.val[0], q.rowptr[1], q.col[1], q.val[1], q.rowptr[2], q.col[2], q.val[2], &r1cs));

  /* the FIRST proof of a fresh key (default policy: auto -> table-free, no tables afterwards), witness from the host, blocking */
  CHECK(gs_groth16_prove_witness_host(gk, r1cs, g.w, g.m, g.rs, g.rs + 4, want, inf));
  CHECK(gs_handle_bytes(gk, &obj, &tab));
  if (tab != 0) { printf("FAIL: a key's first proof built %llu bytes of tables under policy auto\n", (unsigned long long)tab); return 4; }
  /* host-buffer tickets, three in flight, two laps: w + px, w alone */
  for (int lap = 0; lap < 3; ++lap) {
    if (lap == 2) CHECK(gs_alloc_counters(&a0, &f0));
    CHECK(gs_groth16_prove_host_begin(gk, g.w, g.m, g.px, g.npx, g.rs, g.rs + 4, &t[0]));
    CHECK(gs_groth16_prove_witness_host_begin(gk, r1cs, g.w, g.m, g.rs, g.rs + 4, &t[1]));
    CHECK(gs_groth16_prove_host_begin(gk, g.w, g.m, g.px, g.npx, g.rs, g.rs + 4, &t[2]));
    if (gs_groth16_prove_witness_host_begin(gk, r1cs, g.w, g.m, g.rs, g.rs + 4, &t[0]) != GS_ERR_BUSY) { printf("FAIL: a fourth ticket\n"); return 5; }
    for (int i = 0; i < 3; ++i) {
      CHECK(gs_groth16_prove_end(t[i], got, inf2));
      if (!same(got, want, 32) || memcmp(inf, inf2, sizeof inf) != 0) { printf("FAIL: host ticket %d of lap %d\n", i, lap); return 6; }
    }
  }
  CHECK(gs_alloc_counters(&a1, &f1));
  if (a1 != a0 || f1 != f0) { printf("FAIL: %llu hipMalloc / %llu hipFree in a steady lap\n", (unsigned long long)(a1 - a0), (unsigned long long)(f1 - f0)); return 7; }
  /* in-place update of a resident witness while a ticket that reads it is outstanding */
  CHECK(gs_scalars_upload(g.w, g.m, &w));
  CHECK(gs_groth16_prove_witness_begin(gk, r1cs, w, g.rs, g.rs + 4, &t[0]));
  CHECK(gs_scalars_update(w, g.w, g.m));
  if (gs_scalars_update(w, g.w, g.m - 1) != GS_ERR_ARG) { printf("FAIL: short update accepted\n"); return 10; }
  CHECK(gs_groth16_prove_end(t[0], got, inf2));
  if (!same(got, want, 32)) { printf("FAIL: ticket across an update\n"); return 11; }
  /* warm the key: tables now, same proof; never: back to table-free, same proof */
  CHECK(gs_build_tables(gk, 0));
  CHECK(gs_handle_bytes(gk, NULL, &tab));
  if (tab < 4 * obj) { printf("FAIL: gs_build_tables left %llu table bytes\n", (unsigned long long)tab); return 12; }
  CHECK(gs_groth16_prove_witness_host(gk, r1cs, g.w, g.m, g.rs, g.rs + 4, got, inf2));
  if (!same(got, want, 32)) { printf("FAIL: proof on window tables\n"); return 13; }
  CHECK(gs_set_table_policy(2));
  CHECK(gs_release_tables(gk));
  CHECK(gs_groth16_prove(gk, g.w, g.m, g.px, g.npx, g.rs, g.rs + 4, got, inf2));
  CHECK(gs_handle_bytes(gk, NULL, &tab));
  if (!same(got, want, 32) || tab != 0) { printf("FAIL: policy never\n"); return 14; }
  CHECK(gs_set_table_policy(0));
  memcpy(out, want, sizeof want);

  /* snark.GenerateProofs: the same shapes */
  CHECK(gs_pinocchio_prove_witness_host(pk, r1cs, p.w, p.m, pwant, pinf));
  CHECK(gs_pinocchio_prove_host_begin(pk, p.w, p.m, p.px, p.npx, &t[0]));
  CHECK(gs_pinocchio_prove_witness_host_begin(pk, r1cs, p.w, p.m, &t[1]));
  for (int i = 0; i < 2; ++i) {
    CHECK(gs_pinocchio_prove_end(t[i], pgot, pinf2));
    if (!same(pgot, pwant, 72) || memcmp(pinf, pinf2, sizeof pinf) != 0) { printf("FAIL: pinocchio host ticket %d\n", i); return 15; }
  }
  CHECK(gs_build_tables(pk, 1));
  CHECK(gs_pinocchio_prove(pk, p.w, p.m, p.px, p.npx, pgot, pinf2));
  if (!same(pgot, pwant, 72)) { printf("FAIL: pinocchio on window tables\n"); return 16; }
  memcpy(out + 32, pwant, sizeof pwant);
  if (write_words(argv[4], out, 32 + 72)) return 17;
  CHECK(gs_free(w)); CHECK(gs_free(r1cs)); CHECK(gs_free(gk)); CHECK(gs_free(pk));
  gs_shutdown();
  printf("OK\n");
  return 0;
}
