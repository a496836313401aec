/* gosnarkhip.MemoryOf / HandleBytes / ReleaseTables / Trim (go/gosnarkhip/memory.go) + SetTablePolicy / SetMemoryLimit
 * (go/gosnarkhip/stream.go), as C: a key's window tables are visible in the accounting, releasing them gives the bytes back while a
 * ticket that reads them is outstanding (the release queues behind it), the next proof rebuilds them, trimming drops the
 * workspaces -- and the proof never changes.  Round 5: out of memory is not fatal while idle tables exist -- under a cap
 * (gs_set_memory_limit) two keys that cannot both keep their tables prove round-robin, each evicting the other's, and
 * gs_memory.evictions counts it; a ticket's key is never the victim. */
#include "instance.h"

int main(int argc, char** argv) {
  if (argc != 2) return 9;
  groth_instance g;
  if (read_groth_instance(argv[1], &g)) return 8;
  int dev = 0, inf[3], inf2[3];
  gs_handle key, w, px;
  uint64_t t;
  uint64_t want[32], got[32], obj = 0, tab = 0, tab2 = 1;
  gs_memory m0, m1, m2;
  CHECK(gs_init(&dev, 1));
  CHECK(gs_set_table_policy(1));                                      /* tables inside the first call that needs them (rounds 1-4) */
  if (upload_groth_pk(&g, &key)) return 3;
  CHECK(gs_scalars_upload(g.w, g.m, &w));
  CHECK(gs_scalars_upload(g.px, g.npx, &px));
  CHECK(gs_groth16_prove_resident(key, w, px, g.rs, g.rs + 4, want, inf));
  CHECK(gs_handle_bytes(key, &obj, &tab));
  CHECK(gs_groth16_prove_begin(key, w, px, g.rs, g.rs + 4, &t));      /* this ticket reads the tables */
  CHECK(gs_memory_query(&m0));
  if (!obj || tab < 4 * obj || m0.table_bytes < tab || m0.library_bytes < m0.table_bytes + m0.object_bytes || m0.objects < 3) { printf("FAIL: accounting\n"); return 4; }
  CHECK(gs_release_tables(key));
  CHECK(gs_handle_bytes(key, NULL, &tab2));
  CHECK(gs_memory_query(&m1));
  if (tab2 != 0 || m1.table_bytes != m0.table_bytes - tab || m1.library_bytes + tab > m0.library_bytes) { printf("FAIL: release\n"); return 5; }
  CHECK(gs_groth16_prove_end(t, got, inf2));
  if (memcmp(got, want, sizeof got) != 0 || memcmp(inf, inf2, sizeof inf) != 0) { printf("FAIL: ticket across release\n"); return 6; }
  CHECK(gs_groth16_prove_resident(key, w, px, g.rs, g.rs + 4, got, inf2));
  CHECK(gs_handle_bytes(key, NULL, &tab2));
  if (memcmp(got, want, sizeof got) != 0 || tab2 != tab) { printf("FAIL: rebuild\n"); return 7; }
  CHECK(gs_trim());
  CHECK(gs_memory_query(&m2));
  if (m2.workspace_bytes != 0 || m2.table_bytes != m0.table_bytes) { printf("FAIL: trim\n"); return 10; }
  CHECK(gs_groth16_prove_resident(key, w, px, g.rs, g.rs + 4, got, inf2));
  if (memcmp(got, want, sizeof got) != 0) { printf("FAIL: proof after trim\n"); return 11; }
  /* --- eviction instead of failure ------------------------------------------------------------------------------------------ */
  {
    gs_handle key2;
    gs_memory m3, m4;
    uint64_t tabA = 0, tabB = 0;
    if (upload_groth_pk(&g, &key2)) return 12;
    /* (gs_trim above dropped the table builder's scratch slab with the other workspaces: one rebuild brings it back, so that the cap
     *  below is about TABLES) */
    CHECK(gs_release_tables(key));
    CHECK(gs_groth16_prove_resident(key, w, px, g.rs, g.rs + 4, got, inf2));
    CHECK(gs_groth16_prove_begin(key, w, px, g.rs, g.rs + 4, &t));     /* ... and a ticket slot's workspaces */
    CHECK(gs_groth16_prove_end(t, got, inf2));
    CHECK(gs_memory_query(&m3));
    /* room for an eighth of a second set of tables (and the few hundred bytes key2's first quotient caches): key2's first proof must
     * drop key's tables (idle, least recently used), one per allocation that does not fit */
    const uint64_t slack = tab / 8;
    CHECK(gs_set_memory_limit(m3.library_bytes + slack));
    for (int round = 0; round < 3; ++round) {
      printf("eviction round %d: key2 (library %llu bytes, cap %llu)\n", round, (unsigned long long)m3.library_bytes, (unsigned long long)(m3.library_bytes + slack));
      CHECK(gs_groth16_prove_resident(key2, w, px, g.rs, g.rs + 4, got, inf2));
      CHECK(gs_handle_bytes(key, NULL, &tabA)); CHECK(gs_handle_bytes(key2, NULL, &tabB));
      if (memcmp(got, want, sizeof got) != 0 || tabA + tabB > tab + slack || tabB != tab) { printf("FAIL: eviction round %d (key2): %llu %llu\n", round, (unsigned long long)tabA, (unsigned long long)tabB); return 13; }
      printf("eviction round %d: key\n", round);
      CHECK(gs_groth16_prove_resident(key, w, px, g.rs, g.rs + 4, got, inf2));
      CHECK(gs_handle_bytes(key, NULL, &tabA)); CHECK(gs_handle_bytes(key2, NULL, &tabB));
      if (memcmp(got, want, sizeof got) != 0 || tabA != tab || tabA + tabB > tab + slack) { printf("FAIL: eviction round %d (key)\n", round); return 14; }
    }
    CHECK(gs_memory_query(&m4));
    if (m4.evictions < 6 * 3 || m4.library_bytes > m3.library_bytes + slack) { printf("FAIL: %llu evictions\n", (unsigned long long)m4.evictions); return 15; }
    /* a ticket holds key: proving with key2 must NOT take key's tables (nothing else to evict -> the call fails cleanly, the ticket survives) */
    CHECK(gs_groth16_prove_begin(key, w, px, g.rs, g.rs + 4, &t));
    if (gs_groth16_prove_resident(key2, w, px, g.rs, g.rs + 4, got, inf2) != GS_ERR_HIP) { printf("FAIL: a held key was evicted\n"); return 16; }
    CHECK(gs_groth16_prove_end(t, got, inf2));
    if (memcmp(got, want, sizeof got) != 0) { printf("FAIL: ticket under memory pressure\n"); return 17; }
    /* policy auto: key2 has no tables -> it is summed table-free (per-window bucket sets: a few MiB of workspace more), same proof */
    printf("policy auto under a cap\n");
    CHECK(gs_set_table_policy(0));
    CHECK(gs_set_memory_limit(m4.library_bytes + (64u << 20)));
    CHECK(gs_groth16_prove_resident(key2, w, px, g.rs, g.rs + 4, got, inf2));
    if (memcmp(got, want, sizeof got) != 0) { printf("FAIL: table-free under the cap\n"); return 18; }
    CHECK(gs_set_memory_limit(0));
    CHECK(gs_free(key2));
  }
  gs_shutdown();
  printf("OK\n");
  return 0;
}
