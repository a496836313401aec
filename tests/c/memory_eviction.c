/* gosnarkhip.MemoryOf / HandleBytes / ReleaseTables / Trim (go/gosnarkhip/memory.go), as C: a key's window tables are visible in
 * the accounting, releasing them gives the bytes back while a ticket that reads them is outstanding (the release queues behind it),
 * the next proof rebuilds them, trimming drops the workspaces -- and the proof never changes. */
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
  gs_shutdown();
  printf("OK\n");
  return 0;
}
