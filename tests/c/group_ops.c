/* go/bn128hip.G1 / G2 {MulScalar, Add, Double, Sub} and go/gosnarkhip {G1MulScalar, G1Add, G2MulScalar, G2Add, MSMG1Begin / End,
 * CancelTicket, MSMG1Resident, Pairing} as C: every group operation is a one- or two-term multi-scalar multiplication over an
 * uploaded point array (gs_g1_upload + gs_msm_g1).  Points: the first entries of the x^3 + x + 5 key (At in G1, G2.BACGamma in G2);
 * the Python side compares every result with the oracle's restatement of bn128/g1.go, g2.go (affine form).
 * argv: groth16 instance, output. */
#include "instance.h"

int main(int argc, char** argv) {
  if (argc != 3) return 9;
  groth_instance g;
  if (read_groth_instance(argv[1], &g)) return 8;
  int dev = 0, inf = 0;
  CHECK(gs_init(&dev, 1));
  uint64_t out[512];
  size_t pos = 0;
  memset(out, 0, sizeof out);           /* records: 8 (16) words of affine coordinates + 4 words holding the infinity flag */
  const uint64_t k[4] = {0x123456789abcdef1ull, 0x0fedcba987654321ull, 0x1111222233334444ull, 0x0123456789abcdefull};
  const uint64_t ones[8] = {1, 0, 0, 0, 1, 0, 0, 0};
  gs_handle h1, h2, hs, pp;
  /* G1: points P = At[2], Q = At[3] */
  CHECK(gs_g1_upload(g.at + 2 * 12, 2, &h1));
  CHECK(gs_msm_g1(h1, k, 0, 1, out + pos, &inf)); out[pos + 8] = (uint64_t)inf; pos += 12;          /* MulScalar(P, k) */
  CHECK(gs_msm_g1(h1, ones, 0, 2, out + pos, &inf)); out[pos + 8] = (uint64_t)inf; pos += 12;       /* Add(P, Q) */
  uint64_t twice[24];
  memcpy(twice, g.at + 2 * 12, 96); memcpy(twice + 12, g.at + 2 * 12, 96);
  CHECK(gs_g1_upload(twice, 2, &pp));
  CHECK(gs_msm_g1(pp, ones, 0, 2, out + pos, &inf)); out[pos + 8] = (uint64_t)inf; pos += 12;       /* Double(P) = Add(P, P): complete */
  CHECK(gs_free(pp));
  /* the same sum over resident operands, blocking and through a ticket; a second ticket is cancelled */
  CHECK(gs_scalars_upload(ones, 2, &hs));
  CHECK(gs_msm_g1_resident(h1, 0, hs, 0, 2, out + pos, &inf)); out[pos + 8] = (uint64_t)inf; pos += 12;
  uint64_t t1 = 0, t2 = 0;
  CHECK(gs_msm_g1_begin(h1, 0, hs, 0, 2, &t1));
  CHECK(gs_msm_g1_begin(h1, 0, hs, 0, 2, &t2));
  CHECK(gs_ticket_cancel(t2));
  if (gs_msm_end(t2, out + pos, &inf) == 0) { printf("FAIL: a cancelled ticket was collected\n"); return 5; }
  CHECK(gs_msm_end(t1, out + pos, &inf)); out[pos + 8] = (uint64_t)inf; pos += 12;
  /* G2: points of G2.BACGamma */
  CHECK(gs_g2_upload(g.b2 + 2 * 24, 2, &h2));
  CHECK(gs_msm_g2(h2, k, 0, 1, out + pos, &inf)); out[pos + 16] = (uint64_t)inf; pos += 20;        /* G2 MulScalar */
  CHECK(gs_msm_g2(h2, ones, 0, 2, out + pos, &inf)); out[pos + 16] = (uint64_t)inf; pos += 20;     /* G2 Add */
  /* bn128.Pairing(At[2], G2.BACGamma[2]) (host side) */
  CHECK(gs_pairing(g.at + 2 * 12, g.b2 + 2 * 24, out + pos)); pos += 48;
  if (write_words(argv[2], out, pos)) return 11;
  CHECK(gs_free(h1)); CHECK(gs_free(h2)); CHECK(gs_free(hs));
  gs_shutdown();
  printf("OK\n");
  return 0;
}
