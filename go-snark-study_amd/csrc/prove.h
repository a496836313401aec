// Internal: device-resident proving keys and the prover drivers (implemented in prove.hip).
#pragma once
#include <vector>
#include <mutex>
#include "msm.h"
#include "poly.h"
#include "runtime.h"

namespace gs {

struct GrothPkObj : Object {      // groth16.Pk (groth16/groth16.go:15-32), resident
  size_t nvars = 0, npublic = 0, nz = 0, nptd = 0;   // global counts (nptd = len(PowersTauDelta))
  // Key slices (multi-GPU, SURVEY 8e "each GPU holds 1/8 of every pk array"): a key created by gs_groth16_pk_create_shard /
  // gs_groth16_pk_shard holds only the term ranges of shard `shard_index` of `shard_count`: At / BACGamma / BACDelta entries
  // [w_lo, w_lo + n_w) and PowersTauDelta entries [h_lo, h_lo + n_h).  A full key has shard_count = 1, n_w = nvars, n_h = nptd.
  size_t shard_index = 0, shard_count = 1, w_lo = 0, n_w = 0, h_lo = 0, n_h = 0;
  DevBuf at, bacgamma1, bacdelta, ptd;     // packed affine G1 (owned copies)
  DevBuf bacgamma2;                        // packed affine G2
  BaseTable t_at, t_bacgamma1, t_bacdelta, t_ptd, t_bacgamma2;   // their window tables (built on the first prove)
  G1Affine alpha, beta, delta;             // host, Montgomery
  G2Affine beta2, delta2;
  // host-side window tables of delta / delta2 for the tail's result-independent products (built on the first proof of the key)
  std::once_flag fixed_once;
  HostFixedBase<FqTag> delta_fixed;
  HostFixedBase<Fq2Tag> delta2_fixed;
  Divisor z;                               // pk.Z with cached 1/rev(Z) series + spectrum
  // Evaluation-basis copy of PowersTauDelta (optional; gs_groth16_setup builds it, gs_groth16_pk_set_eval attaches one):
  //   ptd_eval[j-1] = l_j(tau) * Z(tau) / delta * G,  l_j = Lagrange basis over the nodes n+1 .. 2n  (j = 1..n_eval = #constraints)
  // so that  sum_j H(n+j) ptd_eval[j-1] = H(tau) Z(tau) / delta * G = sum_i h_i PowersTauDelta[i]  (groth16.go:139-149, 269-271):
  // the witness route runs the h-MSM over H's VALUES and never interpolates H.  A slice holds entries [e_lo, e_lo + n_e).
  size_t n_eval = 0, e_lo = 0, n_e = 0;
  DevBuf ptd_eval;
  BaseTable t_ptd_eval;
  // Which of the held variables appear in B at all (round 5).  The reference's circuit compiler puts a variable into B only as the
  // second operand of a multiplication or a divisor (circuitcompiler/circuit.go:110-128: `+` / `-` / `in` rows have B = [one]), so for
  // its circuits most G1.BACGamma / G2.BACGamma points are the point at infinity.  b_index lists the held variables of which either
  // point is finite (ascending, relative to the first held variable; on the device and on the host, where a call cuts its term range out
  // of it), b_finite is their number: when enough are missing the prover sums B1 and B2 -- 3.8 of a proof's 6.8 job-units -- over a
  // SECOND plan of w that holds the listed terms only (prove.hip, groth16_enqueue).  Scanned once, when the key is created; keys
  // without a missing point keep no list.
  DevBuf b_index;
  std::vector<uint32_t> b_index_host;
  size_t b_finite = 0;
  GrothPkObj() : Object(Kind::GrothPk) {}
};
void groth_pk_scan_sparsity(Ctx& c, GrothPkObj& pk);      // fills b_index / b_finite (synchronises the stream)

struct PinocchioPkObj : Object {  // snark.Pk (snark.go:16-26), resident
  size_t nvars = 0, npublic = 0, nz = 0, ng1t = 0;      // global counts (ng1t = len(G1T))
  // key slices as GrothPkObj's: the seven per-variable arrays hold entries [w_lo, w_lo + n_w), G1T entries [h_lo, h_lo + n_h)
  size_t shard_index = 0, shard_count = 1, w_lo = 0, n_w = 0, h_lo = 0, n_h = 0;
  DevBuf a, ap, bp, c, cp, kp, g1t;        // packed affine G1
  DevBuf b2;                               // packed affine G2
  BaseTable t_a, t_ap, t_bp, t_c, t_cp, t_kp, t_g1t, t_b2;
  Divisor z;
  // evaluation-basis copy of G1T (optional, as GrothPkObj::ptd_eval): g1t_eval[j-1] = l_j(tau) * G over the nodes n+1 .. 2n, so that
  // sum_j H(n+j) g1t_eval[j-1] = H(tau) G = sum_i h_i G1T[i]  (snark.go:239-247, 284-286)
  size_t n_eval = 0, e_lo = 0, n_e = 0;     // a slice holds entries [e_lo, e_lo + n_e)
  DevBuf g1t_eval;
  BaseTable t_g1t_eval;
  // which held variables appear in B (as GrothPkObj::b_index): B (G2) and B' (G1) of a reference-style circuit are mostly infinity
  DevBuf b_index;
  std::vector<uint32_t> b_index_host;
  size_t b_finite = 0;
  PinocchioPkObj() : Object(Kind::PinocchioPk) {}
};
void pinocchio_pk_scan_sparsity(Ctx& c, PinocchioPkObj& pk);

}  // namespace gs
