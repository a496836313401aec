/* gosnark_hip.h -- C ABI of libgosnark_hip.so: the MI355X (gfx950) prover hot path of
 * arnaucube/go-snark-study.
 *
 * The reference has no FFI or plugin seam (pure Go on math/big); the drop-in boundary is the
 * pair of Go functions
 *     groth16.GenerateProofs(circuit, pk, w, px) (Proof, error)     groth16/groth16.go:225-278
 *     snark.GenerateProofs  (circuit, pk, w, px) (Proof, error)     snark.go:254-289
 * plus the finer seams they are built from (bn128/g1.go:140 MulScalar + :32 Add loops,
 * bn128/g2.go:142/:32, r1csqap/r1csqap.go:57-216).  A cgo binding (go/gosnarkhip, shown in
 * INTEGRATION.md) flattens the Go structs and calls the entry points below; each entry point
 * cites the reference code it replaces.
 *
 * Conventions
 *  - Field elements (Fq and Fr): 4 x uint64_t little-endian limbs, STANDARD (non-Montgomery)
 *    form, i.e. exactly big.Int.Bits() of the reference value padded to 4 words.  Scalars and
 *    polynomial coefficients may be any value < 2^256; they are reduced mod r on the device.
 *  - G1 points: Jacobian triples [X, Y, Z] = 12 words, as the reference stores them
 *    ([3]*big.Int, bn128/g1.go:9-12).  G2 points: [[X0,X1],[Y0,Y1],[Z0,Z1]] = 24 words
 *    ([3][2]*big.Int, bn128/g2.go:9-12).  Z == 0 means infinity (g1.go:28-30).
 *  - Results are returned in AFFINE normal form [x, y, 1] (g1.go:157-170 / g2.go:183-200) with
 *    an is-infinity flag; the reference's raw Jacobian Z depends on its exact double/add order
 *    and cannot be reproduced by a bucket method.  Infinity is [0, 0, 0].
 *  - Every function returns 0 on success or a negative gs_status; gs_last_error() gives the
 *    message (thread-local).  Nothing falls back to a CPU path: without a usable gfx950
 *    device every prover/setup/polynomial/MSM call fails with GS_ERR_NO_DEVICE.  (The verifier
 *    entry points at the end are host code by design -- O(1) pairings -- and say so.)
 *  - The library copies caller buffers during the call and never retains host pointers
 *    (cgo pointer rule).  Device memory lives behind opaque handles freed by gs_free().
 *  - One context per LOGICAL DEVICE: gs_init takes a list of HIP ordinals and creates one context -- streams, workspaces,
 *    handle table, lock -- per entry (the same ordinal may appear several times).  A handle or ticket carries its logical
 *    device in its top byte, so every entry point that takes one runs on the right device whatever thread calls it; entry
 *    points that only CREATE objects (uploads, fixed-base batches, setups, the gs_poly_* family) use the calling thread's current
 *    logical device (gs_set_device, default 0; a Go caller wraps the pair in runtime.LockOSThread).  Calls on one logical
 *    device are serialised on its lock and are safe from any number of goroutines/threads; different logical devices run
 *    concurrently.  While pipelined tickets (gs_*_begin) are outstanding, every other call still works: uploads, downloads,
 *    gs_free (deferred until the tickets that read the object are collected) and gs_r1cs_px proceed at once, the rest
 *    first wait for the outstanding device work -- they queue, they do not fail.
 */
#ifndef GOSNARK_HIP_H
#define GOSNARK_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif
#if defined(__GNUC__)
#pragma GCC visibility push(default)
#endif

typedef uint64_t gs_handle;

typedef enum {
  GS_OK = 0,
  GS_ERR_NO_DEVICE = -1,   /* no gfx950 device / HIP runtime unusable */
  GS_ERR_HIP = -2,         /* a HIP call or kernel failed */
  GS_ERR_ARG = -3,         /* bad argument (null pointer, size mismatch, bad handle) */
  GS_ERR_SHAPE = -4,       /* instance violates the reference's shape contract */
  GS_ERR_NOT_INIT = -5,
  GS_ERR_BUSY = -6         /* gs_*_begin: all three in-flight slots of this logical device are taken -- collect one with gs_*_end and retry */
} gs_status;

/* ---- lifetime ------------------------------------------------------------------------- */
/* Create one context per entry of `devices` (HIP ordinals, 1 <= ndev <= 64): "logical device" i drives devices[i].  One
 * process per GPU passes one entry; a single Go process that drives a whole node passes all eight ("one stream +
 * context per (goroutine, device)"); listing one ordinal N times gives N logical devices that time-slice that GPU (how the
 * multi-device entry points below are tested on a 1-GPU box).  Calling it again with the same list is a no-op; a different
 * list needs gs_shutdown first.  gs_shutdown releases everything, on every device. */
int gs_init(const int* devices, int ndev);
void gs_shutdown(void);
/* Number of logical devices; the calling thread's current one (objects are created there); the device a handle/ticket lives on. */
int gs_device_count(void);
int gs_set_device(int logical_device);
int gs_get_device(void);
int gs_handle_device(gs_handle h);
const char* gs_last_error(void);
/* ABI/version string, e.g. "gosnark-hip 0.1 gfx950". */
const char* gs_version(void);
int gs_free(gs_handle h);

/* ---- resident base-point arrays (proving-key material) --------------------------------- */
/* Upload n Jacobian points, convert to Montgomery affine on the device (x = X/Z^2, y = Y/Z^3,
 * bn128/g1.go:157-170), keep them resident.  Replaces the per-call big.Int traffic of
 * pk.G1.At / pk.G1.BACGamma / pk.BACDelta / pk.PowersTauDelta (groth16.go:15-32).
 * Every finite point must satisfy its curve equation (y^2 = x^3 + 3; on the twist y^2 = x^3 + 3/(9+u)): otherwise
 * GS_ERR_ARG, naming the first offender (the reference never checks and would sum garbage). */
int gs_g1_upload(const uint64_t* jacobian /* n x 12 */, size_t n, gs_handle* out);
/* Same for G2 arrays, e.g. pk.G2.BACGamma (groth16.go:29), snark Pk.B (snark.go:19). */
int gs_g2_upload(const uint64_t* jacobian /* n x 24 */, size_t n, gs_handle* out);
/* Number of points behind a base handle, coefficients behind a scalar handle, or At entries a Groth16 key (slice) holds. */
int gs_len(gs_handle h, size_t* out);
/* Read points back as affine Jacobian triples [x, y, 1] / [0,0,0] (testing / serialisation). */
int gs_g1_download(gs_handle bases, uint64_t* jacobian /* n x 12 */, size_t n);
int gs_g2_download(gs_handle bases, uint64_t* jacobian /* n x 24 */, size_t n);

/* Fixed-base batch: out[i] = k[i] * G  for the G1 / G2 generator (bn128.go:52-83); the hot
 * loop of groth16.GenerateTrustedSetup (groth16.go:139-175, `MulScalar(Utils.Bn.G1.G, ...)`)
 * and what the synthetic proving keys of bench.py are built with. */
int gs_g1_fixed_base(const uint64_t* scalars /* n x 4 */, size_t n, gs_handle* out);
int gs_g2_fixed_base(const uint64_t* scalars /* n x 4 */, size_t n, gs_handle* out);

/* ---- resident scalar vectors ------------------------------------------------------------- */
int gs_scalars_upload(const uint64_t* scalars /* n x 4 */, size_t n, gs_handle* out);
int gs_scalars_download(gs_handle scalars, uint64_t* out /* n x 4 */, size_t n);
/* Overwrite a resident vector IN PLACE with n = its length new scalars (a server's next witness into the handle of an earlier one):
 * no hipMalloc, no hipFree -- gs_scalars_upload + gs_free per proof costs both, and hipFree synchronises the whole device under the
 * outstanding tickets.  The copy is ordered behind every device read of the vector that pipelined operations enqueued before the
 * call and has landed when the call returns.  With four vectors rotating under three tickets it never waits.
 * One exception to "a ticket has read its inputs when the update returns": a WITNESS ticket on a key with an evaluation-basis array
 * (gs_*_prove_witness_begin) whose witness violates a constraint is proved again, on the exact route, when it is COLLECTED -- from the
 * resident vector as it is then.  A caller that may submit unsatisfying witnesses must not update a vector before the ticket that
 * reads it has been collected (host-buffer tickets own their copy and are not affected). */
int gs_scalars_update(gs_handle scalars, const uint64_t* values /* n x 4 */, size_t n);

/* Copies of [off, off + n) of a resident vector / base array onto another logical device (device-to-device; across xGMI
 * when the two are different GPUs).  How the shards of a term range are handed to the devices that will sum them. */
int gs_scalars_clone(gs_handle scalars, size_t off, size_t n, int target_device, gs_handle* out);
int gs_g1_clone(gs_handle bases, size_t off, size_t n, int target_device, gs_handle* out);
int gs_g2_clone(gs_handle bases, size_t off, size_t n, int target_device, gs_handle* out);

/* ---- multi-scalar multiplication ----------------------------------------------------------
 * out = sum_{i<n} scalars[i] * bases[off + i]: replaces the loop
 *     acc = G1.Add(acc, G1.MulScalar(bases[i], scalars[i]))
 * of groth16.go:243-250,269-271 / snark.go:265-286 (bn128/g1.go:140-155 + :32-89) by a signed-
 * digit Pippenger bucket method.  out_affine = [x, y] (8 words); *is_inf set when the sum is
 * the identity (then out is all zero). */
int gs_msm_g1(gs_handle bases, const uint64_t* scalars /* n x 4 */, size_t off, size_t n,
              uint64_t out_affine[8], int* is_inf);
/* G2 flavour (bn128/g2.go:142-181 + :32-89); out_affine = [x0, x1, y0, y1] (16 words). */
int gs_msm_g2(gs_handle bases, const uint64_t* scalars /* n x 4 */, size_t off, size_t n,
              uint64_t out_affine[16], int* is_inf);
/* Same with the scalars already resident (handle from gs_scalars_upload / a poly result);
 * soff = first scalar used.  This is what bench.py times. */
int gs_msm_g1_resident(gs_handle bases, size_t off, gs_handle scalars, size_t soff, size_t n,
                       uint64_t out_affine[8], int* is_inf);
int gs_msm_g2_resident(gs_handle bases, size_t off, gs_handle scalars, size_t soff, size_t n,
                       uint64_t out_affine[16], int* is_inf);
/* Pipelined form (everything resident): begin enqueues one MSM and returns a ticket, gs_msm_end waits for that MSM only and
 * writes the affine result (8 words for a G1 ticket, 16 for G2).  Three operations may be outstanding (MSM or proof tickets);
 * the sort of MSM k+1 runs under the accumulation of MSM k and its accumulation starts the moment that one ends. */
int gs_msm_g1_begin(gs_handle bases, size_t off, gs_handle scalars, size_t soff, size_t n, uint64_t* ticket);
int gs_msm_g2_begin(gs_handle bases, size_t off, gs_handle scalars, size_t soff, size_t n, uint64_t* ticket);
int gs_msm_end(uint64_t ticket, uint64_t* out_affine, int* is_inf);
/* Sum of n points given as affine [x,y] pairs with infinity flags (multi-GPU combine of the
 * per-rank partial sums): out = sum_i pts[i]. */
int gs_g1_sum_affine(const uint64_t* pts /* n x 8 */, const int* inf, size_t n, uint64_t out_affine[8], int* is_inf);
int gs_g2_sum_affine(const uint64_t* pts /* n x 16 */, const int* inf, size_t n, uint64_t out_affine[16], int* is_inf);

/* ---- polynomial field over Fr (r1csqap/r1csqap.go) ---------------------------------------- */
/* PolynomialField.Mul (r1csqap.go:57-67): out has na + nb - 1 coefficients. */
int gs_poly_mul(const uint64_t* a, size_t na, const uint64_t* b, size_t nb, uint64_t* out);
/* PolynomialField.Div (r1csqap.go:70-84): quotient (na - nb + 1 coefficients) and, if rem is
 * not NULL, remainder (nb - 1 coefficients).  Requires na >= nb >= 1 and b[nb-1] != 0. */
int gs_poly_div(const uint64_t* a, size_t na, const uint64_t* b, size_t nb, uint64_t* quo, uint64_t* rem);
/* PolynomialField.Add / Sub (r1csqap.go:94-115): out has max(na, nb) coefficients. */
int gs_poly_add(const uint64_t* a, size_t na, const uint64_t* b, size_t nb, uint64_t* out);
int gs_poly_sub(const uint64_t* a, size_t na, const uint64_t* b, size_t nb, uint64_t* out);
/* PolynomialField.Eval (r1csqap.go:118-126). */
int gs_poly_eval(const uint64_t* v, size_t n, const uint64_t x[4], uint64_t out[4]);
/* PolynomialField.LagrangeInterpolation (r1csqap.go:150-158): n values at nodes 1..n ->
 * n coefficients.  Mathematically exact for every n (the reference's NewPolZeroAt overflows a
 * Go int for n >= 22, r1csqap.go:130-136; equal results for n <= 21). */
int gs_lagrange_interpolation(const uint64_t* values, size_t n, uint64_t* coeffs);
/* Z(x) = prod_{i=1}^{deg} (x - i): deg + 1 coefficients (r1csqap.go:177-186, groth16.go:122-131). */
int gs_zpoly(size_t deg, uint64_t* out);
/* Sparse R1CS -> P(x): the scalable replacement of R1CSToQAP + CombinePolynomials
 * (r1csqap.go:161-210).  A, B, C in CSR over n constraints x m variables (row_ptr n+1,
 * col idx, val nnz x 4); w: m scalars.  Outputs ax, bx, cx (n coeffs each, may be NULL) and
 * px = ax*bx - cx (2n - 1 coeffs). */
int gs_r1cs_to_px(size_t n, size_t m,
                  const uint32_t* a_rowptr, const uint32_t* a_col, const uint64_t* a_val,
                  const uint32_t* b_rowptr, const uint32_t* b_col, const uint64_t* b_val,
                  const uint32_t* c_rowptr, const uint32_t* c_col, const uint64_t* c_val,
                  const uint64_t* w, uint64_t* ax, uint64_t* bx, uint64_t* cx, uint64_t* px);

/* The same split per circuit / per proof: gs_r1cs_upload validates and keeps A, B, C resident (free with gs_free);
 * gs_r1cs_px turns a resident witness (gs_scalars_upload, m elements) into the resident px (2n - 1 coefficients: pass
 * *px_inout = 0 to create the vector, or the handle of an earlier call to overwrite it) -- nothing crosses PCIe, and the
 * result feeds gs_groth16_prove_resident / gs_groth16_prove_begin directly.  gs_last_timing().poly_ms = device time. */
int gs_r1cs_upload(size_t n, size_t m,
                   const uint32_t* a_rowptr, const uint32_t* a_col, const uint64_t* a_val,
                   const uint32_t* b_rowptr, const uint32_t* b_col, const uint64_t* b_val,
                   const uint32_t* c_rowptr, const uint32_t* c_col, const uint64_t* c_val, gs_handle* out);
int gs_r1cs_px(gs_handle r1cs, gs_handle w, gs_handle* px_inout);

/* ---- Groth16 prover (groth16/groth16.go) --------------------------------------------------- */
/* Device-resident proving key: groth16.Pk (groth16.go:15-32).  At, BACGamma (G1), BACDelta:
 * m points; G2 BACGamma: m points; PowersTauDelta: len(Z) points; single points as Jacobian
 * triples; Z: nz coefficients (monic).  npublic = circuit.NPublic. */
int gs_groth16_pk_create(gs_handle g1_at, gs_handle g1_bacgamma, gs_handle g2_bacgamma,
                         gs_handle bacdelta, gs_handle powers_tau_delta,
                         const uint64_t g1_alpha[12], const uint64_t g1_beta[12], const uint64_t g1_delta[12],
                         const uint64_t g2_beta[24], const uint64_t g2_delta[24],
                         const uint64_t* z, size_t nz, size_t nvars, size_t npublic, gs_handle* out);
/* groth16.GenerateProofs (groth16.go:225-278) with the randomness injected: r, s are what
 * Utils.FqR.Rand() would have returned (:231-238).  out_proof = PiA [x,y] (8 words) |
 * PiB [x0,x1,y0,y1] (16) | PiC [x,y] (8); inf[3] flags.  Requires len(w) == nvars and
 * npx - nz + 1 <= len(PowersTauDelta) (the reference indexes PowersTauDelta[i] for
 * i < len(hx), :269-271). */
int gs_groth16_prove(gs_handle pk, const uint64_t* w, size_t nw, const uint64_t* px, size_t npx,
                     const uint64_t r[4], const uint64_t s[4], uint64_t out_proof[32], int inf[3]);
/* Same with w and px already resident (gs_scalars_upload); what bench.py times. */
int gs_groth16_prove_resident(gs_handle pk, gs_handle w, gs_handle px,
                              const uint64_t r[4], const uint64_t s[4], uint64_t out_proof[32], int inf[3]);

/* Sparse R1CS + witness -> proof in ONE call (CombinePolynomials r1csqap.go:191-210 on the resident sparse system, then
 * GenerateProofs groth16.go:225-278): px is computed on the stream H(x) waits on, while the main stream already accumulates the
 * four sums over w, which do not need px.  *px_inout as in gs_r1cs_px (0 = create; px stays available to the caller).
 * Same proof as gs_r1cs_px followed by gs_groth16_prove_resident. */
int gs_groth16_prove_r1cs(gs_handle pk, gs_handle r1cs, gs_handle w, gs_handle* px_inout, const uint64_t r[4], const uint64_t s[4],
                          uint64_t out_proof[32], int inf[3]);

/* Witness -> proof WITHOUT px: H(x) = (A(x) B(x) - C(x)) / Z(x) is computed straight from the constraint values A w, B w, C w
 * (their values at the nodes n+1..2n by one batched convolution, ONE interpolation instead of three, a Taylor shift; no size-2n
 * product, no division) -- the fast form of CombinePolynomials + DivisorPolynomial (r1csqap.go:191-216) for a witness that
 * satisfies the R1CS, which is the only case in which a proof means anything.  If a constraint is violated the call falls back to
 * the exact route and returns what gs_groth16_prove_r1cs returns.  Same proof as gs_r1cs_px + gs_groth16_prove_resident. */
int gs_groth16_prove_witness(gs_handle pk, gs_handle r1cs, gs_handle w, const uint64_t r[4], const uint64_t s[4],
                             uint64_t out_proof[32], int inf[3]);
/* Keys with an EVALUATION-BASIS copy of PowersTauDelta take a shorter way still.  The prover only needs the group element
 * sum_i h_i PowersTauDelta[i] = H(tau) Z(tau)/delta G (groth16.go:139-149, 269-271); with E[j-1] = l_j(tau) Z(tau)/delta G, l_j the
 * Lagrange basis over the nodes n+1..2n (n = #constraints), the same element is sum_j H(n+j) E[j-1] -- an MSM over the VALUES of H,
 * which fall out of the one batched convolution: no interpolation, no Taylor shift, no host wait (the violated-constraint count
 * is read when the proof is collected; a non-zero count repeats the proof on the exact route).  gs_groth16_setup builds E while it
 * knows tau (64 B per constraint + its window table; gs_handle_bytes reports both); gs_groth16_pk_set_eval attaches one to a key
 * loaded from a file (`bases`: n G1 points from gs_g1_upload; nobody can check them against tau -- a wrong array yields proofs that
 * do not verify, as a wrong PowersTauDelta does); gs_groth16_pk_export which = 7 reads it back.  Bit-identical proofs either way.
 * gs_groth16_prove_witness_begin is the pipelined form (collect with gs_groth16_prove_end; same three slots per device). */
int gs_groth16_pk_set_eval(gs_handle pk, gs_handle bases);
int gs_pk_eval_count(gs_handle pk, size_t* count);     /* Groth16 or Pinocchio key: evaluation-basis points it holds (0 = none) */
int gs_groth16_prove_witness_begin(gs_handle pk, gs_handle r1cs, gs_handle w, const uint64_t r[4], const uint64_t s[4], uint64_t* ticket);

/* Pipelined proving (inputs resident): gs_groth16_prove_begin enqueues the whole device side of one proof and returns a
 * ticket without waiting; gs_groth16_prove_end waits for THAT proof only, then runs the host tail and writes the proof
 * (same layout as gs_groth16_prove).  At most three tickets (proofs or MSMs) may be outstanding per logical device (a fourth begin returns
 * GS_ERR_BUSY); they own disjoint workspaces, so the plan and
 * bucket accumulations of proof k+1 run while the reduction tails, result download and host tail of proof k are still in
 * progress.  Other entry points stay usable meanwhile (see "Conventions"): the key and the vectors a ticket reads may even be
 * gs_free'd -- they are released when the ticket has been collected. */
int gs_groth16_prove_begin(gs_handle pk, gs_handle w, gs_handle px, const uint64_t r[4], const uint64_t s[4], uint64_t* ticket);
int gs_groth16_prove_end(uint64_t ticket, uint64_t out_proof[32], int inf[3]);
/* HOST-BUFFER TICKETS: the reference's own call shape -- groth16.GenerateProofs(circuit, pk, w, px) gets a NEW w (and px) in host
 * memory with every call (groth16/groth16.go:225; cli/main.go:480-501 computes them per proof) -- at the pipelined rate.  The
 * arrays are staged into device buffers that the ticket's SLOT owns (grow-only: once the three slots have been used a stream of
 * proofs performs no hipMalloc and no hipFree, see gs_alloc_counters) on a copy stream of their own: the PCIe transfer of proof
 * k + 3 runs beside the accumulations of proofs k + 1 and k + 2, and the call enqueues the proof once the copy has landed (measured:
 * cheaper than a cross-stream event; the device is busy with the tickets before this one meanwhile).  The caller's arrays have been
 * consumed when the call returns.  Collect with gs_groth16_prove_end; same three slots per device.
 *   gs_groth16_prove_host_begin          w and px from the host (32 + 64 MiB at 2^20 constraints)
 *   gs_groth16_prove_witness_host_begin  w only, against a resident sparse R1CS (gs_r1cs_upload): the witness routes above
 *   gs_groth16_prove_witness_host        the blocking form of the latter (the upload is part of the call, as in gs_groth16_prove) */
int gs_groth16_prove_host_begin(gs_handle pk, const uint64_t* w, size_t nw, const uint64_t* px, size_t npx, const uint64_t r[4], const uint64_t s[4],
                                uint64_t* ticket);
int gs_groth16_prove_witness_host_begin(gs_handle pk, gs_handle r1cs, const uint64_t* w, size_t nw, const uint64_t r[4], const uint64_t s[4],
                                        uint64_t* ticket);
int gs_groth16_prove_witness_host(gs_handle pk, gs_handle r1cs, const uint64_t* w, size_t nw, const uint64_t r[4], const uint64_t s[4],
                                  uint64_t out_proof[32], int inf[3]);
/* Abandon a ticket of any kind (proof or MSM) without collecting its result: waits for its device work, frees its slot and the
 * references it holds.  An error path that cannot call the matching _end must call this, or the slot stays occupied. */
int gs_ticket_cancel(uint64_t ticket);

/* One proof over several GPUs: every rank holds the key (or just its slice of it, below) and the resident w / px, takes shard `shard_index` of
 * `shard_count` of the term ranges (contiguous, first ranges one longer when they do not divide), computes H(x) locally
 * (the polynomial stage is replicated) and returns its five raw MSM sums as affine points:
 * out_sums = At (8 words) | G1.BACGamma (8) | G2.BACGamma (16) | BACDelta (8) | PowersTauDelta.h (8); inf[5] in that order.
 * The ranks exchange these 416-byte records (ncclAllGather as bytes), add them with gs_g1_sum_affine / gs_g2_sum_affine
 * and call gs_groth16_finish, which applies the O(1) tail of groth16.go:253-275. */
int gs_groth16_prove_partials(gs_handle pk, gs_handle w, gs_handle px, size_t shard_index, size_t shard_count,
                              uint64_t out_sums[48], int inf[5]);
/* Key slices: a rank that only ever runs gs_groth16_prove_partials for shard `shard_index` of `shard_count` needs only that
 * shard's entries of the proving key ("each GPU holds 1/8 of every pk array").  gs_groth16_pk_create_shard takes
 * base handles that hold exactly the slices -- At / G1.BACGamma / G2.BACGamma / BACDelta entries of the contiguous split of
 * [0, nvars), PowersTauDelta entries of the split of [0, nptd_total) (first ranges one longer when they do not divide; the
 * same split gs_groth16_prove_partials uses) -- plus the single elements and Z, which every rank keeps.
 * gs_groth16_pk_shard cuts such a slice out of a resident full key (then gs_free the full key).  A slice key is accepted
 * by gs_groth16_prove_partials (with its own shard), gs_groth16_finish and gs_groth16_pk_export; every other entry point
 * refuses it.  Window tables are built for the slice only: 1/shard_count of the key memory per GPU. */
int gs_groth16_pk_create_shard(gs_handle g1_at, gs_handle g1_bacgamma, gs_handle g2_bacgamma, gs_handle bacdelta, gs_handle ptd,
                               const uint64_t g1_alpha[12], const uint64_t g1_beta[12], const uint64_t g1_delta[12],
                               const uint64_t g2_beta[24], const uint64_t g2_delta[24], const uint64_t* z, size_t nz,
                               size_t nvars, size_t npublic, size_t nptd_total, size_t shard_index, size_t shard_count,
                               gs_handle* out);
int gs_groth16_pk_shard(gs_handle full_pk, size_t shard_index, size_t shard_count, gs_handle* out);
/* The same slice, created on logical device `target_device` (the full key may live on any device). */
int gs_groth16_pk_shard_to(gs_handle full_pk, size_t shard_index, size_t shard_count, int target_device, gs_handle* out);
int gs_groth16_finish(gs_handle pk, const uint64_t sums[48], const int inf_in[5], const uint64_t r[4], const uint64_t s[4],
                      uint64_t out_proof[32], int inf[3]);

/* groth16.GenerateTrustedSetup (groth16.go:94-222) for a SPARSE R1CS (A, B, C in CSR over n constraints x m
 * variables, as gs_r1cs_to_px), with the five toxic scalars injected (what Utils.FqR.Rand() returned at :99-119):
 * toxic = T | Kalpha | Kbeta | Kgamma | Kdelta (5 x 4 words).  Builds the proving key directly on the device --
 * Lagrange basis at tau over the nodes 1..n, transposed sparse mat-vecs for the per-variable evaluations, fixed-base
 * batch multiplications -- and returns it resident in *pk_out (same object as gs_groth16_pk_create).  If vk_out is not
 * NULL it receives the verification key as affine Jacobian triples: G1.Alpha (12 words) | G2.Beta (24) | G2.Gamma (24)
 * | G2.Delta (24) | IC[0..npublic] (12 each).  Shape contract: m in {n+1, n+2}. */
int gs_groth16_setup(size_t n, size_t m, size_t npublic,
                     const uint32_t* a_rowptr, const uint32_t* a_col, const uint64_t* a_val,
                     const uint32_t* b_rowptr, const uint32_t* b_col, const uint64_t* b_val,
                     const uint32_t* c_rowptr, const uint32_t* c_col, const uint64_t* c_val,
                     const uint64_t toxic[20], gs_handle* pk_out, uint64_t* vk_out);
/* snark.GenerateTrustedSetup (snark.go:98-251) for a sparse R1CS; toxic = T | Ka | Kb | Kc | Kbeta | Kgamma | RhoA | RhoB
 * (8 x 4 words; RhoC = RhoA RhoB, :149).  *pk_out: resident Pinocchio key (as gs_pinocchio_pk_create builds).  vk_out
 * (may be NULL), affine Jacobian triples: Vka (24 words, G2) | Vkb (12) | Vkc (24) | G1Kbg (12) | G2Kbg (24) | G2Kg (24)
 * | Vkz (24) | IC[0..npublic] (12 each). */
int gs_pinocchio_setup(size_t n, size_t m, size_t npublic,
                       const uint32_t* a_rowptr, const uint32_t* a_col, const uint64_t* a_val,
                       const uint32_t* b_rowptr, const uint32_t* b_col, const uint64_t* b_val,
                       const uint32_t* c_rowptr, const uint32_t* c_col, const uint64_t* c_val,
                       const uint64_t toxic[32], gs_handle* pk_out, uint64_t* vk_out);
/* Read one array of a resident Pinocchio key back (which = 0 A, 1 Ap, 2 B (G2, 24 words per point), 3 Bp, 4 C, 5 Cp,
 * 6 Kp, 7 G1T; 8 = pk.Z, count coefficients of 4 x u64; 9 = the evaluation-basis copy of G1T, n points or none).  Note A and Ap hold infinity for i <= NPublic (what the
 * prover sums, snark.go:265). */
int gs_pinocchio_pk_export(gs_handle pk, int which, uint64_t* jacobian, size_t count);
/* Read one array of a resident Groth16 key back as affine Jacobian triples: which = 0 G1.At, 1 G1.BACGamma,
 * 2 G2.BACGamma (24 words per point), 3 BACDelta, 4 PowersTauDelta; 5 = the single elements (count = 5: G1 Alpha, Beta,
 * Delta as 3 x 12 words, then G2 Beta, Delta as 2 x 24 words); 6 = pk.Z (count coefficients of 4 x u64); 7 = the
 * evaluation-basis copy of PowersTauDelta (n points, or 0 when the key has none).  count must equal the array length.  With 0..6 a resident key can be written out in full (utils.GrothSetupToString). */
int gs_groth16_pk_export(gs_handle pk, int which, uint64_t* jacobian, size_t count);

/* ---- Pinocchio prover (snark.go) ------------------------------------------------------------ */
/* snark.Pk (snark.go:16-26): A, Ap, Bp, C, Cp, Kp: m G1 points; B: m G2 points; G1T: len(Z). */
int gs_pinocchio_pk_create(gs_handle a, gs_handle ap, gs_handle b_g2, gs_handle bp, gs_handle c, gs_handle cp,
                           gs_handle kp, gs_handle g1t, const uint64_t* z, size_t nz,
                           size_t nvars, size_t npublic, gs_handle* out);
/* snark.GenerateProofs (snark.go:254-289).  out_proof = PiA | PiAp | PiB(16) | PiBp | PiC |
 * PiCp | PiH | PiKp = 7*8 + 16 = 72 words; inf[8] in that order. */
int gs_pinocchio_prove(gs_handle pk, const uint64_t* w, size_t nw, const uint64_t* px, size_t npx,
                       uint64_t out_proof[72], int inf[8]);
/* Same with w and px already resident (gs_scalars_upload); what bench.py --workload prove_pinocchio times. */
int gs_pinocchio_prove_resident(gs_handle pk, gs_handle w, gs_handle px, uint64_t out_proof[72], int inf[8]);
/* snark.GenerateProofs from the witness alone (the Pinocchio twin of gs_groth16_prove_witness): H(x) straight from the constraint
 * values of the resident R1CS, no px; a witness that violates a constraint takes the exact px route (same result as the reference). */
int gs_pinocchio_prove_witness(gs_handle pk, gs_handle r1cs, gs_handle w, uint64_t out_proof[72], int inf[8]);
/* ... with the evaluation-basis copy of G1T (E[j-1] = l_j(tau) G over the nodes n+1..2n, so that sum_j H(n+j) E[j-1] = H(tau) G =
 * sum_i h_i G1T[i], snark.go:239-247, 284-286) when the key has one: gs_pinocchio_setup builds it, gs_pinocchio_pk_set_eval attaches
 * one, gs_pinocchio_pk_export which = 9 reads it back.  gs_pinocchio_prove_witness_begin: pipelined, collect with gs_pinocchio_prove_end. */
int gs_pinocchio_pk_set_eval(gs_handle pk, gs_handle bases);
int gs_pinocchio_prove_witness_begin(gs_handle pk, gs_handle r1cs, gs_handle w, uint64_t* ticket);
/* Pipelined Pinocchio proving: same tickets as gs_groth16_prove_begin / _end (the three in-flight slots are shared between
 * Groth16 proofs, Pinocchio proofs and MSMs). */
int gs_pinocchio_prove_begin(gs_handle pk, gs_handle w, gs_handle px, uint64_t* ticket);
int gs_pinocchio_prove_end(uint64_t ticket, uint64_t out_proof[72], int inf[8]);
/* host-buffer tickets of snark.GenerateProofs (snark.go:254): as the Groth16 ones above; collect with gs_pinocchio_prove_end */
int gs_pinocchio_prove_host_begin(gs_handle pk, const uint64_t* w, size_t nw, const uint64_t* px, size_t npx, uint64_t* ticket);
int gs_pinocchio_prove_witness_host_begin(gs_handle pk, gs_handle r1cs, const uint64_t* w, size_t nw, uint64_t* ticket);
int gs_pinocchio_prove_witness_host(gs_handle pk, gs_handle r1cs, const uint64_t* w, size_t nw, uint64_t out_proof[72], int inf[8]);

/* ---- several GPUs (BASELINE configs[3] "MSM sharded across 8 GPUs" and configs[4] "one proof per GPU") ----------
 * The prover's sums run over independent terms (groth16.go:243-250,269-271): each device sums one contiguous shard of the
 * term ranges, and ONE record per device is exchanged -- the five partial sums of a proof (416 bytes: 48 words of affine
 * points | inf[5], shard index as u32) or one partial point (72 / 136 bytes).  RCCL cannot add curve points, so the exchange is
 * ncclAllGather of the records as ncclUint8 followed by N - 1 additions on the host core.
 *
 * Communicator.  gs_comm_init_local: every RCCL rank lives in this process, one per distinct physical device of gs_init's
 * list (ncclCommInitAll); the *_multi entry points then pass their records through ncclAllGather.  gs_comm_init_rank: one
 * process per GPU -- rank 0 obtains an id with gs_comm_unique_id, the host language distributes the 128 bytes, every process
 * calls gs_comm_init_rank(id, nranks, rank) (its current logical device joins); the *_sharded entry points gather over it.
 * gs_comm_allgather exposes the byte gather itself (host buffers; local mode: `send` holds one block per local rank and
 * `recv` receives rank 0's copy of all blocks).  *collectives counts the ncclAllGather calls completed so far. */
int gs_comm_unique_id(uint8_t out_id[128]);
int gs_comm_init_rank(const uint8_t id[128], int nranks, int rank);
int gs_comm_init_local(void);
void gs_comm_destroy(void);
int gs_comm_info(int* nranks, int* rank, int* is_local, uint64_t* collectives);
int gs_comm_allgather(const void* send, size_t bytes_per_rank, void* recv /* nranks x bytes_per_rank */);

/* One MSM over `ndev` logical devices of this process: bases[d] / scalars[d] hold shard d of the term range (the same
 * contiguous split of both, e.g. made with gs_g1_clone / gs_scalars_clone), resident on one logical device each.  The
 * devices run concurrently (one host thread each); *used_rccl (may be NULL) reports whether the partial points travelled
 * through ncclAllGather (a local communicator exists and the logical devices spread evenly over its ranks) or were simply
 * read from this process's memory.  Same result as one gs_msm_g1 over the whole range. */
int gs_msm_g1_multi(const gs_handle* bases, const gs_handle* scalars, int ndev, uint64_t out_affine[8], int* is_inf, int* used_rccl);
int gs_msm_g2_multi(const gs_handle* bases, const gs_handle* scalars, int ndev, uint64_t out_affine[16], int* is_inf, int* used_rccl);
/* One Groth16 proof over `ndev` logical devices: pk[d] = the full key or slice d of ndev (gs_groth16_pk_shard_to), w[d] / px[d]
 * = replicas of the witness and of P(x) on the same logical device (gs_scalars_clone).  Device d runs
 * gs_groth16_prove_partials for shard d; the records are exchanged as above and the O(1) tail runs once.  Same proof as
 * gs_groth16_prove_resident with the full key. */
int gs_groth16_prove_multi(const gs_handle* pk, const gs_handle* w, const gs_handle* px, int ndev, const uint64_t r[4], const uint64_t s[4],
                           uint64_t out_proof[32], int inf[3], int* used_rccl);
/* One process per GPU: this rank's shard (rank / nranks of the communicator made by gs_comm_init_rank), gathered inside the
 * library; every rank returns the complete result.  pk is the full key or this rank's slice; bases / scalars hold this
 * rank's shard of the term range. */
int gs_groth16_prove_sharded(gs_handle pk, gs_handle w, gs_handle px, const uint64_t r[4], const uint64_t s[4], uint64_t out_proof[32], int inf[3]);
/* Strong scaling WITHOUT a replicated polynomial stage ("run on GPU 0 and broadcast hx shards"), for keys with an
 * evaluation-basis array (gs_groth16_pk_set_eval; slices carry their share of it).  Per proof ONE rank -- the owner; over a stream of
 * proofs the ranks take turns -- runs gs_groth16_witness_values: resident sparse R1CS + witness -> the n values H(n+1), ..., H(2n)
 * as a resident scalar vector (*hv_inout: 0 = create; *violated = number of roots of Z at which the witness breaks a constraint: the
 * values are then meaningless and the proof must take gs_r1cs_px + the px entry points).  The owner scatters the vector -- slice k
 * of the contiguous split to rank k: gs_scalars_clone between the logical devices of one process, gs_scalars_scatter (ncclSend /
 * ncclRecv, one group) between processes -- and every rank sums ONLY its term ranges: gs_groth16_prove_partials_values (four sums
 * over its slice of w, the fifth over its slice of the values; record layout of gs_groth16_prove_partials), wrapped with the
 * exchange and the tail by gs_groth16_prove_multi_values / gs_groth16_prove_sharded_values.  Same proof as every other route. */
int gs_groth16_witness_values(gs_handle pk, gs_handle r1cs, gs_handle w, gs_handle* hv_inout, uint32_t* violated);
int gs_groth16_prove_partials_values(gs_handle pk, gs_handle w, gs_handle hv_slice, size_t shard_index, size_t shard_count,
                                     uint64_t out_sums[48], int inf[5]);
/* ... pipelined: a rank streams its shards of consecutive proofs through the three slots (begin enqueues, end collects the five sums). */
int gs_groth16_partials_values_begin(gs_handle pk, gs_handle w, gs_handle hv_slice, size_t shard_index, size_t shard_count, uint64_t* ticket);
int gs_groth16_partials_end(uint64_t ticket, uint64_t out_sums[48], int inf[5]);
int gs_groth16_prove_multi_values(const gs_handle* pk, const gs_handle* w, const gs_handle* hv_slices, int ndev, const uint64_t r[4], const uint64_t s[4],
                                  uint64_t out_proof[32], int inf[3], int* used_rccl);
int gs_groth16_prove_sharded_values(gs_handle pk, gs_handle w, gs_handle hv_slice, const uint64_t r[4], const uint64_t s[4], uint64_t out_proof[32], int inf[3]);
int gs_scalars_scatter(gs_handle full_on_root, size_t total, int root, gs_handle* slice_inout);
int gs_msm_g1_sharded(gs_handle bases, gs_handle scalars, uint64_t out_affine[8], int* is_inf);
int gs_msm_g2_sharded(gs_handle bases, gs_handle scalars, uint64_t out_affine[16], int* is_inf);
/* A batch of independent proofs (configs[4]; no collective): proof i reads w[i] / px[i], runs on the logical device those
 * handles live on with the key pk_of_device[that device] (0 for unused devices), through the pipelined prover (three in
 * flight per device, devices concurrently).  r, s: nproofs x 4 words; out_proofs: nproofs x 32 words; inf: nproofs x 3. */
int gs_groth16_prove_batch(const gs_handle* pk_of_device, int ndev, const gs_handle* w, const gs_handle* px, size_t nproofs,
                           const uint64_t* r, const uint64_t* s, uint64_t* out_proofs, int* inf);

/* ---- snark.GenerateProofs over several GPUs (snark.go:254-289; the scheme above applied to Pinocchio) ----
 * A Pinocchio proof IS its eight MSM sums -- PiA | PiAp | PiB (G2) | PiBp | PiC | PiCp | PiH | PiKp, snark.go:265-286, there is no
 * tail -- so rank k of N sums its term ranges and the eight partial points of the ranks add up to the proof (616-byte records).
 *   gs_pinocchio_pk_shard / _shard_to      a slice of a resident full key: entries [k/N, (k+1)/N) of the seven per-variable arrays,
 *                                          of G1T and of the evaluation-basis array; Z travels with every slice
 *   gs_pinocchio_prove_partials            the eight sums of shard k from resident w and px (full key or slice k), proof layout
 *   gs_pinocchio_witness_values            the proof owner's polynomial stage, as gs_groth16_witness_values (the same H)
 *   gs_pinocchio_prove_partials_values     the eight sums with PiH over this rank's slice of H's values (no polynomial work here)
 *   gs_pinocchio_combine                   n records (n x 72 words, n x 8 flags) -> the proof (host additions)
 *   gs_pinocchio_prove_multi[_values]      one process, ndev logical devices: partials on every device concurrently, exchange
 *                                          (RCCL when a local communicator spans the devices), addition
 *   gs_pinocchio_prove_sharded[_values]    one process per GPU: this rank's partials, ncclAllGather of the records, addition;
 *                                          every rank returns the complete proof
 *   gs_pinocchio_prove_batch               a batch of independent proofs, three in flight per device, no collective
 * Same proof as gs_pinocchio_prove_resident with the full key, whatever N. */
int gs_pinocchio_pk_shard(gs_handle full_pk, size_t shard_index, size_t shard_count, gs_handle* out);
int gs_pinocchio_pk_shard_to(gs_handle full_pk, size_t shard_index, size_t shard_count, int target_device, gs_handle* out);
int gs_pinocchio_prove_partials(gs_handle pk, gs_handle w, gs_handle px, size_t shard_index, size_t shard_count, uint64_t out_sums[72], int inf[8]);
int gs_pinocchio_witness_values(gs_handle pk, gs_handle r1cs, gs_handle w, gs_handle* hv_inout, uint32_t* violated);
int gs_pinocchio_prove_partials_values(gs_handle pk, gs_handle w, gs_handle hv_slice, size_t shard_index, size_t shard_count,
                                       uint64_t out_sums[72], int inf[8]);
int gs_pinocchio_combine(const uint64_t* sums /* n x 72 */, const int* inf_in /* n x 8 */, size_t n, uint64_t out_proof[72], int inf[8]);
int gs_pinocchio_prove_multi(const gs_handle* pk, const gs_handle* w, const gs_handle* px, int ndev, uint64_t out_proof[72], int inf[8], int* used_rccl);
int gs_pinocchio_prove_multi_values(const gs_handle* pk, const gs_handle* w, const gs_handle* hv_slices, int ndev, uint64_t out_proof[72], int inf[8],
                                    int* used_rccl);
int gs_pinocchio_prove_sharded(gs_handle pk, gs_handle w, gs_handle px, uint64_t out_proof[72], int inf[8]);
int gs_pinocchio_prove_sharded_values(gs_handle pk, gs_handle w, gs_handle hv_slice, uint64_t out_proof[72], int inf[8]);
int gs_pinocchio_prove_batch(const gs_handle* pk_of_device, int ndev, const gs_handle* w, const gs_handle* px, size_t nproofs,
                             uint64_t* out_proofs /* nproofs x 72 */, int* inf /* nproofs x 8 */);

/* ---- timing of the last prove / msm call (device time, HIP events on the library stream) --- */
typedef struct {
  float total_ms;        /* all device work of the call */
  float plan_ms;         /* scalar digit extraction + bucket sort */
  float accumulate_ms;   /* bucket accumulation kernels (dominant) */
  float reduce_ms;       /* bucket reduction + window combination + normalisation */
  float poly_ms;         /* H(x) = P(x)/Z(x) stage */
  float h2d_ms;          /* host-to-device copies inside the call */
  /* the dominant kernel alone (k_bucket_accumulate), summed over its launches in the call: */
  float acc_g1_ms;       /* G1 bucket-accumulation launches */
  float acc_g2_ms;       /* G2 bucket-accumulation launches */
  uint32_t acc_g1_launches, acc_g2_launches;
  uint64_t acc_g1_terms; /* (terms x base arrays) those G1 launches consumed */
  uint64_t acc_g2_terms;
  uint64_t acc_g1_adds;  /* mixed point additions those G1 launches performed = NON-ZERO digits of their plans x base arrays (counted by the plan) */
  uint64_t acc_g2_adds;
  uint32_t window_bits;  /* Pippenger window width c of the last plan; every term costs floor(254 / c) + 1 additions */
  uint32_t fallbacks;    /* witness-route calls that had to repeat on the exact px route because the witness violates a constraint */
  /* What the plans of the call found in their scalars (summed over the MSM groups, per base array): a term has one digit per window;
   * a ZERO digit costs nothing, every other one is one bucket addition.  Uniform scalars: plan_entries ~ plan_digits; a witness full
   * of 0 / 1 / small values: a fraction of it. */
  uint64_t plan_digits;   /* terms x windows x base arrays */
  uint64_t plan_entries;  /* non-zero digits x base arrays = mixed additions the accumulation kernels really performed */
  uint32_t heavy_buckets; /* buckets cut into more than 64 chunks (0/1-heavy witnesses), combined by a block-wide tree; summed per MSM GROUP
                           * (a proof's G2 and G1 groups over w share one plan: its heavy buckets count once for each) */
  uint32_t reserved;
} gs_timing;
int gs_last_timing(gs_timing* out);                         /* the calling thread's current logical device */
int gs_device_timing(int logical_device, gs_timing* out);

/* ---- device memory: accounting and eviction ---------------------------------------------------------------------------
 * The window tables (rows 2^(c j) P_i of every base array, built on a handle's first proof / MSM) are 15x the key data at
 * c = 17: 5.6 GiB per 2^20-constraint Groth16 key.  A host that keeps several keys resident can see what each one costs and
 * drop the tables of the idle ones; they are rebuilt on the handle's next use (~140 ms per 2^20 Groth16 key). */
typedef struct {
  uint64_t device_total_bytes, device_free_bytes;   /* hipMemGetInfo of the current logical device's GPU */
  uint64_t library_bytes;     /* every device byte this library holds, all logical devices of the process */
  uint64_t object_bytes;      /* the current logical device's handles: base arrays, keys (with their divisor caches), scalars, R1CS */
  uint64_t table_bytes;       /* window tables of those handles */
  uint64_t workspace_bytes;   /* bucket sets, chunk partials, result staging, fixed-base tables (plan / polynomial caches: in library_bytes) */
  uint64_t objects;           /* live handles on the current logical device */
  uint64_t evictions;         /* window tables this logical device dropped because an allocation found no memory (was `reserved`: same size) */
} gs_memory;
int gs_memory_query(gs_memory* out);
int gs_handle_bytes(gs_handle h, uint64_t* object_bytes, uint64_t* table_bytes);     /* either pointer may be NULL */
/* Free the window tables of a key or base array (queues behind outstanding tickets); results of later calls are unchanged. */
int gs_release_tables(gs_handle h);
/* WHEN a base array gets its window table (every logical device; results never depend on it):
 *   0 auto (default)  the reference proves once per key load (cli/main.go:330-349), and the tables of a 2^20 key cost ~140 ms and
 *                     5.6 GiB -- fifteen proofs' worth -- before the first proof.  An array without a table is summed TABLE-FREE
 *                     (a bucket set per window, every window adds the base point itself, the window sums recombined by Horner:
 *                     ~1.2x the additions); from its second use on its table is built in INSTALMENTS: every call buys slabs of
 *                     the table in proportion to its own work (~+60-100% of a table-free proof; GS_TABLE_BUDGET_PCT scales it),
 *                     enqueued ahead of its own accumulations, and the first call that finds the table complete switches over.
 *   1 always          the table is built inside the first call that needs it (rounds 1-4)
 *   2 never           table-free only (0.4 GiB per 2^20 key instead of 6; what 2^24 constraints on one GPU use)
 * gs_build_tables builds them NOW (blocking, whatever the policy): a server warming a key it will prove with for hours, and what
 * bench.py's steady-state numbers are measured on.  route: 0 = everything the key can use, 1 = only what the px routes need,
 * 2 = only what the witness routes need (the evaluation-basis array instead of PowersTauDelta / G1T); ignored for base arrays. */
int gs_set_table_policy(int policy);
int gs_build_tables(gs_handle h, int route);
/* Out of device memory is not fatal while window tables exist: an allocation that fails evicts least-recently-used tables of the
 * logical device (never one a ticket or the running call holds), retries once, and only then returns GS_ERR_HIP; an evicted table
 * comes back the way the policy says.  gs_memory.evictions counts them.  gs_set_memory_limit caps the device bytes the library may
 * hold (0 = no cap) -- a development / test hook that makes the condition reachable without filling 288 GB. */
int gs_set_memory_limit(uint64_t bytes);
/* hipMalloc / hipFree calls the library has made so far (process-wide): a stream of pipelined proofs -- resident or host-buffer
 * tickets -- moves neither in steady state.  Either pointer may be NULL. */
int gs_alloc_counters(uint64_t* allocs, uint64_t* frees);
/* sizeof(gs_timing), sizeof(gs_memory) of THIS library: both structs are written through caller pointers, and a binding compiled
 * against another revision of this header can tell before it passes a short buffer (gs_timing grew in round 4). */
int gs_abi_sizes(size_t* timing_bytes, size_t* memory_bytes);
/* Free every cached workspace of the current logical device (bucket sets, plan buffers, NTT twiddles, node trees, factorial
 * tables); rebuilt on demand.  Handles and their tables stay. */
int gs_trim(void);

/* Tunables (0 = automatic): Pippenger window bits. */
int gs_set_window_bits(int c);
/* Witness route of keys that carry an evaluation-basis array (gs_groth16_pk_set_eval): 1 (default) = h-MSM over H's values,
 * 0 = the coefficient route every other key takes.  Identical proofs; for measurements and tests. */
int gs_set_eval_basis(int enabled);

/* ---- verifier (host side) ---------------------------------------------------------------- */
/* The step after the prover.  These four entry points run on the calling host thread: a proof check is
 * O(1) work (a handful of pairings, a few thousand Fq products, latency bound), so there is nothing to offload.  They
 * need no gs_init, never touch the device and are not a fallback for anything above; they do not take the library
 * mutex, so a verifier thread runs beside in-flight proofs.
 *
 * bn128.Pairing(p1, p2) (bn128/bn128.go:179-186): the reduced optimal ate pairing MillerLoop^((q^12 - 1)/r) as the
 * reference's [2][3][2]*big.Int, 12 x 4 words in that nesting order, standard form.  Bit-identical to the reference's
 * value (same tower Fq2[u]/(u^2+1), Fq6 = Fq2[v]/(v^3-(9+u)), Fq12 = Fq6[w]/(w^2-v); lines differ only by a factor
 * the final exponentiation removes).  Either point at infinity gives 1.  GS_ERR_ARG if a point is off its curve. */
int gs_pairing(const uint64_t g1[12], const uint64_t g2[24], uint64_t out_fq12[48]);
/* *ok = [ prod_i e(g1_i, g2_i) == 1 ]: one multi-Miller loop (shared squarings, one batched inversion per step) and
 * ONE final exponentiation for all k pairs.  Points off the curve, or G2 points outside the order-r subgroup
 * (psi(Q) != [6x^2]Q; the twist has a large cofactor), give *ok = 0 (the reference would compute a meaningless Fq12
 * value and compare it).  gs_groth16_verify / gs_pinocchio_verify apply the subgroup check to the prover's G2 element
 * (PiB); the G2 elements of the verification key are key material, validated by whoever installs the key. */
int gs_pairing_check(const uint64_t* g1 /* k x 12 */, const uint64_t* g2 /* k x 24 */, size_t k, int* ok);
/* Input strictness of the two verifiers (process-wide).  0 (default) answers exactly like the reference's big.Int code: a
 * coordinate X and X + q, a public signal x and x + r name the same element, and fewer public signals than the vk has IC
 * points verify the statement with the missing inputs taken as 0 (groth16.go:283-286 loops over publicSignals only).  1 = strict:
 * coordinates must be < q and public signals < r (otherwise *ok = 0: no proof / input aliasing), and npublic must equal
 * nic - 1 (otherwise GS_ERR_SHAPE).  Deployments that accept proofs from untrusted parties should switch it on. */
int gs_verify_set_strict(int on);

/* groth16.VerifyProof(vk, proof, publicSignals, debug) (groth16/groth16.go:281-305):
 *   icPubl = IC[0] + sum_i publicSignals[i] * IC[i+1];   e(PiA, PiB) == e(Alpha, Beta) e(icPubl, Gamma) e(PiC, Delta)
 * as one 4-pair product check.  *ok = 1 accept / 0 reject.  GS_ERR_SHAPE when nic < npublic + 1 (the reference panics
 * on vk.IC[i+1] out of range). */
int gs_groth16_verify(const uint64_t vk_g1_alpha[12], const uint64_t vk_g2_beta[24], const uint64_t vk_g2_gamma[24],
                      const uint64_t vk_g2_delta[24], const uint64_t* vk_ic /* nic x 12 */, size_t nic,
                      const uint64_t* public_signals /* npublic x 4 */, size_t npublic,
                      const uint64_t pi_a[12], const uint64_t pi_b[24], const uint64_t pi_c[12], int* ok);
/* snark.VerifyProof (snark.go:292-368): the five Pinocchio checks in the reference's order, each a product check;
 * proof = PiA, PiAp (12 words each), PiB (24), PiBp, PiC, PiCp, PiH, PiKp (12 each) = 108 words (snark.go:59-69).
 * *failed_check (optional) = 0, or 1..5 for the first equation that does not hold (the reference's debug prints). */
int gs_pinocchio_verify(const uint64_t vka[24], const uint64_t vkb[12], const uint64_t vkc[24], const uint64_t g1kbg[12],
                        const uint64_t g2kbg[24], const uint64_t g2kg[24], const uint64_t vkz[24],
                        const uint64_t* vk_ic /* nic x 12 */, size_t nic, const uint64_t* public_signals, size_t npublic,
                        const uint64_t* proof /* 108 words */, int* ok, int* failed_check);

#if defined(__GNUC__)
#pragma GCC visibility pop
#endif
#ifdef __cplusplus
}
#endif
#endif /* GOSNARK_HIP_H */
