package gosnarkhip

/*
#include "gosnark_hip.h"
*/
import "C"

import (
	"errors"
	"math/big"
	"runtime"
	"unsafe"
)

// The finer seams SURVEY 8b lists "with stable signatures": polynomial add / sub / eval, Z, R1CSToQAP, CombinePolynomials,
// single scalar multiplications and additions, the raw pairing; plus the tickets and collectives that round 2 left unbound.
// The drop-in packages r1csqaphip and bn128hip put the reference's method signatures on top of these.

func polyAddSub(a, b []*big.Int, order *big.Int, sub bool) ([]*big.Int, error) {
	n := len(a)
	if len(b) > n {
		n = len(b)
	}
	if n == 0 {
		return nil, nil
	}
	ab, err := Scalars(a, order)
	if err != nil {
		return nil, err
	}
	bb, err := Scalars(b, order)
	if err != nil {
		return nil, err
	}
	out := make([]uint64, 4*n)
	err = call(func() C.int {
		if sub {
			return C.gs_poly_sub(ptr(ab), C.size_t(len(a)), ptr(bb), C.size_t(len(b)), ptr(out))
		}
		return C.gs_poly_add(ptr(ab), C.size_t(len(a)), ptr(bb), C.size_t(len(b)), ptr(out))
	})
	runtime.KeepAlive(ab)
	runtime.KeepAlive(bb)
	if err != nil {
		return nil, err
	}
	return unpackScalars(out), nil
}

// PolyAdd is PolynomialField.Add (r1csqap/r1csqap.go:94-103); PolySub is Sub (:106-115): coefficient-wise, the shorter operand
// zero-extended, results in [0, r).
func PolyAdd(a, b []*big.Int, order *big.Int) ([]*big.Int, error) { return polyAddSub(a, b, order, false) }
func PolySub(a, b []*big.Int, order *big.Int) ([]*big.Int, error) { return polyAddSub(a, b, order, true) }

// PolyEval is PolynomialField.Eval (r1csqap.go:118-126): sum_i v_i x^i.
func PolyEval(v []*big.Int, x, order *big.Int) (*big.Int, error) {
	vb, err := Scalars(v, order)
	if err != nil {
		return nil, err
	}
	xb, err := Scalars([]*big.Int{x}, order)
	if err != nil {
		return nil, err
	}
	var out [4]uint64
	err = call(func() C.int { return C.gs_poly_eval(ptr(vb), C.size_t(len(v)), ptr(xb), (*C.uint64_t)(unsafe.Pointer(&out[0]))) })
	runtime.KeepAlive(vb)
	runtime.KeepAlive(xb)
	if err != nil {
		return nil, err
	}
	return word(out[:]), nil
}

// ZPoly is Z(x) = prod_{i=1}^{deg} (x - i): deg + 1 coefficients (r1csqap.go:177-186, groth16.go:122-131, snark.go:221-231).
func ZPoly(deg int) ([]*big.Int, error) {
	if deg < 0 {
		return nil, errors.New("gosnark-hip: negative degree")
	}
	out := make([]uint64, 4*(deg+1))
	if err := call(func() C.int { return C.gs_zpoly(C.size_t(deg), ptr(out)) }); err != nil {
		return nil, err
	}
	return unpackScalars(out), nil
}

// R1CSToPx is the scalable form of R1CSToQAP + CombinePolynomials (r1csqap.go:161-210) on a sparse system held in host memory:
// ax, bx, cx (n coefficients each: the interpolants of A w, B w, C w over the nodes 1..n) and px = ax * bx - cx (2n - 1).
func R1CSToPx(a, b, c CSR, nvars int, w []*big.Int, order *big.Int) (ax, bx, cx, px []*big.Int, err error) {
	n := len(a.RowPtr) - 1
	if n < 1 || len(b.RowPtr) != n+1 || len(c.RowPtr) != n+1 {
		return nil, nil, nil, nil, errors.New("gosnark-hip: A, B, C must have the same number of constraints")
	}
	if len(w) != nvars {
		return nil, nil, nil, nil, errors.New("gosnark-hip: len(w) != number of variables")
	}
	wb, err := Scalars(w, order)
	if err != nil {
		return
	}
	oa, ob, oc, op := make([]uint64, 4*n), make([]uint64, 4*n), make([]uint64, 4*n), make([]uint64, 4*(2*n-1))
	err = call(func() C.int {
		return C.gs_r1cs_to_px(C.size_t(n), C.size_t(nvars), ptr32(a.RowPtr), ptr32(a.Col), ptr(a.Val), ptr32(b.RowPtr), ptr32(b.Col), ptr(b.Val),
			ptr32(c.RowPtr), ptr32(c.Col), ptr(c.Val), ptr(wb), ptr(oa), ptr(ob), ptr(oc), ptr(op))
	})
	runtime.KeepAlive(a)
	runtime.KeepAlive(b)
	runtime.KeepAlive(c)
	runtime.KeepAlive(wb)
	if err != nil {
		return
	}
	return unpackScalars(oa), unpackScalars(ob), unpackScalars(oc), unpackScalars(op), nil
}

// G1MulScalar is bn128.G1.MulScalar (bn128/g1.go:140-155) as a one-term MSM: the affine representative [x, y, 1] of e * p.
func G1MulScalar(device int, p [3]*big.Int, e, order *big.Int) ([3]*big.Int, error) {
	h, err := UploadG1(device, [][3]*big.Int{p})
	if err != nil {
		return [3]*big.Int{}, err
	}
	defer Free(h)
	s, err := Scalars([]*big.Int{e}, order)
	if err != nil {
		return [3]*big.Int{}, err
	}
	return MSMG1(h, s, 0)
}

// G1Add is bn128.G1.Add (g1.go:32-89) as the two-term MSM 1 * p1 + 1 * p2 -- complete: P + P and P + (-P) are handled, where the
// reference's formula returns Z = 0 for P + P (SURVEY fact 9).
func G1Add(device int, p1, p2 [3]*big.Int) ([3]*big.Int, error) {
	h, err := UploadG1(device, [][3]*big.Int{p1, p2})
	if err != nil {
		return [3]*big.Int{}, err
	}
	defer Free(h)
	return MSMG1(h, []uint64{1, 0, 0, 0, 1, 0, 0, 0}, 0)
}

// G2MulScalar / G2Add: bn128.G2.MulScalar (g2.go:142-181) / Add (:32-89).
func G2MulScalar(device int, p [3][2]*big.Int, e, order *big.Int) ([3][2]*big.Int, error) {
	h, err := UploadG2(device, [][3][2]*big.Int{p})
	if err != nil {
		return [3][2]*big.Int{}, err
	}
	defer Free(h)
	s, err := Scalars([]*big.Int{e}, order)
	if err != nil {
		return [3][2]*big.Int{}, err
	}
	return MSMG2(h, s, 0)
}
func G2Add(device int, p1, p2 [3][2]*big.Int) ([3][2]*big.Int, error) {
	h, err := UploadG2(device, [][3][2]*big.Int{p1, p2})
	if err != nil {
		return [3][2]*big.Int{}, err
	}
	defer Free(h)
	return MSMG2(h, []uint64{1, 0, 0, 0, 1, 0, 0, 0}, 0)
}

// Pairing is bn128.Pairing (bn128/bn128.go:179-186): e(g1, g2) as the reference's Fq12 value, 12 coefficients in the order
// [[[c000, c001], [c010, c011], [c020, c021]], [[c100, ...], ...]] flattened.  Host code: needs no Init.
func Pairing(g1 [3]*big.Int, g2 [3][2]*big.Int) ([12]*big.Int, error) {
	var res [12]*big.Int
	a, err := G1Points([][3]*big.Int{g1})
	if err != nil {
		return res, err
	}
	b, err := G2Points([][3][2]*big.Int{g2})
	if err != nil {
		return res, err
	}
	out := make([]uint64, 48)
	err = call(func() C.int { return C.gs_pairing(ptr(a), ptr(b), ptr(out)) })
	runtime.KeepAlive(a)
	runtime.KeepAlive(b)
	if err != nil {
		return res, err
	}
	for i := range res {
		res[i] = word(out[4*i:])
	}
	return res, nil
}

// ---- pipelined MSMs: at most three tickets (proofs or MSMs) outstanding per logical device -----------------------------------

// MSMTicket is an MSM whose device work is enqueued but not collected.
type MSMTicket struct {
	t  uint64
	g2 bool
}

// MSMG1Begin / MSMG2Begin enqueue sum_i scalars[soff+i] * bases[off+i] over resident operands and return at once.
func MSMG1Begin(bases Handle, off int, scalars Handle, soff, n int) (MSMTicket, error) {
	var t C.uint64_t
	err := call(func() C.int {
		return C.gs_msm_g1_begin(C.gs_handle(bases), C.size_t(off), C.gs_handle(scalars), C.size_t(soff), C.size_t(n), &t)
	})
	return MSMTicket{uint64(t), false}, err
}
func MSMG2Begin(bases Handle, off int, scalars Handle, soff, n int) (MSMTicket, error) {
	var t C.uint64_t
	err := call(func() C.int {
		return C.gs_msm_g2_begin(C.gs_handle(bases), C.size_t(off), C.gs_handle(scalars), C.size_t(soff), C.size_t(n), &t)
	})
	return MSMTicket{uint64(t), true}, err
}

// End waits for that MSM only.  The result is [x, y, 1] (G1) in g1 or [[x0,x1],[y0,y1],[1,0]] in g2, whichever the ticket is.
func (k MSMTicket) End() (g1 [3]*big.Int, g2 [3][2]*big.Int, err error) {
	var out [16]uint64
	var inf C.int
	err = call(func() C.int { return C.gs_msm_end(C.uint64_t(k.t), (*C.uint64_t)(unsafe.Pointer(&out[0])), &inf) })
	if err != nil {
		return
	}
	if k.g2 {
		g2 = G2FromAffine(out[:], inf != 0)
	} else {
		g1 = G1FromAffine(out[:8], inf != 0)
	}
	return
}

// CancelTicket abandons any outstanding ticket (proof or MSM) without its result: error paths that cannot call the matching
// End must, or the slot stays occupied (gs_ticket_cancel).
func CancelTicket(ticket uint64) error { return call(func() C.int { return C.gs_ticket_cancel(C.uint64_t(ticket)) }) }

// Cancel abandons this MSM.
func (k MSMTicket) Cancel() error { return CancelTicket(k.t) }

// MSMG1Resident / MSMG2Resident: blocking MSM over resident operands.
func MSMG1Resident(bases Handle, off int, scalars Handle, soff, n int) ([3]*big.Int, error) {
	var out [8]uint64
	var inf C.int
	err := call(func() C.int {
		return C.gs_msm_g1_resident(C.gs_handle(bases), C.size_t(off), C.gs_handle(scalars), C.size_t(soff), C.size_t(n), (*C.uint64_t)(unsafe.Pointer(&out[0])), &inf)
	})
	return G1FromAffine(out[:], inf != 0), err
}
func MSMG2Resident(bases Handle, off int, scalars Handle, soff, n int) ([3][2]*big.Int, error) {
	var out [16]uint64
	var inf C.int
	err := call(func() C.int {
		return C.gs_msm_g2_resident(C.gs_handle(bases), C.size_t(off), C.gs_handle(scalars), C.size_t(soff), C.size_t(n), (*C.uint64_t)(unsafe.Pointer(&out[0])), &inf)
	})
	return G2FromAffine(out[:], inf != 0), err
}

// ---- one process per GPU (communicator of CommInitRank): the sharded entry points -----------------------------------------

// MSMG1Sharded / MSMG2Sharded: this rank's shard of the bases and scalars; the partial points are gathered inside the library
// (ncclAllGather of 72 / 136-byte records) and every rank returns the same sum.
func MSMG1Sharded(bases, scalars Handle) ([3]*big.Int, error) {
	var out [8]uint64
	var inf C.int
	err := call(func() C.int { return C.gs_msm_g1_sharded(C.gs_handle(bases), C.gs_handle(scalars), (*C.uint64_t)(unsafe.Pointer(&out[0])), &inf) })
	return G1FromAffine(out[:], inf != 0), err
}
func MSMG2Sharded(bases, scalars Handle) ([3][2]*big.Int, error) {
	var out [16]uint64
	var inf C.int
	err := call(func() C.int { return C.gs_msm_g2_sharded(C.gs_handle(bases), C.gs_handle(scalars), (*C.uint64_t)(unsafe.Pointer(&out[0])), &inf) })
	return G2FromAffine(out[:], inf != 0), err
}

// ScatterScalars is the owner's scatter of the values route (gs_scalars_scatter): rank `root` holds `total` scalars in `full`
// (ignored elsewhere); every rank receives its slice of the contiguous split (slice = 0 creates the vector).
func ScatterScalars(full Handle, total, root int, slice Handle) (Handle, error) {
	h := C.gs_handle(slice)
	err := call(func() C.int { return C.gs_scalars_scatter(C.gs_handle(full), C.size_t(total), C.int(root), &h) })
	return Handle(h), err
}

// ProveSharded: this rank's slice of the key, the replicated witness and px; the 416-byte records of the five sums are gathered
// inside the library and every rank returns the same proof (gs_groth16_prove_sharded).
func (k *Groth16Key) ProveSharded(w, px Handle, r, s, order *big.Int) (piA [3]*big.Int, piB [3][2]*big.Int, piC [3]*big.Int, err error) {
	rs, err := Scalars([]*big.Int{r, s}, order)
	if err != nil {
		return
	}
	var out [32]uint64
	var inf [3]C.int
	err = call(func() C.int {
		return C.gs_groth16_prove_sharded(C.gs_handle(k.h), C.gs_handle(w), C.gs_handle(px), ptr(rs[0:]), ptr(rs[4:]), (*C.uint64_t)(unsafe.Pointer(&out[0])), &inf[0])
	})
	runtime.KeepAlive(rs)
	if err != nil {
		return
	}
	piA, piB, piC = groth16ProofFromWords(out[:], inf[:])
	return
}

// WitnessValues is the owner's polynomial stage of the values route (gs_groth16_witness_values): resident R1CS + witness -> the n
// values H(n+1..2n), resident (hv = 0 creates the vector).  violated != 0: the witness breaks a constraint, the values are void.
func (k *Groth16Key) WitnessValues(q *R1CS, w, hv Handle) (Handle, uint32, error) {
	h := C.gs_handle(hv)
	var bad C.uint32_t
	err := call(func() C.int { return C.gs_groth16_witness_values(C.gs_handle(k.h), C.gs_handle(q.h), C.gs_handle(w), &h, &bad) })
	return Handle(h), uint32(bad), err
}

// ProveShardedValues: values route, one process per GPU: hvSlice = this rank's slice of H's values (ScatterScalars).
func (k *Groth16Key) ProveShardedValues(w, hvSlice Handle, r, s, order *big.Int) (piA [3]*big.Int, piB [3][2]*big.Int, piC [3]*big.Int, err error) {
	rs, err := Scalars([]*big.Int{r, s}, order)
	if err != nil {
		return
	}
	var out [32]uint64
	var inf [3]C.int
	err = call(func() C.int {
		return C.gs_groth16_prove_sharded_values(C.gs_handle(k.h), C.gs_handle(w), C.gs_handle(hvSlice), ptr(rs[0:]), ptr(rs[4:]), (*C.uint64_t)(unsafe.Pointer(&out[0])), &inf[0])
	})
	runtime.KeepAlive(rs)
	if err != nil {
		return
	}
	piA, piB, piC = groth16ProofFromWords(out[:], inf[:])
	return
}

// ---- evaluation-basis keys and pipelined witness proofs ---------------------------------------------------------------------

// SetEvalBasis attaches an evaluation-basis copy of PowersTauDelta (n = #constraints points E[j-1] = l_j(tau) Z(tau)/delta G, read
// from a key file) to a resident key: the witness route then runs its h-MSM over H's values (gs_groth16_pk_set_eval).
func (k *Groth16Key) SetEvalBasis(points [][3]*big.Int) error {
	h, err := UploadG1(DeviceOf(k.h), points)
	if err != nil {
		return err
	}
	defer Free(h)
	return call(func() C.int { return C.gs_groth16_pk_set_eval(C.gs_handle(k.h), C.gs_handle(h)) })
}

// EvalBasisCount is the number of evaluation-basis points the key holds (0 = none).
func (k *Groth16Key) EvalBasisCount() (int, error) {
	var n C.size_t
	err := call(func() C.int { return C.gs_pk_eval_count(C.gs_handle(k.h), &n) })
	return int(n), err
}

// SetEvalBasis for a Pinocchio key: E[j-1] = l_j(tau) G (gs_pinocchio_pk_set_eval).
func (k *PinocchioKey) SetEvalBasis(points [][3]*big.Int) error {
	h, err := UploadG1(DeviceOf(k.h), points)
	if err != nil {
		return err
	}
	defer Free(h)
	return call(func() C.int { return C.gs_pinocchio_pk_set_eval(C.gs_handle(k.h), C.gs_handle(h)) })
}

// ProveWitnessBegin enqueues one witness -> proof and returns its ticket (collect with ProveEnd; abandon with CancelTicket).
func (k *Groth16Key) ProveWitnessBegin(q *R1CS, w Handle, r, s, order *big.Int) (Groth16Ticket, error) {
	rs, err := Scalars([]*big.Int{r, s}, order)
	if err != nil {
		return 0, err
	}
	var t C.uint64_t
	err = call(func() C.int {
		return C.gs_groth16_prove_witness_begin(C.gs_handle(k.h), C.gs_handle(q.h), C.gs_handle(w), ptr(rs[0:]), ptr(rs[4:]), &t)
	})
	runtime.KeepAlive(rs)
	return Groth16Ticket(t), err
}

// ProveBegin / ProveWitnessBegin / PinocchioProveEnd: pipelined Pinocchio proofs (the three slots are shared with Groth16 and MSMs).
func (k *PinocchioKey) ProveBegin(w, px Handle) (uint64, error) {
	var t C.uint64_t
	err := call(func() C.int { return C.gs_pinocchio_prove_begin(C.gs_handle(k.h), C.gs_handle(w), C.gs_handle(px), &t) })
	return uint64(t), err
}
func (k *PinocchioKey) ProveWitnessBegin(q *R1CS, w Handle) (uint64, error) {
	var t C.uint64_t
	err := call(func() C.int { return C.gs_pinocchio_prove_witness_begin(C.gs_handle(k.h), C.gs_handle(q.h), C.gs_handle(w), &t) })
	return uint64(t), err
}
func PinocchioProveEnd(ticket uint64) (PinocchioProof, error) {
	var out [72]uint64
	var inf [8]C.int
	err := call(func() C.int { return C.gs_pinocchio_prove_end(C.uint64_t(ticket), (*C.uint64_t)(unsafe.Pointer(&out[0])), &inf[0]) })
	if err != nil {
		return PinocchioProof{}, err
	}
	return pinocchioProofFromWords(out[:], inf[:]), nil
}

// ProveResident: snark.GenerateProofs with w and px already resident.
func (k *PinocchioKey) ProveResident(w, px Handle) (PinocchioProof, error) {
	var out [72]uint64
	var inf [8]C.int
	err := call(func() C.int {
		return C.gs_pinocchio_prove_resident(C.gs_handle(k.h), C.gs_handle(w), C.gs_handle(px), (*C.uint64_t)(unsafe.Pointer(&out[0])), &inf[0])
	})
	if err != nil {
		return PinocchioProof{}, err
	}
	return pinocchioProofFromWords(out[:], inf[:]), nil
}

// SetWindowBits (0 = automatic), SetEvalBasisRoute and VerifySetStrict are the library's process-wide tunables.
func SetWindowBits(c int) error { return call(func() C.int { return C.gs_set_window_bits(C.int(c)) }) }
func SetEvalBasisRoute(on bool) error {
	v := C.int(0)
	if on {
		v = 1
	}
	return call(func() C.int { return C.gs_set_eval_basis(v) })
}
func VerifySetStrict(on bool) {
	v := C.int(0)
	if on {
		v = 1
	}
	C.gs_verify_set_strict(v)
}

// ---- the rest of the header, so that every entry point of include/gosnark_hip.h has a Go name ---------------------------------

// Len is the element count of a resident base array or scalar vector.
func Len(h Handle) (int, error) {
	var n C.size_t
	err := call(func() C.int { return C.gs_len(C.gs_handle(h), &n) })
	return int(n), err
}

// CurrentDevice is the calling OS thread's current logical device (gs_get_device).
func CurrentDevice() int { return int(C.gs_get_device()) }

// DownloadG1 / DownloadG2 read a resident base array back as affine Jacobian triples.
func DownloadG1(h Handle, n int) ([][3]*big.Int, error) {
	buf := make([]uint64, 12*n)
	if err := call(func() C.int { return C.gs_g1_download(C.gs_handle(h), ptr(buf), C.size_t(n)) }); err != nil {
		return nil, err
	}
	out := make([][3]*big.Int, n)
	for i := range out {
		out[i] = G1FromJacobian(buf[12*i:])
	}
	return out, nil
}
func DownloadG2(h Handle, n int) ([][3][2]*big.Int, error) {
	buf := make([]uint64, 24*n)
	if err := call(func() C.int { return C.gs_g2_download(C.gs_handle(h), ptr(buf), C.size_t(n)) }); err != nil {
		return nil, err
	}
	out := make([][3][2]*big.Int, n)
	for i := range out {
		out[i] = G2FromJacobian(buf[24*i:])
	}
	return out, nil
}

// FixedBaseG1 / FixedBaseG2: k_i * G for every scalar, resident (the MulScalar(Utils.Bn.G1.G, k) loops of the trusted setups,
// groth16.go:139-175): synthetic keys and tests.
func FixedBaseG1(device int, scalars []*big.Int, order *big.Int) (Handle, error) {
	buf, err := Scalars(scalars, order)
	if err != nil {
		return 0, err
	}
	var h C.gs_handle
	err = onDevice(device, func() C.int { return C.gs_g1_fixed_base(ptr(buf), C.size_t(len(scalars)), &h) })
	runtime.KeepAlive(buf)
	return Handle(h), err
}
func FixedBaseG2(device int, scalars []*big.Int, order *big.Int) (Handle, error) {
	buf, err := Scalars(scalars, order)
	if err != nil {
		return 0, err
	}
	var h C.gs_handle
	err = onDevice(device, func() C.int { return C.gs_g2_fixed_base(ptr(buf), C.size_t(len(scalars)), &h) })
	runtime.KeepAlive(buf)
	return Handle(h), err
}

// Timing mirrors gs_timing: device time of the last prove / MSM call on a logical device (HIP events on the library's streams).
type Timing struct {
	TotalMs, PlanMs, AccumulateMs, ReduceMs, PolyMs, H2DMs, AccG1Ms, AccG2Ms float32
	AccG1Launches, AccG2Launches                                               uint32
	AccG1Terms, AccG2Terms, AccG1Adds, AccG2Adds                               uint64
	WindowBits, Fallbacks                                                      uint32
	PlanDigits, PlanEntries                                                    uint64 // digits / non-zero digits (= real bucket additions) x base arrays
	HeavyBuckets                                                               uint32
}

func timingFromC(t *C.gs_timing) Timing {
	return Timing{float32(t.total_ms), float32(t.plan_ms), float32(t.accumulate_ms), float32(t.reduce_ms), float32(t.poly_ms), float32(t.h2d_ms),
		float32(t.acc_g1_ms), float32(t.acc_g2_ms), uint32(t.acc_g1_launches), uint32(t.acc_g2_launches), uint64(t.acc_g1_terms), uint64(t.acc_g2_terms),
		uint64(t.acc_g1_adds), uint64(t.acc_g2_adds), uint32(t.window_bits), uint32(t.fallbacks),
		uint64(t.plan_digits), uint64(t.plan_entries), uint32(t.heavy_buckets)}
}

// LastTiming (current logical device of the calling thread) / DeviceTiming (a named one).
func LastTiming() (Timing, error) {
	var t C.gs_timing
	err := call(func() C.int { return C.gs_last_timing(&t) })
	return timingFromC(&t), err
}
func DeviceTiming(device int) (Timing, error) {
	var t C.gs_timing
	err := call(func() C.int { return C.gs_device_timing(C.int(device), &t) })
	return timingFromC(&t), err
}

// CommInfo reports the communicator: ranks, this process's rank (-1 in local mode), local mode, collectives completed.
func CommInfo() (nranks, rank int, local bool, collectives uint64) {
	var n, r, l C.int
	var c C.uint64_t
	C.gs_comm_info(&n, &r, &l, &c)
	return int(n), int(r), l != 0, uint64(c)
}

// CommAllGather gathers `send` (the same length on every rank) from all ranks: recv gets nranks blocks.
func CommAllGather(send []byte, nranks int) ([]byte, error) {
	if len(send) == 0 {
		return nil, nil
	}
	recv := make([]byte, len(send)*nranks)
	err := call(func() C.int { return C.gs_comm_allgather(unsafe.Pointer(&send[0]), C.size_t(len(send)), unsafe.Pointer(&recv[0])) })
	runtime.KeepAlive(send)
	return recv, err
}

// Partials holds a rank's five raw sums (gs_groth16_prove_partials layout: At | G1.BACGamma | G2.BACGamma | BACDelta | h) and flags.
type Partials struct {
	Sums [48]uint64
	Inf  [5]int32
}

func partialsCall(f func(out *C.uint64_t, inf *C.int) C.int) (Partials, error) {
	var p Partials
	var inf [5]C.int
	err := call(func() C.int { return f((*C.uint64_t)(unsafe.Pointer(&p.Sums[0])), &inf[0]) })
	for i := range inf {
		p.Inf[i] = int32(inf[i])
	}
	return p, err
}

// ProvePartials: the five sums over this rank's term ranges, px route (gs_groth16_prove_partials).
func (k *Groth16Key) ProvePartials(w, px Handle, shard, count int) (Partials, error) {
	return partialsCall(func(out *C.uint64_t, inf *C.int) C.int {
		return C.gs_groth16_prove_partials(C.gs_handle(k.h), C.gs_handle(w), C.gs_handle(px), C.size_t(shard), C.size_t(count), out, inf)
	})
}

// ProvePartialsValues: the same on the values route (hvSlice = this rank's slice of H's values); PartialsValuesBegin / PartialsEnd
// are the pipelined form (three in flight per device).
func (k *Groth16Key) ProvePartialsValues(w, hvSlice Handle, shard, count int) (Partials, error) {
	return partialsCall(func(out *C.uint64_t, inf *C.int) C.int {
		return C.gs_groth16_prove_partials_values(C.gs_handle(k.h), C.gs_handle(w), C.gs_handle(hvSlice), C.size_t(shard), C.size_t(count), out, inf)
	})
}
func (k *Groth16Key) PartialsValuesBegin(w, hvSlice Handle, shard, count int) (uint64, error) {
	var t C.uint64_t
	err := call(func() C.int {
		return C.gs_groth16_partials_values_begin(C.gs_handle(k.h), C.gs_handle(w), C.gs_handle(hvSlice), C.size_t(shard), C.size_t(count), &t)
	})
	return uint64(t), err
}
func PartialsEnd(ticket uint64) (Partials, error) {
	return partialsCall(func(out *C.uint64_t, inf *C.int) C.int { return C.gs_groth16_partials_end(C.uint64_t(ticket), out, inf) })
}

// SumAffineG1 / SumAffineG2: the exchange step's complete additions of gathered partial points (host arithmetic).
func SumAffineG1(pts []uint64, inf []int32) ([3]*big.Int, error) {
	n := len(inf)
	ci := make([]C.int, n)
	for i, v := range inf {
		ci[i] = C.int(v)
	}
	var out [8]uint64
	var oi C.int
	err := call(func() C.int { return C.gs_g1_sum_affine(ptr(pts), &ci[0], C.size_t(n), (*C.uint64_t)(unsafe.Pointer(&out[0])), &oi) })
	runtime.KeepAlive(pts)
	return G1FromAffine(out[:], oi != 0), err
}
func SumAffineG2(pts []uint64, inf []int32) ([3][2]*big.Int, error) {
	n := len(inf)
	ci := make([]C.int, n)
	for i, v := range inf {
		ci[i] = C.int(v)
	}
	var out [16]uint64
	var oi C.int
	err := call(func() C.int { return C.gs_g2_sum_affine(ptr(pts), &ci[0], C.size_t(n), (*C.uint64_t)(unsafe.Pointer(&out[0])), &oi) })
	runtime.KeepAlive(pts)
	return G2FromAffine(out[:], oi != 0), err
}

// Finish applies the O(1) tail of groth16.go:253-275 to combined sums (gs_groth16_finish).
func (k *Groth16Key) Finish(p Partials, r, s, order *big.Int) (piA [3]*big.Int, piB [3][2]*big.Int, piC [3]*big.Int, err error) {
	rs, err := Scalars([]*big.Int{r, s}, order)
	if err != nil {
		return
	}
	var inf5 [5]C.int
	for i, v := range p.Inf {
		inf5[i] = C.int(v)
	}
	var out [32]uint64
	var inf [3]C.int
	err = call(func() C.int {
		return C.gs_groth16_finish(C.gs_handle(k.h), (*C.uint64_t)(unsafe.Pointer(&p.Sums[0])), &inf5[0], ptr(rs[0:]), ptr(rs[4:]), (*C.uint64_t)(unsafe.Pointer(&out[0])), &inf[0])
	})
	runtime.KeepAlive(rs)
	if err != nil {
		return
	}
	piA, piB, piC = groth16ProofFromWords(out[:], inf[:])
	return
}

// ShardLocal cuts slice `index` of `count` out of a resident full key on the SAME logical device (gs_groth16_pk_shard).
func (k *Groth16Key) ShardLocal(index, count int) (*Groth16Key, error) {
	var h C.gs_handle
	err := call(func() C.int { return C.gs_groth16_pk_shard(C.gs_handle(k.h), C.size_t(index), C.size_t(count), &h) })
	if err != nil {
		return nil, err
	}
	return &Groth16Key{Handle(h), k.NVars, k.NPublic}, nil
}

// NewGroth16KeyShard assembles a key slice from base handles that hold exactly the slices (gs_groth16_pk_create_shard): a rank
// that loads only its share of a key file never sees the full key.  parts carries the single elements, Z, NVars, NPublic.
func NewGroth16KeyShard(at, bacGamma1, bacGamma2, bacDelta, ptd Handle, parts Groth16KeyParts, nptdTotal, index, count int, order *big.Int) (*Groth16Key, error) {
	singles1, err := G1Points([][3]*big.Int{parts.Alpha, parts.Beta, parts.Delta})
	if err != nil {
		return nil, err
	}
	singles2, err := G2Points([][3][2]*big.Int{parts.Beta2, parts.Delta2})
	if err != nil {
		return nil, err
	}
	z, err := Scalars(parts.Z, order)
	if err != nil {
		return nil, err
	}
	var h C.gs_handle
	err = call(func() C.int {
		return C.gs_groth16_pk_create_shard(C.gs_handle(at), C.gs_handle(bacGamma1), C.gs_handle(bacGamma2), C.gs_handle(bacDelta), C.gs_handle(ptd),
			ptr(singles1[0:]), ptr(singles1[12:]), ptr(singles1[24:]), ptr(singles2[0:]), ptr(singles2[24:]), ptr(z), C.size_t(len(parts.Z)),
			C.size_t(parts.NVars), C.size_t(parts.NPublic), C.size_t(nptdTotal), C.size_t(index), C.size_t(count), &h)
	})
	runtime.KeepAlive(singles1)
	runtime.KeepAlive(singles2)
	runtime.KeepAlive(z)
	if err != nil {
		return nil, err
	}
	return &Groth16Key{Handle(h), parts.NVars, parts.NPublic}, nil
}

// ProveResident: groth16.GenerateProofs with w and px already resident (what bench.py times, blocking).
func (k *Groth16Key) ProveResident(w, px Handle, r, s, order *big.Int) (piA [3]*big.Int, piB [3][2]*big.Int, piC [3]*big.Int, err error) {
	rs, err := Scalars([]*big.Int{r, s}, order)
	if err != nil {
		return
	}
	var out [32]uint64
	var inf [3]C.int
	err = call(func() C.int {
		return C.gs_groth16_prove_resident(C.gs_handle(k.h), C.gs_handle(w), C.gs_handle(px), ptr(rs[0:]), ptr(rs[4:]), (*C.uint64_t)(unsafe.Pointer(&out[0])), &inf[0])
	})
	runtime.KeepAlive(rs)
	if err != nil {
		return
	}
	piA, piB, piC = groth16ProofFromWords(out[:], inf[:])
	return
}

// ProveR1CS: px from the resident sparse system and the proof in one call (gs_groth16_prove_r1cs); px = 0 creates the vector.
func (k *Groth16Key) ProveR1CS(q *R1CS, w, px Handle, r, s, order *big.Int) (piA [3]*big.Int, piB [3][2]*big.Int, piC [3]*big.Int, pxOut Handle, err error) {
	rs, err := Scalars([]*big.Int{r, s}, order)
	if err != nil {
		return
	}
	var out [32]uint64
	var inf [3]C.int
	h := C.gs_handle(px)
	err = call(func() C.int {
		return C.gs_groth16_prove_r1cs(C.gs_handle(k.h), C.gs_handle(q.h), C.gs_handle(w), &h, ptr(rs[0:]), ptr(rs[4:]), (*C.uint64_t)(unsafe.Pointer(&out[0])), &inf[0])
	})
	runtime.KeepAlive(rs)
	pxOut = Handle(h)
	if err != nil {
		return
	}
	piA, piB, piC = groth16ProofFromWords(out[:], inf[:])
	return
}

// ProveMultiValues: one proof over the logical devices of this process on the values route (gs_groth16_prove_multi_values).
func ProveMultiValues(keys []*Groth16Key, w, hvSlices []Handle, r, s, order *big.Int) (piA [3]*big.Int, piB [3][2]*big.Int, piC [3]*big.Int, usedRCCL bool, err error) {
	n := len(keys)
	if n == 0 || len(w) != n || len(hvSlices) != n {
		err = errors.New("gosnark-hip: ProveMultiValues needs one key, w and slice of H's values per device")
		return
	}
	kh := make([]Handle, n)
	for d, k := range keys {
		kh[d] = k.h
	}
	rs, err := Scalars([]*big.Int{r, s}, order)
	if err != nil {
		return
	}
	var out [32]uint64
	var inf [3]C.int
	var used C.int
	err = call(func() C.int {
		return C.gs_groth16_prove_multi_values(handles(kh), handles(w), handles(hvSlices), C.int(n), ptr(rs[0:]), ptr(rs[4:]), (*C.uint64_t)(unsafe.Pointer(&out[0])), &inf[0], &used)
	})
	runtime.KeepAlive(kh)
	runtime.KeepAlive(rs)
	if err != nil {
		return
	}
	piA, piB, piC = groth16ProofFromWords(out[:], inf[:])
	usedRCCL = used != 0
	return
}

// MSMG2Multi is MSMG1Multi over G2.
func MSMG2Multi(bases, scalars []Handle) (p [3][2]*big.Int, usedRCCL bool, err error) {
	if len(bases) == 0 || len(bases) != len(scalars) {
		err = errors.New("gosnark-hip: MSMG2Multi needs one base and one scalar shard per device")
		return
	}
	var out [16]uint64
	var inf, used C.int
	err = call(func() C.int {
		return C.gs_msm_g2_multi(handles(bases), handles(scalars), C.int(len(bases)), (*C.uint64_t)(unsafe.Pointer(&out[0])), &inf, &used)
	})
	runtime.KeepAlive(bases)
	runtime.KeepAlive(scalars)
	return G2FromAffine(out[:], inf != 0), used != 0, err
}
