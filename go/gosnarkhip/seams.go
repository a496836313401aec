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
