// Package gosnarkhip is the cgo binding of libgosnark_hip.so (include/gosnark_hip.h): the MI355X
// implementation of go-snark-study's prover hot path.  It only packs the reference's big.Int
// structures into flat little-endian limb buffers and calls the C ABI; all arithmetic happens in
// the HIP library.  NOTE: the build image has no Go toolchain, so this file is reviewed-not-compiled
// there; the same ABI is exercised from Python (ctypes) by the test-suite.
//
// cgo rules observed: only flat []uint64 / []uint32 buffers cross the boundary, the library copies
// during the call and keeps no Go pointer, device memory lives behind opaque handles.
package gosnarkhip

/*
#cgo CFLAGS: -I${SRCDIR}/../../include
#cgo LDFLAGS: -L${SRCDIR}/../../go-snark-study_amd -lgosnark_hip -Wl,-rpath,${SRCDIR}/../../go-snark-study_amd
#include <stdlib.h>
#include "gosnark_hip.h"
*/
import "C"

import (
	"errors"
	"fmt"
	"math/big"
	"runtime"
	"unsafe"
)

// Handle is an opaque device object (resident base array, scalar vector or proving key).
type Handle uint64

func status(code C.int) error {
	if code == 0 {
		return nil
	}
	return fmt.Errorf("gosnark-hip: status %d: %s", int(code), C.GoString(C.gs_last_error()))
}

// Init selects the GPU this process drives (one process per GPU).
func Init(device int) error {
	d := C.int(device)
	return status(C.gs_init(&d, 1))
}

// Free releases a device object.
func Free(h Handle) { C.gs_free(C.gs_handle(h)) }

// limbs writes v (0 <= v < 2^256) as 4 little-endian 64-bit words: exactly big.Int.Bits() padded.
func limbs(dst []uint64, v *big.Int) error {
	if v.Sign() < 0 {
		return errors.New("gosnark-hip: negative value (the reference drops the sign, fields/fq.go:138-140; rejected here)")
	}
	bits := v.Bits()
	if len(bits) > 4 {
		return errors.New("gosnark-hip: value wider than 256 bits")
	}
	for i := range dst[:4] {
		dst[i] = 0
	}
	for i, w := range bits {
		dst[i] = uint64(w)
	}
	return nil
}

// Scalars packs field elements (reduced mod r first: the reference's witness values are not
// canonical, circuitcompiler/circuit.go:176-182) into n x 4 words.
func Scalars(vals []*big.Int, r *big.Int) ([]uint64, error) {
	out := make([]uint64, 4*len(vals))
	t := new(big.Int)
	for i, v := range vals {
		if v.Sign() < 0 {
			return nil, errors.New("gosnark-hip: negative scalar")
		}
		t.Mod(v, r)
		if err := limbs(out[4*i:], t); err != nil {
			return nil, err
		}
	}
	return out, nil
}

// G1Points packs [][3]*big.Int Jacobian triples (bn128/g1.go:9-12) into n x 12 words.
func G1Points(pts [][3]*big.Int) ([]uint64, error) {
	out := make([]uint64, 12*len(pts))
	for i, p := range pts {
		for k := 0; k < 3; k++ {
			if err := limbs(out[12*i+4*k:], p[k]); err != nil {
				return nil, err
			}
		}
	}
	return out, nil
}

// G2Points packs [][3][2]*big.Int (bn128/g2.go:9-12) into n x 24 words.
func G2Points(pts [][3][2]*big.Int) ([]uint64, error) {
	out := make([]uint64, 24*len(pts))
	for i, p := range pts {
		for k := 0; k < 3; k++ {
			for j := 0; j < 2; j++ {
				if err := limbs(out[24*i+8*k+4*j:], p[k][j]); err != nil {
					return nil, err
				}
			}
		}
	}
	return out, nil
}

func ptr(b []uint64) *C.uint64_t {
	if len(b) == 0 {
		return nil
	}
	return (*C.uint64_t)(unsafe.Pointer(&b[0]))
}

// UploadG1 makes a base-point array resident (affine-normalised on the device).
func UploadG1(pts [][3]*big.Int) (Handle, error) {
	buf, err := G1Points(pts)
	if err != nil {
		return 0, err
	}
	var h C.gs_handle
	err = status(C.gs_g1_upload(ptr(buf), C.size_t(len(pts)), &h))
	runtime.KeepAlive(buf)
	return Handle(h), err
}

// UploadG2 is UploadG1 for G2 arrays.
func UploadG2(pts [][3][2]*big.Int) (Handle, error) {
	buf, err := G2Points(pts)
	if err != nil {
		return 0, err
	}
	var h C.gs_handle
	err = status(C.gs_g2_upload(ptr(buf), C.size_t(len(pts)), &h))
	runtime.KeepAlive(buf)
	return Handle(h), err
}

func word(b []uint64) *big.Int {
	ws := make([]big.Word, 4)
	for i := range ws {
		ws[i] = big.Word(b[i])
	}
	return new(big.Int).SetBits(ws)
}

// G1FromAffine rebuilds the reference's triple [x, y, 1] (or [0,0,0] for infinity) from 8 words.
func G1FromAffine(b []uint64, inf bool) [3]*big.Int {
	if inf {
		return [3]*big.Int{big.NewInt(0), big.NewInt(0), big.NewInt(0)}
	}
	return [3]*big.Int{word(b[0:]), word(b[4:]), big.NewInt(1)}
}

// G2FromAffine rebuilds [[x0,x1],[y0,y1],[1,0]] from 16 words.
func G2FromAffine(b []uint64, inf bool) [3][2]*big.Int {
	z := func() *big.Int { return big.NewInt(0) }
	if inf {
		return [3][2]*big.Int{{z(), z()}, {z(), z()}, {z(), z()}}
	}
	return [3][2]*big.Int{{word(b[0:]), word(b[4:])}, {word(b[8:]), word(b[12:])}, {big.NewInt(1), z()}}
}

// MSMG1 = sum_i scalars[i] * bases[off+i]: the loop of groth16.go:243-250 as one call.
func MSMG1(bases Handle, scalars []uint64, off int) ([3]*big.Int, error) {
	var out [8]uint64
	var inf C.int
	err := status(C.gs_msm_g1(C.gs_handle(bases), ptr(scalars), C.size_t(off), C.size_t(len(scalars)/4),
		(*C.uint64_t)(unsafe.Pointer(&out[0])), &inf))
	runtime.KeepAlive(scalars)
	return G1FromAffine(out[:], inf != 0), err
}

// Groth16Key is a proving key resident in HBM (groth16.Pk, groth16/groth16.go:15-32).
type Groth16Key struct{ h Handle }

// Groth16KeyParts carries the reference's Pk fields without importing the groth16 package.
type Groth16KeyParts struct {
	At, BACGamma1, BACDelta, PowersTauDelta [][3]*big.Int
	BACGamma2                               [][3][2]*big.Int
	Alpha, Beta, Delta                      [3]*big.Int
	Beta2, Delta2                           [3][2]*big.Int
	Z                                       []*big.Int
	NVars, NPublic                          int
}

// NewGroth16Key uploads the key once per circuit (SURVEY.md hard part 4: never per proof).
func NewGroth16Key(p Groth16KeyParts, r *big.Int) (*Groth16Key, error) {
	at, err := UploadG1(p.At)
	if err != nil {
		return nil, err
	}
	defer Free(at)
	b1, err := UploadG1(p.BACGamma1)
	if err != nil {
		return nil, err
	}
	defer Free(b1)
	b2, err := UploadG2(p.BACGamma2)
	if err != nil {
		return nil, err
	}
	defer Free(b2)
	cd, err := UploadG1(p.BACDelta)
	if err != nil {
		return nil, err
	}
	defer Free(cd)
	pt, err := UploadG1(p.PowersTauDelta)
	if err != nil {
		return nil, err
	}
	defer Free(pt)
	singles1, _ := G1Points([][3]*big.Int{p.Alpha, p.Beta, p.Delta})
	singles2, _ := G2Points([][3][2]*big.Int{p.Beta2, p.Delta2})
	z, err := Scalars(p.Z, r)
	if err != nil {
		return nil, err
	}
	var h C.gs_handle
	err = status(C.gs_groth16_pk_create(C.gs_handle(at), C.gs_handle(b1), C.gs_handle(b2), C.gs_handle(cd), C.gs_handle(pt),
		ptr(singles1[0:]), ptr(singles1[12:]), ptr(singles1[24:]), ptr(singles2[0:]), ptr(singles2[24:]),
		ptr(z), C.size_t(len(p.Z)), C.size_t(p.NVars), C.size_t(p.NPublic), &h))
	runtime.KeepAlive(singles1)
	runtime.KeepAlive(singles2)
	runtime.KeepAlive(z)
	if err != nil {
		return nil, err
	}
	k := &Groth16Key{Handle(h)}
	runtime.SetFinalizer(k, func(k *Groth16Key) { Free(k.h) })
	return k, nil
}

// Prove is groth16.GenerateProofs (groth16.go:225-278) with r, s = what Utils.FqR.Rand() returned.
// Returns PiA, PiB, PiC in the affine normal form (G1.Affine / G2.Affine of the reference's result).
func (k *Groth16Key) Prove(w, px []*big.Int, r, s, order *big.Int) (piA [3]*big.Int, piB [3][2]*big.Int, piC [3]*big.Int, err error) {
	wb, err := Scalars(w, order)
	if err != nil {
		return
	}
	pb, err := Scalars(px, order)
	if err != nil {
		return
	}
	rs, err := Scalars([]*big.Int{r, s}, order)
	if err != nil {
		return
	}
	var out [32]uint64
	var inf [3]C.int
	err = status(C.gs_groth16_prove(C.gs_handle(k.h), ptr(wb), C.size_t(len(w)), ptr(pb), C.size_t(len(px)),
		ptr(rs[0:]), ptr(rs[4:]), (*C.uint64_t)(unsafe.Pointer(&out[0])), &inf[0]))
	runtime.KeepAlive(wb)
	runtime.KeepAlive(pb)
	runtime.KeepAlive(rs)
	if err != nil {
		return
	}
	piA = G1FromAffine(out[0:8], inf[0] != 0)
	piB = G2FromAffine(out[8:24], inf[1] != 0)
	piC = G1FromAffine(out[24:32], inf[2] != 0)
	return
}

// PolyDiv is PolynomialField.Div (r1csqap/r1csqap.go:70-84): quotient and remainder.
func PolyDiv(a, b []*big.Int, order *big.Int) (quo, rem []*big.Int, err error) {
	ab, err := Scalars(a, order)
	if err != nil {
		return
	}
	bb, err := Scalars(b, order)
	if err != nil {
		return
	}
	nq, nr := len(a)-len(b)+1, len(b)-1
	if nq < 1 || len(b) == 0 {
		return nil, nil, errors.New("gosnark-hip: PolyDiv needs len(a) >= len(b) >= 1")
	}
	qb := make([]uint64, 4*nq)
	rb := make([]uint64, 4*(nr+1))
	err = status(C.gs_poly_div(ptr(ab), C.size_t(len(a)), ptr(bb), C.size_t(len(b)), ptr(qb), ptr(rb)))
	runtime.KeepAlive(ab)
	runtime.KeepAlive(bb)
	if err != nil {
		return
	}
	for i := 0; i < nq; i++ {
		quo = append(quo, word(qb[4*i:]))
	}
	for i := 0; i < nr; i++ {
		rem = append(rem, word(rb[4*i:]))
	}
	return
}

// Groth16VkParts carries the reference's groth16.Vk fields (groth16/groth16.go:33-43).
type Groth16VkParts struct {
	IC      [][3]*big.Int
	G1Alpha [3]*big.Int
	G2Beta  [3][2]*big.Int
	G2Gamma [3][2]*big.Int
	G2Delta [3][2]*big.Int
}

// Groth16Verify is groth16.VerifyProof (groth16.go:281-305): one 4-pair product check with a shared final
// exponentiation.  Host side (gs_groth16_verify needs no Init and no device).  The reference indexes vk.IC[i+1] for
// every public signal and panics past the end; here that is an error.
func Groth16Verify(vk Groth16VkParts, piA [3]*big.Int, piB [3][2]*big.Int, piC [3]*big.Int, publicSignals []*big.Int, order *big.Int) (bool, error) {
	ic, err := G1Points(vk.IC)
	if err != nil {
		return false, err
	}
	g1, err := G1Points([][3]*big.Int{vk.G1Alpha, piA, piC})
	if err != nil {
		return false, err
	}
	g2, err := G2Points([][3][2]*big.Int{vk.G2Beta, vk.G2Gamma, vk.G2Delta, piB})
	if err != nil {
		return false, err
	}
	pub, err := Scalars(publicSignals, order)
	if err != nil {
		return false, err
	}
	if len(pub) == 0 {
		pub = make([]uint64, 4)
	}
	var ok C.int
	err = status(C.gs_groth16_verify(ptr(g1[0:]), ptr(g2[0:]), ptr(g2[24:]), ptr(g2[48:]), ptr(ic), C.size_t(len(vk.IC)),
		ptr(pub), C.size_t(len(publicSignals)), ptr(g1[12:]), ptr(g2[72:]), ptr(g1[24:]), &ok))
	runtime.KeepAlive(ic)
	runtime.KeepAlive(g1)
	runtime.KeepAlive(g2)
	runtime.KeepAlive(pub)
	return ok == 1, err
}

// PairingCheck reports whether prod_i e(g1[i], g2[i]) == 1 (the seam under both verifiers, bn128/bn128.go:179-186):
// e(A, B) == e(C, D) is PairingCheck([A, -C], [B, D]).
func PairingCheck(g1 [][3]*big.Int, g2 [][3][2]*big.Int) (bool, error) {
	if len(g1) != len(g2) {
		return false, errors.New("gosnark-hip: PairingCheck needs as many G1 as G2 points")
	}
	if len(g1) == 0 {
		return true, nil
	}
	a, err := G1Points(g1)
	if err != nil {
		return false, err
	}
	b, err := G2Points(g2)
	if err != nil {
		return false, err
	}
	var ok C.int
	err = status(C.gs_pairing_check(ptr(a), ptr(b), C.size_t(len(g1)), &ok))
	runtime.KeepAlive(a)
	runtime.KeepAlive(b)
	return ok == 1, err
}
