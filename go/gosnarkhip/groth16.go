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

// Groth16Key is a proving key resident in HBM (groth16.Pk, groth16/groth16.go:15-32) on one logical device.
type Groth16Key struct {
	h              Handle
	NVars, NPublic int
}

// Handle exposes the raw gs_handle (for the multi-device entry points).
func (k *Groth16Key) Handle() Handle { return k.h }

// Free releases the key's HBM (window tables included).  Safe while proofs that use it are in flight.
func (k *Groth16Key) Free() error {
	h := k.h
	k.h = 0
	return Free(h)
}

// Groth16KeyParts carries the reference's Pk fields without importing the groth16 package.
type Groth16KeyParts struct {
	At, BACGamma1, BACDelta, PowersTauDelta [][3]*big.Int
	BACGamma2                               [][3][2]*big.Int
	Alpha, Beta, Delta                      [3]*big.Int
	Beta2, Delta2                           [3][2]*big.Int
	Z                                       []*big.Int
	NVars, NPublic                          int
}

// Groth16VkParts carries the reference's groth16.Vk fields (groth16/groth16.go:33-43).
type Groth16VkParts struct {
	IC      [][3]*big.Int
	G1Alpha [3]*big.Int
	G2Beta  [3][2]*big.Int
	G2Gamma [3][2]*big.Int
	G2Delta [3][2]*big.Int
}

// NewGroth16Key uploads the key once per circuit (SURVEY.md hard part 4: never per proof) onto logical device
// `device`.  Call sequence = tests/c/groth16_generateproofs.c: 5 x gs_g*_upload, gs_groth16_pk_create, 5 x gs_free.
func NewGroth16Key(device int, p Groth16KeyParts, r *big.Int) (*Groth16Key, error) {
	var hs [5]Handle
	defer func() {
		for _, h := range hs {
			_ = Free(h)
		}
	}()
	var err error
	if hs[0], err = UploadG1(device, p.At); err != nil {
		return nil, err
	}
	if hs[1], err = UploadG1(device, p.BACGamma1); err != nil {
		return nil, err
	}
	if hs[2], err = UploadG2(device, p.BACGamma2); err != nil {
		return nil, err
	}
	if hs[3], err = UploadG1(device, p.BACDelta); err != nil {
		return nil, err
	}
	if hs[4], err = UploadG1(device, p.PowersTauDelta); err != nil {
		return nil, err
	}
	singles1, err := G1Points([][3]*big.Int{p.Alpha, p.Beta, p.Delta})
	if err != nil {
		return nil, err
	}
	singles2, err := G2Points([][3][2]*big.Int{p.Beta2, p.Delta2})
	if err != nil {
		return nil, err
	}
	z, err := Scalars(p.Z, r)
	if err != nil {
		return nil, err
	}
	var h C.gs_handle
	err = call(func() C.int {
		return C.gs_groth16_pk_create(C.gs_handle(hs[0]), C.gs_handle(hs[1]), C.gs_handle(hs[2]), C.gs_handle(hs[3]), C.gs_handle(hs[4]),
			ptr(singles1[0:]), ptr(singles1[12:]), ptr(singles1[24:]), ptr(singles2[0:]), ptr(singles2[24:]),
			ptr(z), C.size_t(len(p.Z)), C.size_t(p.NVars), C.size_t(p.NPublic), &h)
	})
	runtime.KeepAlive(singles1)
	runtime.KeepAlive(singles2)
	runtime.KeepAlive(z)
	if err != nil {
		return nil, err
	}
	return &Groth16Key{Handle(h), p.NVars, p.NPublic}, nil
}

func groth16ProofFromWords(out []uint64, inf []C.int) (piA [3]*big.Int, piB [3][2]*big.Int, piC [3]*big.Int) {
	return G1FromAffine(out[0:8], inf[0] != 0), G2FromAffine(out[8:24], inf[1] != 0), G1FromAffine(out[24:32], inf[2] != 0)
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
	err = call(func() C.int {
		return C.gs_groth16_prove(C.gs_handle(k.h), ptr(wb), C.size_t(len(w)), ptr(pb), C.size_t(len(px)),
			ptr(rs[0:]), ptr(rs[4:]), (*C.uint64_t)(unsafe.Pointer(&out[0])), &inf[0])
	})
	runtime.KeepAlive(wb)
	runtime.KeepAlive(pb)
	runtime.KeepAlive(rs)
	if err != nil {
		return
	}
	piA, piB, piC = groth16ProofFromWords(out[:], inf[:])
	return
}

// Groth16Ticket is one proof in flight (gs_groth16_prove_begin): up to three per logical device.
type Groth16Ticket uint64

// ProveBegin enqueues one proof over resident inputs and returns at once; ProveEnd collects it.  A stream of
// proofs with two or three tickets outstanding keeps the GPU's accumulation pipeline full.
func (k *Groth16Key) ProveBegin(w, px Handle, r, s, order *big.Int) (Groth16Ticket, error) {
	rs, err := Scalars([]*big.Int{r, s}, order)
	if err != nil {
		return 0, err
	}
	var t C.uint64_t
	err = call(func() C.int { return C.gs_groth16_prove_begin(C.gs_handle(k.h), C.gs_handle(w), C.gs_handle(px), ptr(rs[0:]), ptr(rs[4:]), &t) })
	runtime.KeepAlive(rs)
	return Groth16Ticket(t), err
}

// ProveEnd waits for that proof only and returns it.
func ProveEnd(t Groth16Ticket) (piA [3]*big.Int, piB [3][2]*big.Int, piC [3]*big.Int, err error) {
	var out [32]uint64
	var inf [3]C.int
	err = call(func() C.int { return C.gs_groth16_prove_end(C.uint64_t(t), (*C.uint64_t)(unsafe.Pointer(&out[0])), &inf[0]) })
	if err != nil {
		return
	}
	piA, piB, piC = groth16ProofFromWords(out[:], inf[:])
	return
}

func (k *Groth16Key) exportG1(which, n int) ([][3]*big.Int, error) {
	buf := make([]uint64, 12*n+12)
	err := call(func() C.int { return C.gs_groth16_pk_export(C.gs_handle(k.h), C.int(which), ptr(buf), C.size_t(n)) })
	if err != nil {
		return nil, err
	}
	out := make([][3]*big.Int, n)
	for i := range out {
		out[i] = G1FromJacobian(buf[12*i:])
	}
	return out, nil
}

// Export reads the resident key back as the reference's Pk fields (affine triples [x, y, 1]): what
// GenerateTrustedSetup must return to its caller (cli/main.go:280 writes it to trustedsetup.json).
func (k *Groth16Key) Export(nz int) (Groth16KeyParts, error) {
	p := Groth16KeyParts{NVars: k.NVars, NPublic: k.NPublic}
	var err error
	if p.At, err = k.exportG1(0, k.NVars); err != nil {
		return p, err
	}
	if p.BACGamma1, err = k.exportG1(1, k.NVars); err != nil {
		return p, err
	}
	b2 := make([]uint64, 24*k.NVars+24)
	if err = call(func() C.int { return C.gs_groth16_pk_export(C.gs_handle(k.h), 2, ptr(b2), C.size_t(k.NVars)) }); err != nil {
		return p, err
	}
	p.BACGamma2 = make([][3][2]*big.Int, k.NVars)
	for i := range p.BACGamma2 {
		p.BACGamma2[i] = G2FromJacobian(b2[24*i:])
	}
	if p.BACDelta, err = k.exportG1(3, k.NVars); err != nil {
		return p, err
	}
	if p.PowersTauDelta, err = k.exportG1(4, nz); err != nil { // len(PowersTauDelta) == len(Z) (groth16.go:139-147)
		return p, err
	}
	single := make([]uint64, 3*12+2*24)
	if err = call(func() C.int { return C.gs_groth16_pk_export(C.gs_handle(k.h), 5, ptr(single), 5) }); err != nil {
		return p, err
	}
	p.Alpha, p.Beta, p.Delta = G1FromJacobian(single[0:]), G1FromJacobian(single[12:]), G1FromJacobian(single[24:])
	p.Beta2, p.Delta2 = G2FromJacobian(single[36:]), G2FromJacobian(single[60:])
	zb := make([]uint64, 4*nz)
	if err = call(func() C.int { return C.gs_groth16_pk_export(C.gs_handle(k.h), 6, ptr(zb), C.size_t(nz)) }); err != nil {
		return p, err
	}
	p.Z = unpackScalars(zb)
	return p, nil
}

// Groth16Toxic = the five values Utils.FqR.Rand() returns at groth16.go:99-119, in that order.
type Groth16Toxic struct{ T, Kalpha, Kbeta, Kgamma, Kdelta *big.Int }

// Groth16Setup is groth16.GenerateTrustedSetup (groth16.go:94-222) on the device, from the SPARSE R1CS
// (constraints x variables; the reference's alphas/betas/gammas are the interpolants of its columns) and the
// toxic values drawn by the caller.  Returns the resident key and the verification key.
// Call sequence = tests/c/groth16_setup_prove_verify.c.
func Groth16Setup(device int, a, b, c CSR, nvars, npublic int, tox Groth16Toxic, order *big.Int) (*Groth16Key, Groth16VkParts, error) {
	var vk Groth16VkParts
	n := len(a.RowPtr) - 1
	if n < 1 || len(b.RowPtr) != n+1 || len(c.RowPtr) != n+1 {
		return nil, vk, errors.New("gosnark-hip: A, B, C must have the same number of constraints")
	}
	tb, err := Scalars([]*big.Int{tox.T, tox.Kalpha, tox.Kbeta, tox.Kgamma, tox.Kdelta}, order)
	if err != nil {
		return nil, vk, err
	}
	vkb := make([]uint64, 12+3*24+12*(npublic+1))
	var h C.gs_handle
	err = onDevice(device, func() C.int {
		return C.gs_groth16_setup(C.size_t(n), C.size_t(nvars), C.size_t(npublic),
			ptr32(a.RowPtr), ptr32(a.Col), ptr(a.Val), ptr32(b.RowPtr), ptr32(b.Col), ptr(b.Val), ptr32(c.RowPtr), ptr32(c.Col), ptr(c.Val),
			ptr(tb), &h, ptr(vkb))
	})
	runtime.KeepAlive(tb)
	runtime.KeepAlive(a)
	runtime.KeepAlive(b)
	runtime.KeepAlive(c)
	if err != nil {
		return nil, vk, err
	}
	vk.G1Alpha = G1FromJacobian(vkb[0:])
	vk.G2Beta, vk.G2Gamma, vk.G2Delta = G2FromJacobian(vkb[12:]), G2FromJacobian(vkb[36:]), G2FromJacobian(vkb[60:])
	for i := 0; i <= npublic; i++ {
		vk.IC = append(vk.IC, G1FromJacobian(vkb[84+12*i:]))
	}
	return &Groth16Key{Handle(h), nvars, npublic}, vk, nil
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
	err = call(func() C.int {
		return C.gs_groth16_verify(ptr(g1[0:]), ptr(g2[0:]), ptr(g2[24:]), ptr(g2[48:]), ptr(ic), C.size_t(len(vk.IC)),
			ptr(pub), C.size_t(len(publicSignals)), ptr(g1[12:]), ptr(g2[72:]), ptr(g1[24:]), &ok)
	})
	runtime.KeepAlive(ic)
	runtime.KeepAlive(g1)
	runtime.KeepAlive(g2)
	runtime.KeepAlive(pub)
	return ok == 1, err
}
