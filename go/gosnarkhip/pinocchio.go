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

// PinocchioKey is a snark.Pk (snark.go:16-26) resident in HBM.
type PinocchioKey struct {
	h              Handle
	NVars, NPublic int
}

func (k *PinocchioKey) Handle() Handle { return k.h }

func (k *PinocchioKey) Free() error {
	h := k.h
	k.h = 0
	return Free(h)
}

// PinocchioKeyParts carries snark.Pk's fields.
type PinocchioKeyParts struct {
	G1T, A, C, Kp, Ap, Bp, Cp [][3]*big.Int
	B                         [][3][2]*big.Int
	Z                         []*big.Int
	NVars, NPublic            int
}

// PinocchioVkParts carries snark.Vk's fields (snark.go:28-37).
type PinocchioVkParts struct {
	Vka, Vkc, G2Kbg, G2Kg, Vkz [3][2]*big.Int
	Vkb, G1Kbg                 [3]*big.Int
	IC                         [][3]*big.Int
}

// PinocchioProof carries snark.Proof's fields (snark.go:59-69), affine normal form.
type PinocchioProof struct {
	PiA, PiAp, PiBp, PiC, PiCp, PiH, PiKp [3]*big.Int
	PiB                                   [3][2]*big.Int
}

// NewPinocchioKey uploads a key onto logical device `device`.  Call sequence = tests/c/snark_generateproofs.c:
// 8 x gs_g*_upload, gs_pinocchio_pk_create, 8 x gs_free.
func NewPinocchioKey(device int, p PinocchioKeyParts, r *big.Int) (*PinocchioKey, error) {
	var hs [8]Handle
	defer func() {
		for _, h := range hs {
			_ = Free(h)
		}
	}()
	g1s := [][][3]*big.Int{p.A, p.Ap, nil, p.Bp, p.C, p.Cp, p.Kp, p.G1T} // order of gs_pinocchio_pk_create's arguments
	var err error
	for i, arr := range g1s {
		if i == 2 {
			hs[i], err = UploadG2(device, p.B)
		} else {
			hs[i], err = UploadG1(device, arr)
		}
		if err != nil {
			return nil, err
		}
	}
	z, err := Scalars(p.Z, r)
	if err != nil {
		return nil, err
	}
	var h C.gs_handle
	err = call(func() C.int {
		return C.gs_pinocchio_pk_create(C.gs_handle(hs[0]), C.gs_handle(hs[1]), C.gs_handle(hs[2]), C.gs_handle(hs[3]), C.gs_handle(hs[4]),
			C.gs_handle(hs[5]), C.gs_handle(hs[6]), C.gs_handle(hs[7]), ptr(z), C.size_t(len(p.Z)), C.size_t(p.NVars), C.size_t(p.NPublic), &h)
	})
	runtime.KeepAlive(z)
	if err != nil {
		return nil, err
	}
	return &PinocchioKey{Handle(h), p.NVars, p.NPublic}, nil
}

// Prove is snark.GenerateProofs (snark.go:254-289); deterministic (the reference draws no randomness there).
func (k *PinocchioKey) Prove(w, px []*big.Int, order *big.Int) (PinocchioProof, error) {
	var proof PinocchioProof
	wb, err := Scalars(w, order)
	if err != nil {
		return proof, err
	}
	pb, err := Scalars(px, order)
	if err != nil {
		return proof, err
	}
	var out [72]uint64
	var inf [8]C.int
	err = call(func() C.int {
		return C.gs_pinocchio_prove(C.gs_handle(k.h), ptr(wb), C.size_t(len(w)), ptr(pb), C.size_t(len(px)), (*C.uint64_t)(unsafe.Pointer(&out[0])), &inf[0])
	})
	runtime.KeepAlive(wb)
	runtime.KeepAlive(pb)
	if err != nil {
		return proof, err
	}
	return pinocchioProofFromWords(out[:], inf[:]), nil
}

// out = PiA | PiAp | PiB (16 words) | PiBp | PiC | PiCp | PiH | PiKp
func pinocchioProofFromWords(out []uint64, inf []C.int) (proof PinocchioProof) {
	proof.PiA = G1FromAffine(out[0:], inf[0] != 0)
	proof.PiAp = G1FromAffine(out[8:], inf[1] != 0)
	proof.PiB = G2FromAffine(out[16:], inf[2] != 0)
	proof.PiBp = G1FromAffine(out[32:], inf[3] != 0)
	proof.PiC = G1FromAffine(out[40:], inf[4] != 0)
	proof.PiCp = G1FromAffine(out[48:], inf[5] != 0)
	proof.PiH = G1FromAffine(out[56:], inf[6] != 0)
	proof.PiKp = G1FromAffine(out[64:], inf[7] != 0)
	return proof
}

// PinocchioToxic = the eight values snark.GenerateTrustedSetup draws (snark.go:114-148; RhoC = RhoA RhoB, :149).
type PinocchioToxic struct{ T, Ka, Kb, Kc, Kbeta, Kgamma, RhoA, RhoB *big.Int }

// PinocchioSetup is snark.GenerateTrustedSetup (snark.go:98-251) on the device from the sparse R1CS.
// Call sequence = tests/c/snark_setup_prove_verify.c.
func PinocchioSetup(device int, a, b, c CSR, nvars, npublic int, tox PinocchioToxic, order *big.Int) (*PinocchioKey, PinocchioVkParts, error) {
	var vk PinocchioVkParts
	n := len(a.RowPtr) - 1
	if n < 1 || len(b.RowPtr) != n+1 || len(c.RowPtr) != n+1 {
		return nil, vk, errors.New("gosnark-hip: A, B, C must have the same number of constraints")
	}
	tb, err := Scalars([]*big.Int{tox.T, tox.Ka, tox.Kb, tox.Kc, tox.Kbeta, tox.Kgamma, tox.RhoA, tox.RhoB}, order)
	if err != nil {
		return nil, vk, err
	}
	// Vka (24) | Vkb (12) | Vkc (24) | G1Kbg (12) | G2Kbg (24) | G2Kg (24) | Vkz (24) | IC (12 each)
	vkb := make([]uint64, 144+12*(npublic+1))
	var h C.gs_handle
	err = onDevice(device, func() C.int {
		return C.gs_pinocchio_setup(C.size_t(n), C.size_t(nvars), C.size_t(npublic),
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
	vk.Vka, vk.Vkb, vk.Vkc = G2FromJacobian(vkb[0:]), G1FromJacobian(vkb[24:]), G2FromJacobian(vkb[36:])
	vk.G1Kbg, vk.G2Kbg, vk.G2Kg, vk.Vkz = G1FromJacobian(vkb[60:]), G2FromJacobian(vkb[72:]), G2FromJacobian(vkb[96:]), G2FromJacobian(vkb[120:])
	for i := 0; i <= npublic; i++ {
		vk.IC = append(vk.IC, G1FromJacobian(vkb[144+12*i:]))
	}
	return &PinocchioKey{Handle(h), nvars, npublic}, vk, nil
}

// Export reads the resident key back as snark.Pk's fields.  A and Ap come back with infinity at i <= NPublic:
// those are the entries the prover never reads (snark.go:265-268).
func (k *PinocchioKey) Export(nz int) (PinocchioKeyParts, error) {
	p := PinocchioKeyParts{NVars: k.NVars, NPublic: k.NPublic}
	g1 := func(which, n int) ([][3]*big.Int, error) {
		buf := make([]uint64, 12*n+12)
		if err := call(func() C.int { return C.gs_pinocchio_pk_export(C.gs_handle(k.h), C.int(which), ptr(buf), C.size_t(n)) }); err != nil {
			return nil, err
		}
		out := make([][3]*big.Int, n)
		for i := range out {
			out[i] = G1FromJacobian(buf[12*i:])
		}
		return out, nil
	}
	var err error
	for _, f := range []struct {
		which, n int
		dst      *[][3]*big.Int
	}{{0, k.NVars, &p.A}, {1, k.NVars, &p.Ap}, {3, k.NVars, &p.Bp}, {4, k.NVars, &p.C}, {5, k.NVars, &p.Cp}, {6, k.NVars, &p.Kp}, {7, nz, &p.G1T}} {
		if *f.dst, err = g1(f.which, f.n); err != nil {
			return p, err
		}
	}
	b2 := make([]uint64, 24*k.NVars+24)
	if err = call(func() C.int { return C.gs_pinocchio_pk_export(C.gs_handle(k.h), 2, ptr(b2), C.size_t(k.NVars)) }); err != nil {
		return p, err
	}
	p.B = make([][3][2]*big.Int, k.NVars)
	for i := range p.B {
		p.B[i] = G2FromJacobian(b2[24*i:])
	}
	zb := make([]uint64, 4*nz)
	if err = call(func() C.int { return C.gs_pinocchio_pk_export(C.gs_handle(k.h), 8, ptr(zb), C.size_t(nz)) }); err != nil {
		return p, err
	}
	p.Z = unpackScalars(zb)
	return p, nil
}

// PinocchioVerify is snark.VerifyProof (snark.go:292-368): the five checks in the reference's order; failed = 0 or
// the number (1..5) of the first equation that does not hold.  Host code.
func PinocchioVerify(vk PinocchioVkParts, p PinocchioProof, publicSignals []*big.Int, order *big.Int) (ok bool, failed int, err error) {
	g2, err := G2Points([][3][2]*big.Int{vk.Vka, vk.Vkc, vk.G2Kbg, vk.G2Kg, vk.Vkz})
	if err != nil {
		return
	}
	g1, err := G1Points([][3]*big.Int{vk.Vkb, vk.G1Kbg})
	if err != nil {
		return
	}
	ic, err := G1Points(vk.IC)
	if err != nil {
		return
	}
	pub, err := Scalars(publicSignals, order)
	if err != nil {
		return
	}
	if len(pub) == 0 {
		pub = make([]uint64, 4)
	}
	// proof = PiA, PiAp (12 words each), PiB (24), PiBp, PiC, PiCp, PiH, PiKp (12 each) = 108 words
	pa, err := G1Points([][3]*big.Int{p.PiA, p.PiAp})
	if err != nil {
		return
	}
	pb, err := G2Points([][3][2]*big.Int{p.PiB})
	if err != nil {
		return
	}
	pc, err := G1Points([][3]*big.Int{p.PiBp, p.PiC, p.PiCp, p.PiH, p.PiKp})
	if err != nil {
		return
	}
	proof := append(append(pa, pb...), pc...)
	var cok, cfail C.int
	err = call(func() C.int {
		return C.gs_pinocchio_verify(ptr(g2[0:]), ptr(g1[0:]), ptr(g2[24:]), ptr(g1[12:]), ptr(g2[48:]), ptr(g2[72:]), ptr(g2[96:]),
			ptr(ic), C.size_t(len(vk.IC)), ptr(pub), C.size_t(len(publicSignals)), ptr(proof), &cok, &cfail)
	})
	runtime.KeepAlive(g1)
	runtime.KeepAlive(g2)
	runtime.KeepAlive(ic)
	runtime.KeepAlive(pub)
	runtime.KeepAlive(proof)
	return cok == 1, int(cfail), err
}
