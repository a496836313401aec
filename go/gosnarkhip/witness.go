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

// Witness -> proof on the device.  The reference's callers (cli/main.go:330-349, 480-501) run R1CSToQAP +
// CombinePolynomials on the CPU to obtain px and hand it to GenerateProofs; here the circuit's sparse R1CS is
// uploaded once (UploadR1CS) and every proof needs only the witness: CombinePolynomials and Div
// (r1csqap.go:191-216) collapse into H(x) computed from the constraint values (gs_*_prove_witness).
// Call sequence of UploadR1CS + Px + ProveWitness (both protocols) = tests/c/witness_to_proof.c.

// R1CS is a sparse constraint system resident on one logical device.
type R1CS struct {
	h           Handle
	Constraints int
	NVars       int
}

// Handle exposes the resident object.
func (q *R1CS) Handle() Handle { return q.h }

// Free releases it (deferred while proofs that read it are in flight).
func (q *R1CS) Free() error {
	if q == nil || q.h == 0 {
		return nil
	}
	err := Free(q.h)
	if err == nil {
		q.h = 0
	}
	return err
}

// UploadR1CS validates A, B, C (constraints x variables, CSR) and keeps them resident on logical device `device`.
func UploadR1CS(device int, a, b, c CSR, nvars int) (*R1CS, error) {
	n := len(a.RowPtr) - 1
	if n < 1 || len(b.RowPtr) != n+1 || len(c.RowPtr) != n+1 {
		return nil, errors.New("gosnark-hip: A, B, C must have the same number of constraints")
	}
	var h C.gs_handle
	err := onDevice(device, func() C.int {
		return C.gs_r1cs_upload(C.size_t(n), C.size_t(nvars),
			ptr32(a.RowPtr), ptr32(a.Col), ptr(a.Val), ptr32(b.RowPtr), ptr32(b.Col), ptr(b.Val), ptr32(c.RowPtr), ptr32(c.Col), ptr(c.Val), &h)
	})
	runtime.KeepAlive(a)
	runtime.KeepAlive(b)
	runtime.KeepAlive(c)
	if err != nil {
		return nil, err
	}
	return &R1CS{Handle(h), n, nvars}, nil
}

// Px is CombinePolynomials' px = ax * bx - cx (r1csqap.go:191-210) from a resident witness, left resident:
// pass px = 0 to create the vector or an earlier result to overwrite it.  For callers that need px itself;
// ProveWitness does not.
func (q *R1CS) Px(w Handle, px Handle) (Handle, error) {
	h := C.gs_handle(px)
	err := call(func() C.int { return C.gs_r1cs_px(C.gs_handle(q.h), C.gs_handle(w), &h) })
	return Handle(h), err
}

// DownloadScalars reads a resident scalar vector back (n elements).
func DownloadScalars(h Handle, n int) ([]*big.Int, error) {
	buf := make([]uint64, 4*n)
	if err := call(func() C.int { return C.gs_scalars_download(C.gs_handle(h), ptr(buf), C.size_t(n)) }); err != nil {
		return nil, err
	}
	return unpackScalars(buf), nil
}

// ProveWitness is groth16.GenerateProofs without px: witness (resident, NVars elements) -> proof.
func (k *Groth16Key) ProveWitness(q *R1CS, w Handle, r, s, order *big.Int) (piA [3]*big.Int, piB [3][2]*big.Int, piC [3]*big.Int, err error) {
	rs, err := Scalars([]*big.Int{r, s}, order)
	if err != nil {
		return
	}
	var out [32]uint64
	var inf [3]C.int
	err = call(func() C.int {
		return C.gs_groth16_prove_witness(C.gs_handle(k.h), C.gs_handle(q.h), C.gs_handle(w), ptr(rs[0:]), ptr(rs[4:]),
			(*C.uint64_t)(unsafe.Pointer(&out[0])), &inf[0])
	})
	runtime.KeepAlive(rs)
	if err != nil {
		return
	}
	piA, piB, piC = groth16ProofFromWords(out[:], inf[:])
	return
}

// ProveWitness is snark.GenerateProofs without px.
func (k *PinocchioKey) ProveWitness(q *R1CS, w Handle) (PinocchioProof, error) {
	var out [72]uint64
	var inf [8]C.int
	err := call(func() C.int {
		return C.gs_pinocchio_prove_witness(C.gs_handle(k.h), C.gs_handle(q.h), C.gs_handle(w), (*C.uint64_t)(unsafe.Pointer(&out[0])), &inf[0])
	})
	if err != nil {
		return PinocchioProof{}, err
	}
	return pinocchioProofFromWords(out[:], inf[:]), nil
}
