package gosnarkhip

/*
#include "gosnark_hip.h"
*/
import "C"

import (
	"fmt"
	"math/big"
	"runtime"
	"unsafe"
)

// Round 5: the reference's two call shapes at speed.
//
// (1) groth16.GenerateProofs / snark.GenerateProofs receive a NEW witness (and px) in host memory with every call
// (groth16/groth16.go:225, snark.go:254; cli/main.go:480-501).  UploadScalars + ProveBegin + Free costs a hipMalloc, a blocking
// copy and a hipFree -- a device-wide synchronisation -- per proof.  The host-buffer tickets below stage the caller's arrays
// into buffers the ticket's slot owns (no allocation in steady state, AllocCounters shows it); UpdateScalars overwrites a
// resident vector in place.
//
// (2) cli/main.go:330-349 loads a key and proves ONCE.  SetTablePolicy(TablesAuto) -- the default -- sums a fresh key
// table-free instead of first spending ~140 ms and 5.6 GiB on window tables; BuildTables warms a key that will serve for hours.
//
// Call sequence of everything in this file = tests/c/stream_host.c; of prover.go (the streaming provers) = tests/c/stream_producer.c.

// UpdateScalars overwrites a resident scalar vector in place (len(vals) must equal its length).  The copy is ordered
// behind every read of the vector by tickets that are still outstanding and has landed when the call returns.
func UpdateScalars(h Handle, vals []*big.Int, order *big.Int) error {
	buf, err := Scalars(vals, order)
	if err != nil {
		return err
	}
	err = call(func() C.int { return C.gs_scalars_update(C.gs_handle(h), ptr(buf), C.size_t(len(vals))) })
	runtime.KeepAlive(buf)
	return err
}

// packRS packs (r, s) into the 8 words the prover entry points take.
func packRS(r, s, order *big.Int) (rs [8]uint64, err error) {
	err = ScalarsInto(rs[:], []*big.Int{r, s}, order)
	return
}

// ProveHostBegin is GenerateProofs' own argument list as a pipelined ticket: w and px from host memory, consumed
// when the call returns.  Collect with ProveEnd.  (Packing: parallel, no Mod, pooled buffers -- pack.go.)
func (k *Groth16Key) ProveHostBegin(w, px []*big.Int, r, s, order *big.Int) (Groth16Ticket, error) {
	rs, err := packRS(r, s, order)
	if err != nil {
		return 0, err
	}
	wb, pb := LimbPool.Get(4*len(w)), LimbPool.Get(4*len(px))
	defer LimbPool.Put(wb)
	defer LimbPool.Put(pb)
	if err = ScalarsInto(wb, w, order); err != nil {
		return 0, err
	}
	if err = ScalarsInto(pb, px, order); err != nil {
		return 0, err
	}
	return k.ProveHostBeginLimbs(wb, pb, &rs)
}

// ProveWitnessHostBegin: a fresh host witness against the resident sparse R1CS (no px at all).  Collect with ProveEnd.
func (k *Groth16Key) ProveWitnessHostBegin(q *R1CS, w []*big.Int, r, s, order *big.Int) (Groth16Ticket, error) {
	rs, err := packRS(r, s, order)
	if err != nil {
		return 0, err
	}
	wb := LimbPool.Get(4 * len(w))
	defer LimbPool.Put(wb)
	if err = ScalarsInto(wb, w, order); err != nil {
		return 0, err
	}
	return k.ProveWitnessHostBeginLimbs(q, wb, &rs)
}

// ProveWitnessHost is the blocking form: host witness -> proof.
func (k *Groth16Key) ProveWitnessHost(q *R1CS, w []*big.Int, r, s, order *big.Int) (piA [3]*big.Int, piB [3][2]*big.Int, piC [3]*big.Int, err error) {
	rs, err := packRS(r, s, order)
	if err != nil {
		return
	}
	wb := LimbPool.Get(4 * len(w))
	defer LimbPool.Put(wb)
	if err = ScalarsInto(wb, w, order); err != nil {
		return
	}
	var out [32]uint64
	var inf [3]C.int
	err = call(func() C.int {
		return C.gs_groth16_prove_witness_host(C.gs_handle(k.h), C.gs_handle(q.h), ptr(wb), C.size_t(len(w)), ptr(rs[0:]), ptr(rs[4:]),
			(*C.uint64_t)(unsafe.Pointer(&out[0])), &inf[0])
	})
	runtime.KeepAlive(wb)
	if err != nil {
		return
	}
	piA, piB, piC = groth16ProofFromWords(out[:], inf[:])
	return
}

// ProveHostBegin / ProveWitnessHostBegin / ProveWitnessHost of snark.GenerateProofs (collect the tickets with PinocchioProveEnd).
func (k *PinocchioKey) ProveHostBegin(w, px []*big.Int, order *big.Int) (uint64, error) {
	wb, pb := LimbPool.Get(4*len(w)), LimbPool.Get(4*len(px))
	defer LimbPool.Put(wb)
	defer LimbPool.Put(pb)
	if err := ScalarsInto(wb, w, order); err != nil {
		return 0, err
	}
	if err := ScalarsInto(pb, px, order); err != nil {
		return 0, err
	}
	return k.ProveHostBeginLimbs(wb, pb)
}

func (k *PinocchioKey) ProveWitnessHostBegin(q *R1CS, w []*big.Int, order *big.Int) (uint64, error) {
	wb := LimbPool.Get(4 * len(w))
	defer LimbPool.Put(wb)
	if err := ScalarsInto(wb, w, order); err != nil {
		return 0, err
	}
	return k.ProveWitnessHostBeginLimbs(q, wb)
}

func (k *PinocchioKey) ProveWitnessHost(q *R1CS, w []*big.Int, order *big.Int) (PinocchioProof, error) {
	wb := LimbPool.Get(4 * len(w))
	defer LimbPool.Put(wb)
	if err := ScalarsInto(wb, w, order); err != nil {
		return PinocchioProof{}, err
	}
	var out [72]uint64
	var inf [8]C.int
	err := call(func() C.int {
		return C.gs_pinocchio_prove_witness_host(C.gs_handle(k.h), C.gs_handle(q.h), ptr(wb), C.size_t(len(w)), (*C.uint64_t)(unsafe.Pointer(&out[0])), &inf[0])
	})
	runtime.KeepAlive(wb)
	if err != nil {
		return PinocchioProof{}, err
	}
	return pinocchioProofFromWords(out[:], inf[:]), nil
}

// TablePolicy says when a base array gets its window table (gs_set_table_policy).  Results never depend on it.
type TablePolicy int

const (
	TablesAuto   TablePolicy = 0 // table-free until an array's second use, then a build in instalments (default)
	TablesAlways TablePolicy = 1 // inside the first call that needs them (~140 ms per 2^20 Groth16 key)
	TablesNever  TablePolicy = 2 // table-free only: 0.4 GiB per 2^20 key instead of 6
)

// SetTablePolicy applies to every logical device.
func SetTablePolicy(p TablePolicy) error { return call(func() C.int { return C.gs_set_table_policy(C.int(p)) }) }

// BuildTables builds the window tables of a key or base array now (blocking), whatever the policy.
// route: 0 = everything the key can use, 1 = what the px routes need, 2 = what the witness routes need.
func BuildTables(h Handle, route int) error {
	return call(func() C.int { return C.gs_build_tables(C.gs_handle(h), C.int(route)) })
}
func (k *Groth16Key) BuildTables(route int) error   { return BuildTables(k.h, route) }
func (k *PinocchioKey) BuildTables(route int) error { return BuildTables(k.h, route) }

// SetMemoryLimit caps the device bytes the library may hold (0 = none): a development / test hook.  An allocation
// beyond the cap -- like a real out-of-memory -- evicts least-recently-used window tables and retries once.
func SetMemoryLimit(bytes uint64) error {
	return call(func() C.int { return C.gs_set_memory_limit(C.uint64_t(bytes)) })
}

// AllocCounters returns how often the library has called hipMalloc / hipFree so far.
func AllocCounters() (allocs, frees uint64) {
	var a, f C.uint64_t
	C.gs_alloc_counters(&a, &f)
	return uint64(a), uint64(f)
}

// CheckABI compares sizeof(gs_timing) / sizeof(gs_memory) of the loaded library with the header this package was
// compiled against (both structs are written through our pointers).  Init callers should run it once.
func CheckABI() error {
	var tb, mb C.size_t
	C.gs_abi_sizes(&tb, &mb)
	var t C.gs_timing
	var m C.gs_memory
	if uintptr(tb) != unsafe.Sizeof(t) || uintptr(mb) != unsafe.Sizeof(m) {
		return fmt.Errorf("gosnark-hip: library writes gs_timing/gs_memory of %d/%d bytes, this binding expects %d/%d: rebuild",
			uint64(tb), uint64(mb), unsafe.Sizeof(t), unsafe.Sizeof(m))
	}
	return nil
}
