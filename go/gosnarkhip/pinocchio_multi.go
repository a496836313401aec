package gosnarkhip

/*
#include "gosnark_hip.h"
*/
import "C"

import (
	"errors"
	"runtime"
	"unsafe"
)

// snark.GenerateProofs over several GPUs (SURVEY 8e applied to snark.go:254-289).  A Pinocchio proof is eight plain MSM sums, so
// rank k of N sums its term ranges and the ranks' eight partial points add up to the proof: key slices, partial sums (from px, or
// from the owner's slice of H's values), the addition, and the two deployment shapes (one process with N logical devices; one
// process per GPU over the communicator of CommInitRank).  C call sequences: tests/test_gpu_zy_multi.py (pinocchio tests).

// PinocchioPartials is one rank's record: the eight sums in the layout of a proof, with their infinity flags.
type PinocchioPartials struct {
	Sums [72]uint64
	Inf  [8]int32
}

// Shard cuts slice `index` of `count` out of a resident full key on the same logical device (gs_pinocchio_pk_shard).
func (k *PinocchioKey) Shard(index, count int) (*PinocchioKey, error) {
	var h C.gs_handle
	err := call(func() C.int { return C.gs_pinocchio_pk_shard(C.gs_handle(k.h), C.size_t(index), C.size_t(count), &h) })
	if err != nil {
		return nil, err
	}
	return &PinocchioKey{Handle(h), k.NVars, k.NPublic}, nil
}

// ShardTo creates the slice on logical device `target` (gs_pinocchio_pk_shard_to; the copies cross xGMI between physical GPUs).
func (k *PinocchioKey) ShardTo(index, count, target int) (*PinocchioKey, error) {
	var h C.gs_handle
	err := call(func() C.int {
		return C.gs_pinocchio_pk_shard_to(C.gs_handle(k.h), C.size_t(index), C.size_t(count), C.int(target), &h)
	})
	if err != nil {
		return nil, err
	}
	return &PinocchioKey{Handle(h), k.NVars, k.NPublic}, nil
}

func pinocchioPartialsCall(f func(out *C.uint64_t, inf *C.int) C.int) (PinocchioPartials, error) {
	var p PinocchioPartials
	var inf [8]C.int
	err := call(func() C.int { return f((*C.uint64_t)(unsafe.Pointer(&p.Sums[0])), &inf[0]) })
	for i := range inf {
		p.Inf[i] = int32(inf[i])
	}
	return p, err
}

// ProvePartials: the eight sums of shard `shard` of `count` from resident w and px (gs_pinocchio_prove_partials).
func (k *PinocchioKey) ProvePartials(w, px Handle, shard, count int) (PinocchioPartials, error) {
	return pinocchioPartialsCall(func(out *C.uint64_t, inf *C.int) C.int {
		return C.gs_pinocchio_prove_partials(C.gs_handle(k.h), C.gs_handle(w), C.gs_handle(px), C.size_t(shard), C.size_t(count), out, inf)
	})
}

// WitnessValues is the proof owner's polynomial stage (gs_pinocchio_witness_values), as Groth16Key.WitnessValues.
func (k *PinocchioKey) WitnessValues(q *R1CS, w, hv Handle) (Handle, uint32, error) {
	h := C.gs_handle(hv)
	var bad C.uint32_t
	err := call(func() C.int { return C.gs_pinocchio_witness_values(C.gs_handle(k.h), C.gs_handle(q.h), C.gs_handle(w), &h, &bad) })
	return Handle(h), uint32(bad), err
}

// ProvePartialsValues: the eight sums with PiH over this rank's slice of H's values (gs_pinocchio_prove_partials_values).
func (k *PinocchioKey) ProvePartialsValues(w, hvSlice Handle, shard, count int) (PinocchioPartials, error) {
	return pinocchioPartialsCall(func(out *C.uint64_t, inf *C.int) C.int {
		return C.gs_pinocchio_prove_partials_values(C.gs_handle(k.h), C.gs_handle(w), C.gs_handle(hvSlice), C.size_t(shard), C.size_t(count), out, inf)
	})
}

// PinocchioCombine adds the ranks' records up to the proof (gs_pinocchio_combine; host arithmetic).
func PinocchioCombine(records []PinocchioPartials) (PinocchioProof, error) {
	n := len(records)
	if n == 0 {
		return PinocchioProof{}, errors.New("gosnark-hip: PinocchioCombine needs at least one record")
	}
	sums := make([]uint64, 72*n)
	flags := make([]C.int, 8*n)
	for i, r := range records {
		copy(sums[72*i:], r.Sums[:])
		for j, f := range r.Inf {
			flags[8*i+j] = C.int(f)
		}
	}
	var out [72]uint64
	var inf [8]C.int
	err := call(func() C.int {
		return C.gs_pinocchio_combine(ptr(sums), &flags[0], C.size_t(n), (*C.uint64_t)(unsafe.Pointer(&out[0])), &inf[0])
	})
	runtime.KeepAlive(sums)
	runtime.KeepAlive(flags)
	if err != nil {
		return PinocchioProof{}, err
	}
	return pinocchioProofFromWords(out[:], inf[:]), nil
}

func pinocchioMulti(keys []*PinocchioKey, w, third []Handle, values bool) (PinocchioProof, bool, error) {
	n := len(keys)
	if n == 0 || len(w) != n || len(third) != n {
		return PinocchioProof{}, false, errors.New("gosnark-hip: one key, one w and one px (or slice of H's values) per device")
	}
	kh := make([]Handle, n)
	for d, k := range keys {
		kh[d] = k.h
	}
	var out [72]uint64
	var inf [8]C.int
	var used C.int
	err := call(func() C.int {
		o := (*C.uint64_t)(unsafe.Pointer(&out[0]))
		if values {
			return C.gs_pinocchio_prove_multi_values(handles(kh), handles(w), handles(third), C.int(n), o, &inf[0], &used)
		}
		return C.gs_pinocchio_prove_multi(handles(kh), handles(w), handles(third), C.int(n), o, &inf[0], &used)
	})
	runtime.KeepAlive(kh)
	runtime.KeepAlive(w)
	runtime.KeepAlive(third)
	if err != nil {
		return PinocchioProof{}, false, err
	}
	return pinocchioProofFromWords(out[:], inf[:]), used != 0, nil
}

// PinocchioProveMulti: ONE proof over len(keys) logical devices of this process (gs_pinocchio_prove_multi): keys[d] = slice d or a
// full replica, w[d] / px[d] replicas on the same device.
func PinocchioProveMulti(keys []*PinocchioKey, w, px []Handle) (PinocchioProof, bool, error) {
	return pinocchioMulti(keys, w, px, false)
}

// PinocchioProveMultiValues: the same on the values route; hvSlices[d] = device d's slice of H's values.
func PinocchioProveMultiValues(keys []*PinocchioKey, w, hvSlices []Handle) (PinocchioProof, bool, error) {
	return pinocchioMulti(keys, w, hvSlices, true)
}

func (k *PinocchioKey) sharded(w, third Handle, values bool) (PinocchioProof, error) {
	var out [72]uint64
	var inf [8]C.int
	err := call(func() C.int {
		o := (*C.uint64_t)(unsafe.Pointer(&out[0]))
		if values {
			return C.gs_pinocchio_prove_sharded_values(C.gs_handle(k.h), C.gs_handle(w), C.gs_handle(third), o, &inf[0])
		}
		return C.gs_pinocchio_prove_sharded(C.gs_handle(k.h), C.gs_handle(w), C.gs_handle(third), o, &inf[0])
	})
	if err != nil {
		return PinocchioProof{}, err
	}
	return pinocchioProofFromWords(out[:], inf[:]), nil
}

// ProveSharded: one process per GPU; this rank's shard, the 616-byte records gathered inside the library (gs_pinocchio_prove_sharded).
func (k *PinocchioKey) ProveSharded(w, px Handle) (PinocchioProof, error) { return k.sharded(w, px, false) }

// ProveShardedValues: the same with this rank's slice of H's values (ScatterScalars from the proof's owner).
func (k *PinocchioKey) ProveShardedValues(w, hvSlice Handle) (PinocchioProof, error) {
	return k.sharded(w, hvSlice, true)
}

// PinocchioProveBatch: independent proofs round-robined over the logical devices (gs_pinocchio_prove_batch): proof i runs where
// w[i] lives, with keyOfDevice[that device] (nil for unused devices); three in flight per device, no collective.
func PinocchioProveBatch(keyOfDevice []*PinocchioKey, w, px []Handle) ([]PinocchioProof, error) {
	n := len(w)
	if len(px) != n || len(keyOfDevice) == 0 {
		return nil, errors.New("gosnark-hip: PinocchioProveBatch needs one px per w and at least one key")
	}
	if n == 0 {
		return nil, nil
	}
	kh := make([]Handle, len(keyOfDevice))
	for d, k := range keyOfDevice {
		if k != nil {
			kh[d] = k.h
		}
	}
	out := make([]uint64, 72*n)
	inf := make([]C.int, 8*n)
	err := call(func() C.int {
		return C.gs_pinocchio_prove_batch(handles(kh), C.int(len(kh)), handles(w), handles(px), C.size_t(n), ptr(out), &inf[0])
	})
	runtime.KeepAlive(kh)
	runtime.KeepAlive(w)
	runtime.KeepAlive(px)
	if err != nil {
		return nil, err
	}
	proofs := make([]PinocchioProof, n)
	for i := range proofs {
		proofs[i] = pinocchioProofFromWords(out[72*i:72*i+72], inf[8*i:8*i+8])
	}
	return proofs, nil
}
