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

// Several MI355X from one Go process (SURVEY 8b / 8e): Init(0, 1, ..., 7) creates one context per GPU; the
// term ranges of a proof are cut into contiguous shards, one per device, and the library exchanges one 416-byte
// record per device (through ncclAllGather once CommInitLocal has been called).  Call sequence of
// ShardGroth16Key + ProveMulti = tests/c/multi_device.c; of ProveBatch = tests/c/batch_devices.c.

// CommInitLocal creates the in-process RCCL communicator (one rank per distinct physical device).
func CommInitLocal() error { return call(func() C.int { return C.gs_comm_init_local() }) }

// CommUniqueID / CommInitRank: one process per GPU.  Rank 0 draws the id, the application distributes its 128
// bytes (any channel), every process joins with its rank.
func CommUniqueID() ([128]byte, error) {
	var id [128]byte
	err := call(func() C.int { return C.gs_comm_unique_id((*C.uint8_t)(unsafe.Pointer(&id[0]))) })
	return id, err
}
func CommInitRank(id [128]byte, nranks, rank int) error {
	return call(func() C.int { return C.gs_comm_init_rank((*C.uint8_t)(unsafe.Pointer(&id[0])), C.int(nranks), C.int(rank)) })
}
func CommDestroy() { C.gs_comm_destroy() }

// Shard cuts slice `index` of `count` out of a resident full key and creates it on logical device `target`
// (each GPU then holds 1/count of every key array and builds window tables for its slice only).
func (k *Groth16Key) Shard(index, count, target int) (*Groth16Key, error) {
	var h C.gs_handle
	err := call(func() C.int { return C.gs_groth16_pk_shard_to(C.gs_handle(k.h), C.size_t(index), C.size_t(count), C.int(target), &h) })
	if err != nil {
		return nil, err
	}
	return &Groth16Key{Handle(h), k.NVars, k.NPublic}, nil
}

func handles(hs []Handle) *C.gs_handle { return (*C.gs_handle)(unsafe.Pointer(&hs[0])) }

// ProveMulti is ONE Groth16 proof over len(keys) logical devices: keys[d] = slice d (or a full replica), w[d] /
// px[d] = the witness and P(x) resident on the same device.  usedRCCL tells whether the records went through
// ncclAllGather.  Same proof as keys[0].Prove on a full key.
func ProveMulti(keys []*Groth16Key, w, px []Handle, r, s, order *big.Int) (piA [3]*big.Int, piB [3][2]*big.Int, piC [3]*big.Int, usedRCCL bool, err error) {
	n := len(keys)
	if n == 0 || len(w) != n || len(px) != n {
		err = errors.New("gosnark-hip: ProveMulti needs one key, w and px per device")
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
		return C.gs_groth16_prove_multi(handles(kh), handles(w), handles(px), C.int(n), ptr(rs[0:]), ptr(rs[4:]),
			(*C.uint64_t)(unsafe.Pointer(&out[0])), &inf[0], &used)
	})
	runtime.KeepAlive(kh)
	runtime.KeepAlive(w)
	runtime.KeepAlive(px)
	runtime.KeepAlive(rs)
	if err != nil {
		return
	}
	piA, piB, piC = groth16ProofFromWords(out[:], inf[:])
	usedRCCL = used != 0
	return
}

// BatchProof is one result of ProveBatch.
type BatchProof struct {
	PiA [3]*big.Int
	PiB [3][2]*big.Int
	PiC [3]*big.Int
}

// ProveBatch runs independent proofs round-robin over the devices (BASELINE configs[4], no collective): proof i
// reads w[i] / px[i] and runs on the logical device those handles live on, with keyOfDevice[that device] (nil for
// unused devices); three proofs in flight per device, all devices concurrently.
func ProveBatch(keyOfDevice []*Groth16Key, w, px []Handle, r, s []*big.Int, order *big.Int) ([]BatchProof, error) {
	n := len(w)
	if n == 0 {
		return nil, nil
	}
	if len(px) != n || len(r) != n || len(s) != n || len(keyOfDevice) == 0 {
		return nil, errors.New("gosnark-hip: ProveBatch needs w, px, r, s per proof and one key per device")
	}
	kh := make([]Handle, len(keyOfDevice))
	for d, k := range keyOfDevice {
		if k != nil {
			kh[d] = k.h
		}
	}
	rb, err := Scalars(r, order)
	if err != nil {
		return nil, err
	}
	sb, err := Scalars(s, order)
	if err != nil {
		return nil, err
	}
	out := make([]uint64, 32*n)
	inf := make([]C.int, 3*n)
	err = call(func() C.int {
		return C.gs_groth16_prove_batch(handles(kh), C.int(len(kh)), handles(w), handles(px), C.size_t(n), ptr(rb), ptr(sb), ptr(out), &inf[0])
	})
	runtime.KeepAlive(kh)
	runtime.KeepAlive(w)
	runtime.KeepAlive(px)
	runtime.KeepAlive(rb)
	runtime.KeepAlive(sb)
	if err != nil {
		return nil, err
	}
	res := make([]BatchProof, n)
	for i := range res {
		res[i].PiA, res[i].PiB, res[i].PiC = groth16ProofFromWords(out[32*i:32*i+32], inf[3*i:3*i+3])
	}
	return res, nil
}

// MSMG1Multi is one MSM over len(bases) logical devices: bases[d] / scalars[d] = shard d of the term range
// (CloneG1 / CloneScalars), BASELINE configs[3].
func MSMG1Multi(bases, scalars []Handle) (p [3]*big.Int, usedRCCL bool, err error) {
	if len(bases) == 0 || len(bases) != len(scalars) {
		err = errors.New("gosnark-hip: MSMG1Multi needs one base and one scalar shard per device")
		return
	}
	var out [8]uint64
	var inf, used C.int
	err = call(func() C.int {
		return C.gs_msm_g1_multi(handles(bases), handles(scalars), C.int(len(bases)), (*C.uint64_t)(unsafe.Pointer(&out[0])), &inf, &used)
	})
	runtime.KeepAlive(bases)
	runtime.KeepAlive(scalars)
	return G1FromAffine(out[:], inf != 0), used != 0, err
}
