package gosnarkhip

import (
	"errors"
	"math/big"
	"math/bits"
	"runtime"
	"sync"
)

// big.Word must be 64 bits wide: the packers copy big.Int.Bits() straight into the ABI's 64-bit limbs.
var _ = [1]struct{}{}[bits.UintSize-64]

// packChunk is the smallest slice of elements a goroutine of its own is started for (a goroutine costs ~1 us: 4096 elements of
// ~10 ns each amortise it).
const packChunk = 4096

// parallelRange runs f over [0, n) cut into at most GOMAXPROCS contiguous chunks and returns the first error.
func parallelRange(n int, f func(lo, hi int) error) error {
	workers := runtime.GOMAXPROCS(0)
	if most := (n + packChunk - 1) / packChunk; workers > most {
		workers = most
	}
	if workers <= 1 {
		return f(0, n)
	}
	errs := make([]error, workers)
	var wg sync.WaitGroup
	for k := 0; k < workers; k++ {
		lo, hi := n*k/workers, n*(k+1)/workers
		wg.Add(1)
		go func(k, lo, hi int) {
			defer wg.Done()
			errs[k] = f(lo, hi)
		}(k, lo, hi)
	}
	wg.Wait()
	for _, e := range errs {
		if e != nil {
			return e
		}
	}
	return nil
}

// scalarInto writes one field element.  Values below 2^256 cross the boundary as they are -- the device reduces every scalar it
// reads (k_digits canonicalises, the polynomial kernels convert through Montgomery form) -- so the common case is a copy of at
// most four words; only wider values (never produced by the reference's own arithmetic, fields/fq.go:32-98 reduces every result)
// pay a Mod.  Negative values are rejected as before.
func scalarInto(dst []uint64, v, r, tmp *big.Int) error {
	if v == nil || v.Sign() < 0 {
		return errors.New("gosnark-hip: nil or negative scalar")
	}
	w := v.Bits()
	if len(w) > 4 {
		tmp.Mod(v, r)
		w = tmp.Bits()
	}
	dst[0], dst[1], dst[2], dst[3] = 0, 0, 0, 0
	for i, x := range w {
		dst[i] = uint64(x)
	}
	return nil
}

// ScalarsInto packs vals into dst[:4*len(vals)] (n x 4 little-endian words), in parallel.  dst may come from LimbPool.
func ScalarsInto(dst []uint64, vals []*big.Int, r *big.Int) error {
	if len(dst) < 4*len(vals) {
		return errors.New("gosnark-hip: ScalarsInto: destination too short")
	}
	return parallelRange(len(vals), func(lo, hi int) error {
		tmp := new(big.Int)
		for i := lo; i < hi; i++ {
			if err := scalarInto(dst[4*i:4*i+4], vals[i], r, tmp); err != nil {
				return err
			}
		}
		return nil
	})
}

// Scalars packs field elements into a new n x 4 word buffer (see ScalarsInto; values < 2^256 are not reduced here, the device
// does that -- the reference's witness values are not canonical either, circuitcompiler/circuit.go:176-182).
func Scalars(vals []*big.Int, r *big.Int) ([]uint64, error) {
	out := make([]uint64, 4*len(vals))
	if err := ScalarsInto(out, vals, r); err != nil {
		return nil, err
	}
	return out, nil
}

// G1Points packs [][3]*big.Int Jacobian triples (bn128/g1.go:9-12) into n x 12 words, in parallel.
func G1Points(pts [][3]*big.Int) ([]uint64, error) {
	out := make([]uint64, 12*len(pts))
	err := parallelRange(len(pts), func(lo, hi int) error {
		for i := lo; i < hi; i++ {
			for k := 0; k < 3; k++ {
				if err := limbs(out[12*i+4*k:], pts[i][k]); err != nil {
					return err
				}
			}
		}
		return nil
	})
	if err != nil {
		return nil, err
	}
	return out, nil
}

// G2Points packs [][3][2]*big.Int (bn128/g2.go:9-12) into n x 24 words, in parallel.
func G2Points(pts [][3][2]*big.Int) ([]uint64, error) {
	out := make([]uint64, 24*len(pts))
	err := parallelRange(len(pts), func(lo, hi int) error {
		for i := lo; i < hi; i++ {
			for k := 0; k < 3; k++ {
				for j := 0; j < 2; j++ {
					if err := limbs(out[24*i+8*k+4*j:], pts[i][k][j]); err != nil {
						return err
					}
				}
			}
		}
		return nil
	})
	if err != nil {
		return nil, err
	}
	return out, nil
}

// LimbPool recycles the limb buffers of the per-proof packers (32 MiB for w and 64 MiB for px at 2^20 constraints): a stream of
// proofs allocates them once instead of once per proof.  Get returns a slice of exactly n words (contents undefined); Put gives
// it back.  The library has copied a buffer when the entry point it was passed to returns (cgo pointer rule), so a buffer may be
// Put right after ProveHostBegin / ProveWitnessHostBegin return.
var LimbPool limbPool

type limbPool struct{ p sync.Pool }

func (lp *limbPool) Get(n int) []uint64 {
	if v := lp.p.Get(); v != nil {
		if b := *(v.(*[]uint64)); cap(b) >= n {
			return b[:n]
		}
	}
	return make([]uint64, n)
}

func (lp *limbPool) Put(b []uint64) {
	if cap(b) > 0 {
		b = b[:cap(b)]
		lp.p.Put(&b)
	}
}
