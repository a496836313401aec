package gosnarkhip

import (
	"errors"
	"math/big"
)

// CSR is one R1CS matrix (constraints x variables) in compressed sparse rows: what gs_groth16_setup,
// gs_pinocchio_setup and gs_r1cs_upload take instead of the reference's dense [][]*big.Int
// (circuitcompiler/circuit.go:22-26), which is unusable past ~2^10 constraints.
type CSR struct {
	RowPtr []uint32 // n + 1
	Col    []uint32 // nnz
	Val    []uint64 // nnz x 4 words, reduced mod r
}

// CSRFromDense packs circuit.R1CS.A / .B / .C ([constraint][variable]) and returns the variable count.
func CSRFromDense(m [][]*big.Int, order *big.Int) (CSR, int, error) {
	var c CSR
	if len(m) == 0 {
		return c, 0, errors.New("gosnark-hip: empty R1CS matrix")
	}
	nvars := len(m[0])
	c.RowPtr = make([]uint32, len(m)+1)
	t := new(big.Int)
	for j, row := range m {
		if len(row) != nvars {
			return c, 0, errors.New("gosnark-hip: ragged R1CS matrix")
		}
		for k, v := range row {
			if v == nil || v.Sign() == 0 {
				continue
			}
			t.Mod(v, order) // the compiler emits negative coefficients (circuit.go:108-118); Mod is Euclidean
			if t.Sign() == 0 {
				continue
			}
			c.Col = append(c.Col, uint32(k))
			var w [4]uint64
			if err := limbs(w[:], t); err != nil {
				return c, 0, err
			}
			c.Val = append(c.Val, w[:]...)
		}
		c.RowPtr[j+1] = uint32(len(c.Col))
	}
	if len(c.Col) == 0 { // keep the pointers non-nil for cgo
		c.Col = []uint32{0}[:0:1]
		c.Val = make([]uint64, 0, 4)
	}
	return c, nvars, nil
}

// R1CSFromQAP recovers the dense R1CS column values from the reference's dense QAP (alphas[i] = coefficients
// of variable i's polynomial, r1csqap.go:161-188): A[j][i] = alphas[i](j+1), nodes 1..n.  Only used when a
// caller hands GenerateTrustedSetup polynomials without circuit.R1CS; O(m n^2) on the host, which is nothing
// next to the dense QAP the caller already built.
func R1CSFromQAP(polys [][]*big.Int, n int, order *big.Int) [][]*big.Int {
	out := make([][]*big.Int, n)
	for j := range out {
		out[j] = make([]*big.Int, len(polys))
		x := big.NewInt(int64(j + 1))
		for i, p := range polys {
			acc := new(big.Int)
			for k := len(p) - 1; k >= 0; k-- { // Horner
				acc.Mul(acc, x)
				acc.Add(acc, p[k])
				acc.Mod(acc, order)
			}
			out[j][i] = acc
		}
	}
	return out
}
