// Package r1csqaphip puts the method set of the reference's r1csqap.PolynomialField (r1csqap/r1csqap.go:44-216) on
// libgosnark_hip.so: same names, same argument meaning, same results -- computed on the MI355X by the NTT engine instead of the
// reference's schoolbook loops (Mul O(n^2), Div O(n^3), LagrangeInterpolation O(n^3) per column).  Two deliberate differences:
// the reference's methods cannot fail and so return no error -- here a device error PANICS, as the reference itself does on bad
// input (snark.go:209) -- and NewPolZeroAt / LagrangeInterpolation are mathematically exact for every n, where the reference
// overflows a Go int from n = 22 on (r1csqap.go:130-136; identical results for n <= 21).  Reviewed-not-compiled in the build
// image (no Go toolchain); C call sequences: tests/c/polynomial_field.c.
package r1csqaphip

import (
	"math/big"

	"github.com/arnaucube/go-snark-study-hip/gosnarkhip"
	"github.com/arnaucube/go-snark-study/fields"
)

// Device is the logical device the polynomial entry points run on.
var Device = 0

// PolynomialField mirrors r1csqap.PolynomialField (r1csqap.go:44-47).
type PolynomialField struct {
	F fields.Fq
}

// NewPolynomialField mirrors r1csqap.NewPolynomialField (r1csqap.go:50-54).
func NewPolynomialField(f fields.Fq) PolynomialField { return PolynomialField{f} }

func must(err error) {
	if err != nil {
		panic(err)
	}
}

// Transpose mirrors r1csqap.Transpose (r1csqap.go:11-21); host work, kept for source compatibility.
func Transpose(matrix [][]*big.Int) [][]*big.Int {
	if len(matrix) == 0 {
		return nil
	}
	r := make([][]*big.Int, len(matrix[0]))
	for x := range r {
		r[x] = make([]*big.Int, len(matrix))
		for y := range matrix {
			r[x][y] = matrix[y][x]
		}
	}
	return r
}

// ArrayOfBigZeros mirrors r1csqap.ArrayOfBigZeros (r1csqap.go:24-30).
func ArrayOfBigZeros(num int) []*big.Int {
	r := make([]*big.Int, num)
	for i := range r {
		r[i] = big.NewInt(0)
	}
	return r
}

// Mul: r1csqap.go:57-67.
func (pf PolynomialField) Mul(a, b []*big.Int) []*big.Int {
	out, err := gosnarkhip.PolyMul(a, b, pf.F.Q)
	must(err)
	return out
}

// Div: r1csqap.go:70-84 (quotient, remainder).
func (pf PolynomialField) Div(a, b []*big.Int) ([]*big.Int, []*big.Int) {
	q, r, err := gosnarkhip.PolyDiv(a, b, pf.F.Q)
	must(err)
	return q, r
}

// Add: r1csqap.go:94-103.  Sub: :106-115.
func (pf PolynomialField) Add(a, b []*big.Int) []*big.Int {
	out, err := gosnarkhip.PolyAdd(a, b, pf.F.Q)
	must(err)
	return out
}
func (pf PolynomialField) Sub(a, b []*big.Int) []*big.Int {
	out, err := gosnarkhip.PolySub(a, b, pf.F.Q)
	must(err)
	return out
}

// Eval: r1csqap.go:118-126.
func (pf PolynomialField) Eval(v []*big.Int, x *big.Int) *big.Int {
	out, err := gosnarkhip.PolyEval(v, x, pf.F.Q)
	must(err)
	return out
}

// NewPolZeroAt: r1csqap.go:129-147 -- the polynomial of degree totalPoints - 1 that is `height` at node pointPos and 0 at the other
// nodes of 1..totalPoints.  That is the Lagrange interpolation of the vector height * e_pointPos.
func (pf PolynomialField) NewPolZeroAt(pointPos, totalPoints int, height *big.Int) []*big.Int {
	v := ArrayOfBigZeros(totalPoints)
	v[pointPos-1] = height
	return pf.LagrangeInterpolation(v)
}

// LagrangeInterpolation: r1csqap.go:150-158 (nodes 1..len(v)).
func (pf PolynomialField) LagrangeInterpolation(v []*big.Int) []*big.Int {
	out, err := gosnarkhip.LagrangeInterpolation(v, pf.F.Q)
	must(err)
	return out
}

// R1CSToQAP: r1csqap.go:161-188 -- per variable the interpolants of its column of a, b, c over the nodes 1..n, and
// Z = prod_{i=1}^{m-2} (x - i) (degree tied to the number of VARIABLES: the reference's shape contract, SURVEY fact 8).
func (pf PolynomialField) R1CSToQAP(a, b, c [][]*big.Int) ([][]*big.Int, [][]*big.Int, [][]*big.Int, []*big.Int) {
	cols := func(m [][]*big.Int) [][]*big.Int {
		t := Transpose(m)
		out := make([][]*big.Int, len(t))
		for i := range t {
			out[i] = pf.LagrangeInterpolation(t[i])
		}
		return out
	}
	alphas, betas, gammas := cols(a), cols(b), cols(c)
	z, err := gosnarkhip.ZPoly(len(alphas) - 2)
	must(err)
	return alphas, betas, gammas, z
}

// CombinePolynomials: r1csqap.go:191-210 -- ax = sum_i r_i ap_i, bx, cx likewise, px = ax * bx - cx.  The linear combinations are
// the interpolants of (A r), (B r), (C r), so the dense polynomials are turned back into their column values (Horner at the nodes
// 1..n on the host: the caller already paid O(m n) to hold them) and the sparse entry point does the rest on the device.
func (pf PolynomialField) CombinePolynomials(r []*big.Int, ap, bp, cp [][]*big.Int) ([]*big.Int, []*big.Int, []*big.Int, []*big.Int) {
	if len(ap) == 0 {
		return nil, nil, nil, nil
	}
	n := len(ap[0])
	order := pf.F.Q
	csr := func(p [][]*big.Int) gosnarkhip.CSR {
		m, _, err := gosnarkhip.CSRFromDense(gosnarkhip.R1CSFromQAP(p, n, order), order)
		must(err)
		return m
	}
	ax, bx, cx, px, err := gosnarkhip.R1CSToPx(csr(ap), csr(bp), csr(cp), len(ap), r, order)
	must(err)
	return ax, bx, cx, px
}

// DivisorPolynomial: r1csqap.go:213-216 (the quotient of px / z).
func (pf PolynomialField) DivisorPolynomial(px, z []*big.Int) []*big.Int {
	q, _ := pf.Div(px, z)
	return q
}
