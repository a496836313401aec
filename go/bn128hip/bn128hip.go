// Package bn128hip mirrors the group operations of the reference's bn128.G1 / bn128.G2 (bn128/g1.go:25-193, g2.go:25-223) on
// libgosnark_hip.so.  The loops the prover spends its time in -- MulScalar (double-and-add on math/big) and Add -- run on the
// MI355X as one- and two-term multi-scalar multiplications; results are the AFFINE representatives [x, y, 1] of the reference's
// Jacobian triples (parity is defined on the affine normal form: SURVEY facts 4-6), and Add is complete (the reference's formula
// returns Z = 0 for P + P, g1.go:32-89).  For more than a handful of terms use gosnarkhip.MSMG1 / MSMG2 directly: one call per
// SUM, not per term.  Device errors panic (the reference's methods return no error).  Reviewed-not-compiled in the build image;
// C call sequence: tests/c/group_ops.c.
package bn128hip

import (
	"math/big"

	"github.com/arnaucube/go-snark-study-hip/gosnarkhip"
)

// Device is the logical device used; Order is the group order r (bn128.go:46-50) scalars are reduced by.
var (
	Device   = 0
	Order, _ = new(big.Int).SetString("21888242871839275222246405745257275088548364400416034343698204186575808495617", 10)
	FieldQ, _ = new(big.Int).SetString("21888242871839275222246405745257275088696311157297823662689037894645226208583", 10)
)

func must(err error) {
	if err != nil {
		panic(err)
	}
}

// G1 mirrors bn128.G1's method set.
type G1 struct{}

func (G1) IsZero(p [3]*big.Int) bool { return p[2].Sign() == 0 } // g1.go:28-30

func (G1) MulScalar(p [3]*big.Int, e *big.Int) [3]*big.Int { // g1.go:140-155
	out, err := gosnarkhip.G1MulScalar(Device, p, e, Order)
	must(err)
	return out
}
func (G1) Add(p1, p2 [3]*big.Int) [3]*big.Int { // g1.go:32-89
	out, err := gosnarkhip.G1Add(Device, p1, p2)
	must(err)
	return out
}
func (G1) Neg(p [3]*big.Int) [3]*big.Int { // g1.go:91-96: (X, -Y, Z)
	return [3]*big.Int{p[0], new(big.Int).Mod(new(big.Int).Neg(p[1]), FieldQ), p[2]}
}
func (g G1) Sub(a, b [3]*big.Int) [3]*big.Int { return g.Add(a, g.Neg(b)) } // g1.go:98-100
func (g G1) Double(p [3]*big.Int) [3]*big.Int  { return g.Add(p, p) }        // g1.go:101-138 (the complete addition doubles)
func (g G1) Affine(p [3]*big.Int) [2]*big.Int { // g1.go:157-170
	if g.IsZero(p) {
		return [2]*big.Int{big.NewInt(0), big.NewInt(0)}
	}
	a := g.MulScalar(p, big.NewInt(1))
	return [2]*big.Int{a[0], a[1]}
}
func (g G1) Equal(p1, p2 [3]*big.Int) bool { // g1.go:172-193, on the affine normal form
	if g.IsZero(p1) || g.IsZero(p2) {
		return g.IsZero(p1) && g.IsZero(p2)
	}
	a, b := g.Affine(p1), g.Affine(p2)
	return a[0].Cmp(b[0]) == 0 && a[1].Cmp(b[1]) == 0
}

// G2 mirrors bn128.G2's method set.
type G2 struct{}

func (G2) IsZero(p [3][2]*big.Int) bool { return p[2][0].Sign() == 0 && p[2][1].Sign() == 0 } // g2.go:28-30

func (G2) MulScalar(p [3][2]*big.Int, e *big.Int) [3][2]*big.Int { // g2.go:142-181
	out, err := gosnarkhip.G2MulScalar(Device, p, e, Order)
	must(err)
	return out
}
func (G2) Add(p1, p2 [3][2]*big.Int) [3][2]*big.Int { // g2.go:32-89
	out, err := gosnarkhip.G2Add(Device, p1, p2)
	must(err)
	return out
}
func (G2) Neg(p [3][2]*big.Int) [3][2]*big.Int { // g2.go:91-97
	n := func(v *big.Int) *big.Int { return new(big.Int).Mod(new(big.Int).Neg(v), FieldQ) }
	return [3][2]*big.Int{p[0], {n(p[1][0]), n(p[1][1])}, p[2]}
}
func (g G2) Sub(a, b [3][2]*big.Int) [3][2]*big.Int { return g.Add(a, g.Neg(b)) } // g2.go:99-101
func (g G2) Double(p [3][2]*big.Int) [3][2]*big.Int  { return g.Add(p, p) }        // g2.go:103-140
func (g G2) Affine(p [3][2]*big.Int) [3][2]*big.Int { // g2.go:183-200: the reference returns the triple with Z = (1, 0)
	if g.IsZero(p) {
		return p
	}
	return g.MulScalar(p, big.NewInt(1))
}
func (g G2) Equal(p1, p2 [3][2]*big.Int) bool { // g2.go:202-223
	if g.IsZero(p1) || g.IsZero(p2) {
		return g.IsZero(p1) && g.IsZero(p2)
	}
	a, b := g.Affine(p1), g.Affine(p2)
	return a[0][0].Cmp(b[0][0]) == 0 && a[0][1].Cmp(b[0][1]) == 0 && a[1][0].Cmp(b[1][0]) == 0 && a[1][1].Cmp(b[1][1]) == 0
}
