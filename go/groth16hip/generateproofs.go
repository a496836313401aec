// Package groth16hip shows the reference-side change: groth16.GenerateProofs keeps its signature
// (groth16/groth16.go:225) and delegates to the resident key.  Drop this next to groth16.go (package
// groth16) or import it from cli/main.go:501 / wasm wrapper call sites.  Reviewed-not-compiled in the
// build image (no Go toolchain there).
package groth16hip

import (
	"math/big"
	"sync"

	"github.com/arnaucube/go-snark-study/circuitcompiler"
	"github.com/arnaucube/go-snark-study/groth16"

	"gosnarkhip"
)

var (
	mu   sync.Mutex
	keys = map[*groth16.Pk]*gosnarkhip.Groth16Key{} // one upload per proving key, not per proof
)

func deviceKey(circuit circuitcompiler.Circuit, pk *groth16.Pk) (*gosnarkhip.Groth16Key, error) {
	mu.Lock()
	defer mu.Unlock()
	if k, ok := keys[pk]; ok {
		return k, nil
	}
	k, err := gosnarkhip.NewGroth16Key(gosnarkhip.Groth16KeyParts{
		At: pk.G1.At, BACGamma1: pk.G1.BACGamma, BACDelta: pk.BACDelta, PowersTauDelta: pk.PowersTauDelta,
		BACGamma2: pk.G2.BACGamma,
		Alpha:     pk.G1.Alpha, Beta: pk.G1.Beta, Delta: pk.G1.Delta,
		Beta2: pk.G2.Beta, Delta2: pk.G2.Delta,
		Z: pk.Z, NVars: circuit.NVars, NPublic: circuit.NPublic,
	}, groth16.Utils.FqR.Q)
	if err == nil {
		keys[pk] = k
	}
	return k, err
}

// GenerateProofs has the reference's signature and semantics (groth16/groth16.go:225-278); the proof
// elements come back as the affine representatives [x, y, 1] of the reference's Jacobian triples.
func GenerateProofs(circuit circuitcompiler.Circuit, pk groth16.Pk, w []*big.Int, px []*big.Int) (groth16.Proof, error) {
	var proof groth16.Proof
	r, err := groth16.Utils.FqR.Rand() // groth16.go:231-234
	if err != nil {
		return proof, err
	}
	s, err := groth16.Utils.FqR.Rand() // :235-238
	if err != nil {
		return proof, err
	}
	return GenerateProofsWithRS(circuit, &pk, w, px, r, s)
}

// GenerateProofsWithRS injects the randomness (needed for parity tests against a recorded proof).
func GenerateProofsWithRS(circuit circuitcompiler.Circuit, pk *groth16.Pk, w, px []*big.Int, r, s *big.Int) (groth16.Proof, error) {
	var proof groth16.Proof
	k, err := deviceKey(circuit, pk)
	if err != nil {
		return proof, err // callers may fall back to groth16.GenerateProofs (the CPU reference)
	}
	proof.PiA, proof.PiB, proof.PiC, err = k.Prove(w, px, r, s, groth16.Utils.FqR.Q)
	return proof, err
}
