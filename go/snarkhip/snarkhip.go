// Package snarkhip is the drop-in for the three Pinocchio functions of the reference's root package
// (snark.go; callers cli/main.go:262-280, :337-356, wasm/go-snark-wasm-wrapper.go:21-110, snark_test.go):
//
//	GenerateTrustedSetup(witnessLength, circuit, alphas, betas, gammas) (snark.Setup, error)   snark.go:98
//	GenerateProofs(circuit, pk, w, px) (snark.Proof, error)                                    snark.go:254
//	VerifyProof(vk, proof, publicSignals, debug) bool                                          snark.go:292
//
// with the reference's exact signatures and types, on libgosnark_hip.so through go/gosnarkhip.  Reviewed-not-compiled
// in the build image; the C call sequences run on the GPU as tests/c/snark_*.c.
package snarkhip

import (
	"errors"
	"fmt"
	"math/big"
	"sync"

	snark "github.com/arnaucube/go-snark-study"
	"github.com/arnaucube/go-snark-study/circuitcompiler"

	"github.com/arnaucube/go-snark-study-hip/gosnarkhip"
)

// Device is the logical device the package's functions use; MaxResidentKeys bounds the keys kept in HBM.
var (
	Device          = 0
	MaxResidentKeys = 4
)

// identity of a key = identity of its arrays (pk arrives by value, its arrays are shared; see groth16hip)
type keyID struct {
	a, g1t *[3]*big.Int
	n      int
}
type entry struct {
	key  *gosnarkhip.PinocchioKey
	r1cs *gosnarkhip.R1CS // uploaded by the first GenerateProofsFromWitness
	used uint64
	refs int  // proofs using the entry right now (see groth16hip: eviction of a pinned entry is deferred to the last unpin)
	dead bool
}

var (
	mu    sync.Mutex
	keys  = map[keyID]*entry{}
	clock uint64
)

func idOf(pk *snark.Pk) (keyID, error) {
	if len(pk.A) == 0 || len(pk.G1T) == 0 {
		return keyID{}, errors.New("snarkhip: empty proving key")
	}
	return keyID{&pk.A[0], &pk.G1T[0], len(pk.A)}, nil
}

func drop(id keyID, e *entry) { // mu held
	delete(keys, id)
	if e.refs > 0 {
		e.dead = true
		return
	}
	_ = e.r1cs.Free()
	_ = e.key.Free()
}

func unpin(e *entry) {
	mu.Lock()
	defer mu.Unlock()
	e.refs--
	if e.dead && e.refs == 0 {
		_ = e.r1cs.Free()
		_ = e.key.Free()
	}
}

func remember(id keyID, k *gosnarkhip.PinocchioKey, pin bool) *entry { // mu held
	clock++
	e := &entry{key: k, used: clock}
	if pin {
		e.refs = 1
	}
	if prev, ok := keys[id]; ok { // the same backing arrays were uploaded again (a setup re-run in place): the old device key must not leak
		drop(id, prev)
	}
	keys[id] = e
	for len(keys) > MaxResidentKeys {
		var old keyID
		var oldest uint64 = ^uint64(0)
		for i, c := range keys {
			if c != e && c.used < oldest {
				old, oldest = i, c.used
			}
		}
		if oldest == ^uint64(0) {
			break
		}
		drop(old, keys[old])
	}
	return e
}

// deviceKey returns the cache entry of pk, PINNED: unpin(e) when the proof is done.
func deviceKey(circuit circuitcompiler.Circuit, pk *snark.Pk) (*entry, error) {
	id, err := idOf(pk)
	if err != nil {
		return nil, err
	}
	mu.Lock()
	defer mu.Unlock()
	if e, ok := keys[id]; ok {
		clock++
		e.used = clock
		e.refs++
		return e, nil
	}
	k, err := gosnarkhip.NewPinocchioKey(Device, gosnarkhip.PinocchioKeyParts{
		G1T: pk.G1T, A: pk.A, B: pk.B, C: pk.C, Kp: pk.Kp, Ap: pk.Ap, Bp: pk.Bp, Cp: pk.Cp, Z: pk.Z,
		NVars: circuit.NVars, NPublic: circuit.NPublic,
	}, snark.Utils.FqR.Q)
	if err != nil {
		return nil, err
	}
	return remember(id, k, true), nil
}

// ReleaseAll frees every cached key.
func ReleaseAll() {
	mu.Lock()
	defer mu.Unlock()
	for id, e := range keys {
		drop(id, e)
	}
}

func toProof(p gosnarkhip.PinocchioProof) snark.Proof {
	var proof snark.Proof
	proof.PiA, proof.PiAp, proof.PiB, proof.PiBp = p.PiA, p.PiAp, p.PiB, p.PiBp
	proof.PiC, proof.PiCp, proof.PiH, proof.PiKp = p.PiC, p.PiCp, p.PiH, p.PiKp
	return proof
}

// GenerateProofs has the reference's signature and semantics (snark.go:254-289): no randomness, eight proof elements,
// returned in the affine normal form.  Round 6: w and px travel as a host-buffer ticket collected at once (concurrent goroutines
// pipeline, nothing is allocated per proof); with all three slots taken the blocking entry point runs on the fourth.
// C call sequence: tests/c/snark_generateproofs.c (blocking form), tests/c/stream_host.c (ticket form).
func GenerateProofs(circuit circuitcompiler.Circuit, pk snark.Pk, w []*big.Int, px []*big.Int) (snark.Proof, error) {
	e, err := deviceKey(circuit, &pk)
	if err != nil {
		return snark.Proof{}, err
	}
	defer unpin(e)
	order := snark.Utils.FqR.Q
	var p gosnarkhip.PinocchioProof
	t, err := e.key.ProveHostBegin(w, px, order)
	if ge, ok := err.(*gosnarkhip.Error); ok && ge.Busy() {
		p, err = e.key.Prove(w, px, order)
	} else if err == nil {
		p, err = gosnarkhip.PinocchioProveEnd(t)
	}
	if err != nil {
		return snark.Proof{}, err
	}
	return toProof(p), nil
}

// GenerateProofsFromWitness is GenerateProofs for callers that have not computed px (the reference's callers run
// R1CSToQAP + CombinePolynomials on the CPU first, cli/main.go:330-349): circuit.R1CS is uploaded once per key and
// H(x) comes from the constraint values of the witness on the device.  Round 6: the witness goes over as a host-buffer ticket
// (no UploadScalars / Free per proof: that was a hipMalloc, a blocking copy and a hipFree each time).
// C call sequence: tests/c/witness_to_proof.c (resident form), tests/c/stream_host.c (host-buffer form used here).
func GenerateProofsFromWitness(circuit circuitcompiler.Circuit, pk snark.Pk, w []*big.Int) (snark.Proof, error) {
	order := snark.Utils.FqR.Q
	e, err := deviceKey(circuit, &pk)
	if err != nil {
		return snark.Proof{}, err
	}
	defer unpin(e)
	q, err := deviceR1CS(circuit, e)
	if err != nil {
		return snark.Proof{}, err
	}
	var p gosnarkhip.PinocchioProof
	t, err := e.key.ProveWitnessHostBegin(q, w, order)
	if ge, ok := err.(*gosnarkhip.Error); ok && ge.Busy() {
		p, err = e.key.ProveWitnessHost(q, w, order)
	} else if err == nil {
		p, err = gosnarkhip.PinocchioProveEnd(t)
	}
	if err != nil {
		return snark.Proof{}, err
	}
	return toProof(p), nil
}

// Prover is the streaming drop-in (see groth16hip.Prover): one key, many witnesses, three proofs in flight, proofs back in
// submission order.  C call sequence: tests/c/stream_producer.c (Groth16 twin) / tests/c/stream_host.c (the Pinocchio tickets).
type Prover struct {
	e *entry
	p *gosnarkhip.PinocchioProver
}

func NewProver(circuit circuitcompiler.Circuit, pk snark.Pk) (*Prover, error) {
	e, err := deviceKey(circuit, &pk)
	if err != nil {
		return nil, err
	}
	var q *gosnarkhip.R1CS
	if len(circuit.R1CS.A) != 0 {
		if q, err = deviceR1CS(circuit, e); err != nil {
			unpin(e)
			return nil, err
		}
	}
	return &Prover{e: e, p: gosnarkhip.NewPinocchioProver(e.key, q, snark.Utils.FqR.Q)}, nil
}

// Submit begins the proof of w (px == nil: H(x) from circuit.R1CS on the device); the slices are consumed when it returns.
func (p *Prover) Submit(w, px []*big.Int) error { return p.p.Submit(w, px) }
func (p *Prover) InFlight() int                  { return p.p.InFlight() }
func (p *Prover) Collect() (snark.Proof, error) {
	g, err := p.p.Collect()
	if err != nil {
		return snark.Proof{}, err
	}
	return toProof(g), nil
}
func (p *Prover) Close() {
	if p.e != nil {
		p.p.Close()
		unpin(p.e)
		p.e = nil
	}
}

func deviceR1CS(circuit circuitcompiler.Circuit, e *entry) (*gosnarkhip.R1CS, error) {
	mu.Lock() // e is pinned by the caller; the lock serialises the one-time upload
	defer mu.Unlock()
	var err error
	if e.r1cs != nil {
		return e.r1cs, nil
	}
	if len(circuit.R1CS.A) == 0 {
		return nil, errors.New("snarkhip: the circuit carries no R1CS (call circuit.GenerateR1CS first)")
	}
	order := snark.Utils.FqR.Q
	ca, nvars, err := gosnarkhip.CSRFromDense(circuit.R1CS.A, order)
	if err != nil {
		return nil, err
	}
	cb, _, err := gosnarkhip.CSRFromDense(circuit.R1CS.B, order)
	if err != nil {
		return nil, err
	}
	cc, _, err := gosnarkhip.CSRFromDense(circuit.R1CS.C, order)
	if err != nil {
		return nil, err
	}
	e.r1cs, err = gosnarkhip.UploadR1CS(Device, ca, cb, cc, nvars)
	return e.r1cs, err
}

// GenerateTrustedSetup has the reference's signature (snark.go:98-251): toxic values drawn as the reference draws them
// (:114-149, RhoC = RhoA * RhoB), key built on the device from circuit.R1CS (or from the polynomials' values when the
// circuit carries none), exported into snark.Setup and kept resident.  C call sequence: tests/c/snark_setup_prove_verify.c.
func GenerateTrustedSetup(witnessLength int, circuit circuitcompiler.Circuit, alphas, betas, gammas [][]*big.Int) (snark.Setup, error) {
	var setup snark.Setup
	var err error
	fq := snark.Utils.FqR
	for _, dst := range []**big.Int{&setup.Toxic.T, &setup.Toxic.Ka, &setup.Toxic.Kb, &setup.Toxic.Kc, &setup.Toxic.Kbeta, &setup.Toxic.Kgamma,
		&setup.Toxic.RhoA, &setup.Toxic.RhoB} {
		if *dst, err = fq.Rand(); err != nil {
			return snark.Setup{}, err
		}
	}
	setup.Toxic.RhoC = fq.Mul(setup.Toxic.RhoA, setup.Toxic.RhoB)
	order := fq.Q
	if len(alphas) != witnessLength || len(alphas) == 0 {
		return snark.Setup{}, fmt.Errorf("snarkhip: %d polynomials for a witness of length %d", len(alphas), witnessLength)
	}
	A, B, C := circuit.R1CS.A, circuit.R1CS.B, circuit.R1CS.C
	if len(A) == 0 {
		n := len(alphas[0])
		A, B, C = gosnarkhip.R1CSFromQAP(alphas, n, order), gosnarkhip.R1CSFromQAP(betas, n, order), gosnarkhip.R1CSFromQAP(gammas, n, order)
	}
	ca, nvars, err := gosnarkhip.CSRFromDense(A, order)
	if err != nil {
		return snark.Setup{}, err
	}
	cb, _, err := gosnarkhip.CSRFromDense(B, order)
	if err != nil {
		return snark.Setup{}, err
	}
	cc, _, err := gosnarkhip.CSRFromDense(C, order)
	if err != nil {
		return snark.Setup{}, err
	}
	k, vk, err := gosnarkhip.PinocchioSetup(Device, ca, cb, cc, nvars, circuit.NPublic, gosnarkhip.PinocchioToxic{
		T: setup.Toxic.T, Ka: setup.Toxic.Ka, Kb: setup.Toxic.Kb, Kc: setup.Toxic.Kc, Kbeta: setup.Toxic.Kbeta, Kgamma: setup.Toxic.Kgamma,
		RhoA: setup.Toxic.RhoA, RhoB: setup.Toxic.RhoB}, order)
	if err != nil {
		return snark.Setup{}, err
	}
	parts, err := k.Export(len(alphas) - 1)
	if err != nil {
		_ = k.Free()
		return snark.Setup{}, err
	}
	setup.Pk.G1T, setup.Pk.A, setup.Pk.B, setup.Pk.C, setup.Pk.Kp = parts.G1T, parts.A, parts.B, parts.C, parts.Kp
	setup.Pk.Ap, setup.Pk.Bp, setup.Pk.Cp, setup.Pk.Z = parts.Ap, parts.Bp, parts.Cp, parts.Z
	setup.Vk.Vka, setup.Vk.Vkb, setup.Vk.Vkc, setup.Vk.IC = vk.Vka, vk.Vkb, vk.Vkc, vk.IC
	setup.Vk.G1Kbg, setup.Vk.G2Kbg, setup.Vk.G2Kg, setup.Vk.Vkz = vk.G1Kbg, vk.G2Kbg, vk.G2Kg, vk.Vkz
	if id, err := idOf(&setup.Pk); err == nil {
		mu.Lock()
		remember(id, k, false)
		mu.Unlock()
	}
	return setup, nil
}

// VerifyProof has the reference's signature (snark.go:292-368); debug prints name the failing check like the
// reference's messages do.
func VerifyProof(vk snark.Vk, proof snark.Proof, publicSignals []*big.Int, debug bool) bool {
	ok, failed, err := gosnarkhip.PinocchioVerify(gosnarkhip.PinocchioVkParts{
		Vka: vk.Vka, Vkb: vk.Vkb, Vkc: vk.Vkc, IC: vk.IC, G1Kbg: vk.G1Kbg, G2Kbg: vk.G2Kbg, G2Kg: vk.G2Kg, Vkz: vk.Vkz,
	}, gosnarkhip.PinocchioProof{PiA: proof.PiA, PiAp: proof.PiAp, PiB: proof.PiB, PiBp: proof.PiBp, PiC: proof.PiC, PiCp: proof.PiCp,
		PiH: proof.PiH, PiKp: proof.PiKp}, publicSignals, snark.Utils.FqR.Q)
	if err != nil || !ok {
		if debug {
			names := []string{"", "e(piA, Va) == e(piA', g2)", "e(Vb, piB) == e(piB', g2)", "e(piC, Vc) == e(piC', g2)",
				"e(Vkx+piA, piB) == e(piH, Vkz) * e(piC, g2)", "e(Vkx+piA+piC, g2KbetaKgamma) * e(g1KbetaKgamma, piB) == e(piK, g2Kgamma)"}
			if err == nil && failed >= 1 && failed <= 5 {
				fmt.Println("❌", names[failed], "not passed")
			} else {
				fmt.Println("❌ verification not passed:", err)
			}
		}
		return false
	}
	if debug {
		fmt.Println("✓ verification passed")
	}
	return true
}
