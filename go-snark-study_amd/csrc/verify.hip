// Verifier entry points: groth16.VerifyProof (groth16/groth16.go:281-305), snark.VerifyProof
// (snark.go:292-368) and the pairing seam under them (bn128/bn128.go:179-186), on the host.  They do not touch the
// device context and do not take its lock: a verifier thread can run beside in-flight proofs.
#include <atomic>

#include "pairing.h"
#include "runtime.h"

using namespace gs;
using namespace gs::pairing;

namespace {

struct ShapeError { const char* msg; };

// gs_verify_set_strict: 0 (default) = the reference's big.Int behaviour, 1 = canonical encodings only
std::atomic<int> g_strict{0};
const uint64_t kRorder[4] = {0x43e1f593f0000001ULL, 0x2833e84879b97091ULL, 0xb85045b68181585dULL, 0x30644e72e131a029ULL};   // bn128.go:46 (R)

// every coordinate < q ?  (words: count of 4-limb field elements)
bool coords_canonical(const uint64_t* w, size_t count) {
  for (size_t i = 0; i < count; ++i)
    if (limbs_geq(w + 4 * i, kP)) return false;
  return true;
}
bool scalars_canonical(const uint64_t* w, size_t count) {
  for (size_t i = 0; i < count; ++i)
    if (limbs_geq(w + 4 * i, kRorder)) return false;
  return true;
}

// prod e(P_i, Q_i) == 1 with one shared final exponentiation.  Off-curve inputs and G2 inputs that hit a vertical
// line inside the loop (not of order r) make the check fail instead of producing a meaningless value.
// `untrusted_g2` lists the indices of G2 inputs that came from a prover (they get the subgroup check; key material is
// checked once by whoever installs the key, not per proof).
bool product_is_one(const std::vector<G1Aff>& ps, const std::vector<G2Aff>& qs, const std::vector<int>& untrusted_g2 = {}) {
  for (const G1Aff& p : ps)
    if (!g1_on_curve(p)) return false;
  for (const G2Aff& q : qs)
    if (!g2_on_curve(q)) return false;
  for (int i : untrusted_g2)
    if (!g2_in_subgroup(qs[(size_t)i])) return false;
  Fp12 f;
  if (!multi_miller_loop(ps, qs, f)) return false;
  return f12_eq(final_exponentiation(f), f12_one());
}

void f12_to_std(const Fp12& f, uint64_t* out) {     // the reference's [2][3][2]*big.Int order
  const Fp2* c[6] = {&f.a0.c0, &f.a0.c1, &f.a0.c2, &f.a1.c0, &f.a1.c1, &f.a1.c2};
  for (int i = 0; i < 6; ++i) {
    fp_to_std(c[i]->c0, out + 8 * i);
    fp_to_std(c[i]->c1, out + 8 * i + 4);
  }
}

// vk.IC[0] + sum_i publicSignals[i] * vk.IC[i+1]   (groth16.go:283-286, snark.go:330-333)
G1Jac accumulate_ic(const uint64_t* ic, size_t nic, const uint64_t* pub, size_t npublic) {
  G1Jac acc = g1j_from(g1_from_jacobian_std(ic));
  for (size_t i = 0; i < npublic; ++i)
    acc = g1j_add(acc, g1j_mul(g1j_from(g1_from_jacobian_std(ic + 12 * (i + 1))), pub + 4 * i));
  return acc;
}

template <class F>
int host_guarded(F&& f) {
  try {
    return f();
  } catch (const std::bad_alloc&) {
    return fail(GS_ERR_HIP, "host allocation failed");
  } catch (const std::exception& e) {
    return fail(GS_ERR_ARG, "%s", e.what());
  }
}

const uint64_t kG2Gen[24] = {      // bn128.go:56-83 (Gg2), Z = (1, 0)
    0x46debd5cd992f6edULL, 0x674322d4f75edaddULL, 0x426a00665e5c4479ULL, 0x1800deef121f1e76ULL,
    0x97e485b7aef312c2ULL, 0xf1aa493335a9e712ULL, 0x7260bfb731fb5d25ULL, 0x198e9393920d483aULL,
    0x4ce6cc0166fa7daaULL, 0xe3d1e7690c43d37bULL, 0x4aab71808dcb408fULL, 0x12c85ea5db8c6debULL,
    0x55acdadcd122975bULL, 0xbc4b313370b38ef3ULL, 0xec9e99ad690c3395ULL, 0x090689d0585ff075ULL,
    1, 0, 0, 0, 0, 0, 0, 0};

}  // namespace

extern "C" {

int gs_pairing(const uint64_t g1[12], const uint64_t g2[24], uint64_t out_fq12[48]) {
  return host_guarded([&]() -> int {
    if (!g1 || !g2 || !out_fq12) return fail(GS_ERR_ARG, "gs_pairing: null argument");
    std::vector<G1Aff> ps{g1_from_jacobian_std(g1)};
    std::vector<G2Aff> qs{g2_from_jacobian_std(g2)};
    if (!g1_on_curve(ps[0]) || !g2_on_curve(qs[0])) return fail(GS_ERR_ARG, "gs_pairing: point not on the curve");
    Fp12 f;
    if (!multi_miller_loop(ps, qs, f)) return fail(GS_ERR_ARG, "gs_pairing: G2 point is not of order r");
    f12_to_std(final_exponentiation(f), out_fq12);
    return GS_OK;
  });
}

int gs_pairing_check(const uint64_t* g1, const uint64_t* g2, size_t k, int* ok) {
  return host_guarded([&]() -> int {
    if (!ok || (k && (!g1 || !g2))) return fail(GS_ERR_ARG, "gs_pairing_check: null argument");
    std::vector<G1Aff> ps(k);
    std::vector<G2Aff> qs(k);
    for (size_t i = 0; i < k; ++i) {
      ps[i] = g1_from_jacobian_std(g1 + 12 * i);
      qs[i] = g2_from_jacobian_std(g2 + 24 * i);
    }
    std::vector<int> all(k);
    for (size_t i = 0; i < k; ++i) all[i] = (int)i;
    *ok = product_is_one(ps, qs, all) ? 1 : 0;
    return GS_OK;
  });
}

int gs_verify_set_strict(int on) {
  g_strict.store(on ? 1 : 0);
  return GS_OK;
}

int gs_groth16_verify(const uint64_t vk_g1_alpha[12], const uint64_t vk_g2_beta[24], const uint64_t vk_g2_gamma[24],
                      const uint64_t vk_g2_delta[24], const uint64_t* vk_ic, size_t nic, const uint64_t* public_signals,
                      size_t npublic, const uint64_t pi_a[12], const uint64_t pi_b[24], const uint64_t pi_c[12], int* ok) {
  return host_guarded([&]() -> int {
    if (!vk_g1_alpha || !vk_g2_beta || !vk_g2_gamma || !vk_g2_delta || !vk_ic || !pi_a || !pi_b || !pi_c || !ok || (npublic && !public_signals))
      return fail(GS_ERR_ARG, "gs_groth16_verify: null argument");
    // the reference indexes vk.IC[i+1] for every public signal and panics past the end (groth16.go:285)
    if (nic < npublic + 1) return fail(GS_ERR_SHAPE, "gs_groth16_verify: %zu public signals need %zu IC points, vk has %zu", npublic, npublic + 1, nic);
    if (g_strict.load()) {
      // strict: exactly one signal per IC point (a short list would silently verify the statement "the missing inputs are 0"),
      // and canonical encodings only (x and x + r, X and X + q name the same element: accepting both makes proofs malleable)
      if (nic != npublic + 1) return fail(GS_ERR_SHAPE, "gs_groth16_verify (strict): vk has %zu IC points, so exactly %zu public signals are required (got %zu)", nic, nic - 1, npublic);
      if (!scalars_canonical(public_signals, npublic) || !coords_canonical(pi_a, 3) || !coords_canonical(pi_b, 6) || !coords_canonical(pi_c, 3)) {
        *ok = 0;
        return GS_OK;
      }
    }
    G1Aff ic = g1j_affine(accumulate_ic(vk_ic, nic, public_signals, npublic));
    // e(A, B) == e(alpha, beta) e(IC, gamma) e(C, delta)   <=>   e(-A, B) e(alpha, beta) e(IC, gamma) e(C, delta) == 1
    std::vector<G1Aff> ps{g1_neg(g1_from_jacobian_std(pi_a)), g1_from_jacobian_std(vk_g1_alpha), ic, g1_from_jacobian_std(pi_c)};
    std::vector<G2Aff> qs{g2_from_jacobian_std(pi_b), g2_from_jacobian_std(vk_g2_beta), g2_from_jacobian_std(vk_g2_gamma),
                          g2_from_jacobian_std(vk_g2_delta)};
    *ok = product_is_one(ps, qs, {0}) ? 1 : 0;                  // PiB is the prover's G2 element
    return GS_OK;
  });
}

int gs_pinocchio_verify(const uint64_t vka[24], const uint64_t vkb[12], const uint64_t vkc[24], const uint64_t g1kbg[12],
                        const uint64_t g2kbg[24], const uint64_t g2kg[24], const uint64_t vkz[24], const uint64_t* vk_ic, size_t nic,
                        const uint64_t* public_signals, size_t npublic, const uint64_t* proof, int* ok, int* failed_check) {
  return host_guarded([&]() -> int {
    if (!vka || !vkb || !vkc || !g1kbg || !g2kbg || !g2kg || !vkz || !vk_ic || !proof || !ok || (npublic && !public_signals))
      return fail(GS_ERR_ARG, "gs_pinocchio_verify: null argument");
    if (nic < npublic + 1) return fail(GS_ERR_SHAPE, "gs_pinocchio_verify: %zu public signals need %zu IC points, vk has %zu", npublic, npublic + 1, nic);
    if (g_strict.load()) {
      if (nic != npublic + 1) return fail(GS_ERR_SHAPE, "gs_pinocchio_verify (strict): vk has %zu IC points, so exactly %zu public signals are required (got %zu)", nic, nic - 1, npublic);
      if (!scalars_canonical(public_signals, npublic) || !coords_canonical(proof, 27)) {
        *ok = 0;
        if (failed_check) *failed_check = 0;
        return GS_OK;
      }
    }
    // proof layout: PiA, PiAp (G1) | PiB (G2) | PiBp, PiC, PiCp, PiH, PiKp (G1)  -- snark.go:59-69
    const G1Aff piA = g1_from_jacobian_std(proof), piAp = g1_from_jacobian_std(proof + 12);
    const G2Aff piB = g2_from_jacobian_std(proof + 24);
    const G1Aff piBp = g1_from_jacobian_std(proof + 48), piC = g1_from_jacobian_std(proof + 60), piCp = g1_from_jacobian_std(proof + 72);
    const G1Aff piH = g1_from_jacobian_std(proof + 84), piKp = g1_from_jacobian_std(proof + 96);
    const G2Aff g2 = g2_from_jacobian_std(kG2Gen);
    const G2Aff Vka = g2_from_jacobian_std(vka), Vkc = g2_from_jacobian_std(vkc), G2Kbg = g2_from_jacobian_std(g2kbg);
    const G2Aff G2Kg = g2_from_jacobian_std(g2kg), Vkz = g2_from_jacobian_std(vkz);
    const G1Aff Vkb = g1_from_jacobian_std(vkb), G1Kbg = g1_from_jacobian_std(g1kbg);
    int bad = 0;
    auto check = [&](int which, std::vector<G1Aff> ps, std::vector<G2Aff> qs) {
      if (!bad && !product_is_one(ps, qs)) bad = which;
    };
    check(1, {piA, g1_neg(piAp)}, {Vka, g2});                       // e(piA, Va) == e(piA', g2)            snark.go:294-304
    if (!bad && (!g2_on_curve(piB) || !g2_in_subgroup(piB))) bad = 2;   // the prover's only G2 element, first used by equation 2
    check(2, {Vkb, g1_neg(piBp)}, {piB, g2});                       // e(Vb, piB) == e(piB', g2)            :306-316
    check(3, {piC, g1_neg(piCp)}, {Vkc, g2});                       // e(piC, Vc) == e(piC', g2)            :318-328
    if (!bad) {
      G1Jac vkx = accumulate_ic(vk_ic, nic, public_signals, npublic);
      G1Jac vkx_a = g1j_add(vkx, g1j_from(piA));
      G1Aff xa = g1j_affine(vkx_a);
      check(4, {xa, g1_neg(piH), g1_neg(piC)}, {piB, Vkz, g2});    // e(Vkx+piA, piB) == e(piH, Vkz) e(piC, g2)   :336-347
      G1Aff xac = g1j_affine(g1j_add(vkx_a, g1j_from(piC)));
      check(5, {xac, G1Kbg, g1_neg(piKp)}, {G2Kbg, piB, G2Kg});    // e(Vkx+piA+piC, g2Kbg) e(g1Kbg, piB) == e(piK, g2Kg)   :353-363
    }
    *ok = bad ? 0 : 1;
    if (failed_check) *failed_check = bad;
    return GS_OK;
  });
}

}  // extern "C"
