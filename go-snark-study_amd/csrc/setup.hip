// groth16.GenerateTrustedSetup (groth16/groth16.go:94-222) for a SPARSE R1CS, on the device.
//
// The reference evaluates every dense QAP polynomial at tau (3 m Evals of n coefficients, each with an Exp per term)
// and encrypts the values with ~5 m naive MulScalar(G, .) calls.  Here:
//   L_j(tau), j = 1..n           Lagrange basis at tau over the interpolation nodes (barycentric weights from the
//                                cached node tree, one Fermat inversion per node, all in one kernel)
//   at_i = sum_j A[j][i] L_j     transposed sparse mat-vec (CSC built on the host in O(nnz)), likewise bt_i, ct_i
//   scalars of every key array   (kernels k_setup_scalars / k_scaled_powers)
//   points                       fixed-base batch multiplication (k_fixed_base_mul, 2^j G tables)
// and the result is a resident proving key (same object gs_groth16_pk_create builds) plus the verification key.
#include <algorithm>
#include <vector>

#include "point_io.h"
#include "prove.h"

using namespace gs;

namespace {

struct HostCsc {
  std::vector<uint32_t> colptr, rowidx;
  std::vector<uint64_t> vals;
};

// CSR (n rows x m columns) -> CSC; returns an error string or nullptr
const char* csr_to_csc(size_t n, size_t m, const uint32_t* rowptr, const uint32_t* col, const uint64_t* val, HostCsc& out) {
  if (rowptr[0] != 0) return "row_ptr[0] must be 0";
  for (size_t r = 0; r < n; ++r) if (rowptr[r + 1] < rowptr[r]) return "row_ptr not monotone";
  const size_t nnz = rowptr[n];
  if (nnz && (!col || !val)) return "null column/value array";
  out.colptr.assign(m + 1, 0);
  for (size_t e = 0; e < nnz; ++e) {
    if (col[e] >= m) return "column index out of range";
    out.colptr[col[e] + 1]++;
  }
  for (size_t i = 0; i < m; ++i) out.colptr[i + 1] += out.colptr[i];
  out.rowidx.resize(std::max<size_t>(nnz, 1));
  out.vals.resize(std::max<size_t>(nnz, 1) * 4);
  std::vector<uint32_t> cur(out.colptr.begin(), out.colptr.end() - 1);
  for (size_t r = 0; r < n; ++r)
    for (uint32_t e = rowptr[r]; e < rowptr[r + 1]; ++e) {
      const uint32_t pos = cur[col[e]]++;
      out.rowidx[pos] = (uint32_t)r;
      memcpy(&out.vals[(size_t)pos * 4], &val[(size_t)e * 4], 32);
    }
  return nullptr;
}

// per-variable evaluations x_i = sum_j M[j][i] * L_j : transposed SpMV through the CSC
void eval_columns(Ctx& c, const HostCsc& csc, size_t n, size_t m, const uint32_t* lag_mont, uint32_t* out_std) {
  const size_t nnz = csc.colptr[m];
  DevBuf cp((m + 1) * 4), ri(std::max<size_t>(nnz, 1) * 4), vl(std::max<size_t>(nnz, 1) * 32);
  GS_HIP(hipMemcpyAsync(cp.p, csc.colptr.data(), (m + 1) * 4, hipMemcpyHostToDevice, c.stream));
  if (nnz) {
    GS_HIP(hipMemcpyAsync(ri.p, csc.rowidx.data(), nnz * 4, hipMemcpyHostToDevice, c.stream));
    GS_HIP(hipMemcpyAsync(vl.p, csc.vals.data(), nnz * 32, hipMemcpyHostToDevice, c.stream));
  }
  spmv_dev(c, cp.as<uint32_t>(), ri.as<uint32_t>(), vl.as<uint32_t>(), lag_mont, m, n, out_std);
  GS_HIP(hipStreamSynchronize(c.stream));
}

void force_infinity_points(Ctx& c, DevBuf& pts, size_t count, size_t words) {
  if (count) GS_HIP(hipMemsetAsync(pts.p, 0, count * words * 4, c.stream));
}

// scalars of an evaluation-basis array: out[j-1] = scale * l_j(tau), l_j = Lagrange basis over the nodes n+1 .. 2n, j = 1..n
// (canonical standard form).  l_j(tau) = L_j(tau - n) with L over 1..n, so the node tree's weights serve.  false when tau is
// one of those nodes (the basis then degenerates; the key simply gets no evaluation-basis copy).
bool eval_basis_scalars(Ctx& c, size_t n, const uint64_t tau[4], const uint64_t scale_std[4], DevBuf& lag, DevBuf& out) {
  uint64_t nn[4] = {(uint64_t)n, 0, 0, 0}, tn[4], mtn[4];
  fr_sub_words(tau, nn, tn);
  fr_falling_product_words(tn, n, mtn);                       // prod_{k=1}^{n} (tau - n - k)
  if (fr_is_zero_words(mtn)) return false;
  lagrange_at_dev(c, n, tn, mtn, lag.as<uint32_t>());         // Montgomery
  scale_mont_by_std_dev(c, lag.as<uint32_t>(), scale_std, n, out.as<uint32_t>());
  return true;
}

template <class T>
Affine<T> download_point(Ctx& c, const uint32_t* packed_dev) {
  uint32_t w[PointIO<T>::kAffineWords];
  GS_HIP(hipMemcpyAsync(w, packed_dev, sizeof w, hipMemcpyDeviceToHost, c.stream));
  GS_HIP(hipStreamSynchronize(c.stream));
  return PointIO<T>::load_affine(w);
}

}  // namespace

template <class T>
static void host_affine_to_jacobian_std(const Affine<T>& a, uint64_t* out) {
  constexpr int cw = PointIO<T>::kCoordWords;
  uint32_t* o = reinterpret_cast<uint32_t*>(out);
  memset(o, 0, 3 * cw * 4);
  if (is_inf(a)) return;                  // infinity = all-zero triple (reference convention, g1.go:28-30)
  PointIO<T>::store_std(o, a.x);
  PointIO<T>::store_std(o + cw, a.y);
  o[2 * cw] = 1;                          // Z = 1 (G2: (1, 0))
}

extern "C" {

int gs_groth16_setup(size_t n, size_t m, size_t npublic,
                     const uint32_t* a_rowptr, const uint32_t* a_col, const uint64_t* a_val,
                     const uint32_t* b_rowptr, const uint32_t* b_col, const uint64_t* b_val,
                     const uint32_t* c_rowptr, const uint32_t* c_col, const uint64_t* c_val,
                     const uint64_t toxic[20], gs_handle* pk_out, uint64_t* vk_out) {
  return guarded([&](Ctx& c) -> int {
    if (!a_rowptr || !b_rowptr || !c_rowptr || !toxic || !pk_out) return fail(GS_ERR_ARG, "gs_groth16_setup: null argument");
    if (n == 0 || m < 2 || npublic + 1 > m) return fail(GS_ERR_SHAPE, "gs_groth16_setup: need n >= 1, m >= 2, NPublic + 1 <= m");
    if (n >= (1ull << 26) || m >= (1ull << 26)) return fail(GS_ERR_ARG, "gs_groth16_setup: system too large");
    // the prover indexes PowersTauDelta[i] for i < len(hx) = 2n - 1 - (m - 1) + 1 (groth16.go:269-271): m in {n+1, n+2}
    if (2 * n + 1 < m || 2 * n - m + 1 > m - 1) return fail(GS_ERR_SHAPE, "gs_groth16_setup: len(hx) = 2n - m + 1 would exceed len(PowersTauDelta) = m - 1 (SURVEY fact 8)");
    // Z(x) = prod_{k=1}^{m-2} (x - k) (groth16.go:122-131): for m > n + 2 it has roots at nodes beyond the n constraints, where
    // A B - C does not vanish -- no witness could ever produce a verifying proof with such a key.  (m = n + 1 leaves constraint n
    // outside Z: the reference's own gap, kept for parity and stated in DESIGN.md.)
    if (m > n + 2) return fail(GS_ERR_SHAPE, "gs_groth16_setup: m = %zu variables for n = %zu constraints: Z would have %zu roots, more than there are constraints (need m <= n + 2)", m, n, m - 2);
    const uint64_t* T = toxic;
    const uint64_t* Kalpha = toxic + 4;
    const uint64_t* Kbeta = toxic + 8;
    const uint64_t* Kgamma = toxic + 12;
    const uint64_t* Kdelta = toxic + 16;
    if (fr_is_zero_words(Kgamma) || fr_is_zero_words(Kdelta)) return fail(GS_ERR_ARG, "gs_groth16_setup: gamma and delta must be invertible");
    HostCsc ca, cb, cc;
    if (const char* e = csr_to_csc(n, m, a_rowptr, a_col, a_val, ca)) return fail(GS_ERR_ARG, "gs_groth16_setup: A: %s", e);
    if (const char* e = csr_to_csc(n, m, b_rowptr, b_col, b_val, cb)) return fail(GS_ERR_ARG, "gs_groth16_setup: B: %s", e);
    if (const char* e = csr_to_csc(n, m, c_rowptr, c_col, c_val, cc)) return fail(GS_ERR_ARG, "gs_groth16_setup: C: %s", e);
    // M(tau) = prod_{k=1}^{n} (tau - k) and Z(tau) = prod_{k=1}^{m-2} (tau - k)                 [groth16.go:122-133]
    uint64_t mt[4], zt[4], inv_delta[4], inv_gamma[4], zt_inv_delta[4];
    fr_falling_product_words(T, n, mt);
    fr_falling_product_words(T, m - 2, zt);
    if (fr_is_zero_words(mt)) return fail(GS_ERR_ARG, "gs_groth16_setup: tau collides with an interpolation node");
    fr_inv_words(Kdelta, inv_delta);
    fr_inv_words(Kgamma, inv_gamma);
    fr_mul_words(inv_delta, zt, zt_inv_delta);
    // --- evaluations at tau ---------------------------------------------------------------------------------
    DevBuf lag(n * 32), at(m * 32), bt(m * 32), ct(m * 32), cd(m * 32), ic(m * 32), pw(std::max<size_t>(m - 1, 1) * 32);
    lagrange_at_dev(c, n, T, mt, lag.as<uint32_t>());
    eval_columns(c, ca, n, m, lag.as<uint32_t>(), at.as<uint32_t>());           // at_i = alphas[i](tau)   :163
    eval_columns(c, cb, n, m, lag.as<uint32_t>(), bt.as<uint32_t>());           // bt_i                    :167
    eval_columns(c, cc, n, m, lag.as<uint32_t>(), ct.as<uint32_t>());           // ct_i                    :185
    poly_canon_dev(c, at.as<uint32_t>(), m, 0);
    poly_canon_dev(c, bt.as<uint32_t>(), m, 0);
    setup_scalars_dev(c, at.as<uint32_t>(), bt.as<uint32_t>(), ct.as<uint32_t>(), m, npublic, Kalpha, Kbeta, inv_delta, inv_gamma,
                      cd.as<uint32_t>(), ic.as<uint32_t>());
    scaled_powers_dev(c, T, zt_inv_delta, m - 1, pw.as<uint32_t>());            // tau^i Z(tau) / delta    :139-149
    // --- encryption: k * G batches --------------------------------------------------------------------------
    auto pk = std::make_unique<GrothPkObj>();
    pk->nvars = m; pk->npublic = npublic; pk->nz = m - 1; pk->nptd = m - 1; pk->n_w = m; pk->n_h = m - 1;
    pk->at.alloc(m * 64); pk->bacgamma1.alloc(m * 64); pk->bacdelta.alloc(m * 64); pk->ptd.alloc(std::max<size_t>(m - 1, 1) * 64);
    pk->bacgamma2.alloc(m * 128);
    fixed_base_g1(c, at.as<uint32_t>(), (uint32_t)m, pk->at.as<uint32_t>());                    // Pk.G1.At        :164-165
    fixed_base_g1(c, bt.as<uint32_t>(), (uint32_t)m, pk->bacgamma1.as<uint32_t>());             // Pk.G1.BACGamma  :168,171
    fixed_base_g2(c, bt.as<uint32_t>(), (uint32_t)m, pk->bacgamma2.as<uint32_t>());             // Pk.G2.BACGamma  :169,173
    fixed_base_g1(c, cd.as<uint32_t>(), (uint32_t)m, pk->bacdelta.as<uint32_t>());              // Pk.BACDelta     :177-200 (i <= NPublic: infinity)
    fixed_base_g1(c, pw.as<uint32_t>(), (uint32_t)(m - 1), pk->ptd.as<uint32_t>());             // PowersTauDelta  :139-149
    // The same group element through H's VALUES (prove.h): ptd_eval[j-1] = l_j(tau) Z(tau) / delta * G with l_j the Lagrange basis
    // over the nodes n+1 .. 2n, i.e. L_j(tau - n) over 1 .. n -- one more fixed-base batch while tau is still known.
    DevBuf lag2(n * 32), qe(n * 32);
    if (eval_basis_scalars(c, n, T, zt_inv_delta, lag2, qe)) {
      pk->ptd_eval.alloc(n * 64);
      fixed_base_g1(c, qe.as<uint32_t>(), (uint32_t)n, pk->ptd_eval.as<uint32_t>());
      pk->n_eval = n; pk->e_lo = 0; pk->n_e = n;
    }
    // single points: alpha, beta, delta in G1; beta, gamma, delta in G2                          :151-160
    DevBuf s1(3 * 32), s2(3 * 32), p1(3 * 64), p2(3 * 128);
    uint64_t h1[12], h2[12];
    memcpy(h1, Kalpha, 32); memcpy(h1 + 4, Kbeta, 32); memcpy(h1 + 8, Kdelta, 32);
    memcpy(h2, Kbeta, 32); memcpy(h2 + 4, Kgamma, 32); memcpy(h2 + 8, Kdelta, 32);
    GS_HIP(hipMemcpyAsync(s1.p, h1, 96, hipMemcpyHostToDevice, c.stream));
    GS_HIP(hipMemcpyAsync(s2.p, h2, 96, hipMemcpyHostToDevice, c.stream));
    fixed_base_g1(c, s1.as<uint32_t>(), 3, p1.as<uint32_t>());
    fixed_base_g2(c, s2.as<uint32_t>(), 3, p2.as<uint32_t>());
    pk->alpha = download_point<FqTag>(c, p1.as<uint32_t>());
    pk->beta = download_point<FqTag>(c, p1.as<uint32_t>() + 16);
    pk->delta = download_point<FqTag>(c, p1.as<uint32_t>() + 32);
    pk->beta2 = download_point<Fq2Tag>(c, p2.as<uint32_t>());
    pk->delta2 = download_point<Fq2Tag>(c, p2.as<uint32_t>() + 64);
    // Pk.Z                                                                                       :122-131
    DevBuf zc((m - 1) * 32);
    zpoly_dev(c, m - 2, zc.as<uint32_t>());
    divisor_init(c, pk->z, zc.as<uint32_t>(), m - 1);
    // --- verification key (alpha | beta2 | gamma2 | delta2 | IC[0..NPublic]) as affine Jacobian triples --------
    if (vk_out) {
      const size_t nic = npublic + 1;
      DevBuf icp(nic * 64), j1((1 + nic) * 96), j2(3 * 192);
      fixed_base_g1(c, ic.as<uint32_t>(), (uint32_t)nic, icp.as<uint32_t>());                   // Vk.IC           :202-219
      affine_to_jacobian_std_g1(c, p1.as<uint32_t>(), 1, j1.as<uint32_t>());
      affine_to_jacobian_std_g1(c, icp.as<uint32_t>(), (uint32_t)nic, j1.as<uint32_t>() + 24);
      affine_to_jacobian_std_g2(c, p2.as<uint32_t>(), 3, j2.as<uint32_t>());
      GS_HIP(hipMemcpyAsync(vk_out, j1.p, 96, hipMemcpyDeviceToHost, c.stream));
      GS_HIP(hipMemcpyAsync(vk_out + 12, j2.p, 3 * 192, hipMemcpyDeviceToHost, c.stream));
      GS_HIP(hipMemcpyAsync(vk_out + 12 + 72, j1.as<uint32_t>() + 24, nic * 96, hipMemcpyDeviceToHost, c.stream));
    }
    GS_HIP(hipStreamSynchronize(c.stream));
    groth_pk_scan_sparsity(c, *pk);
    *pk_out = c.put(std::move(pk));
    return GS_OK;
  });
}

// snark.GenerateTrustedSetup (snark.go:98-251) for a sparse R1CS, toxic = T | Ka | Kb | Kc | Kbeta | Kgamma | RhoA | RhoB.
int gs_pinocchio_setup(size_t n, size_t m, size_t npublic,
                       const uint32_t* a_rowptr, const uint32_t* a_col, const uint64_t* a_val,
                       const uint32_t* b_rowptr, const uint32_t* b_col, const uint64_t* b_val,
                       const uint32_t* c_rowptr, const uint32_t* c_col, const uint64_t* c_val,
                       const uint64_t toxic[32], gs_handle* pk_out, uint64_t* vk_out) {
  return guarded([&](Ctx& c) -> int {
    if (!a_rowptr || !b_rowptr || !c_rowptr || !toxic || !pk_out) return fail(GS_ERR_ARG, "gs_pinocchio_setup: null argument");
    if (n == 0 || m < 2 || npublic + 1 > m) return fail(GS_ERR_SHAPE, "gs_pinocchio_setup: need n >= 1, m >= 2, NPublic + 1 <= m");
    if (n >= (1ull << 26) || m >= (1ull << 26)) return fail(GS_ERR_ARG, "gs_pinocchio_setup: system too large");
    if (m < n + 1 || m > 2 * n + 1) return fail(GS_ERR_SHAPE, "gs_pinocchio_setup: need n + 1 <= m <= 2n + 1 (len(hx) must fit len(G1T), snark.go:284-286)");
    if (m > n + 2) return fail(GS_ERR_SHAPE, "gs_pinocchio_setup: m = %zu variables for n = %zu constraints: Z would have %zu roots, more than there are constraints (need m <= n + 2)", m, n, m - 2);
    const uint64_t *T = toxic, *Ka = toxic + 4, *Kb = toxic + 8, *Kc = toxic + 12, *Kbeta = toxic + 16, *Kgamma = toxic + 20, *RhoA = toxic + 24,
                   *RhoB = toxic + 28;
    HostCsc ca, cb, cc;
    if (const char* e = csr_to_csc(n, m, a_rowptr, a_col, a_val, ca)) return fail(GS_ERR_ARG, "gs_pinocchio_setup: A: %s", e);
    if (const char* e = csr_to_csc(n, m, b_rowptr, b_col, b_val, cb)) return fail(GS_ERR_ARG, "gs_pinocchio_setup: B: %s", e);
    if (const char* e = csr_to_csc(n, m, c_rowptr, c_col, c_val, cc)) return fail(GS_ERR_ARG, "gs_pinocchio_setup: C: %s", e);
    uint64_t mt[4], zt[4], rhoc[4], kbg[4], rhoczt[4];
    fr_falling_product_words(T, n, mt);
    fr_falling_product_words(T, m - 2, zt);
    if (fr_is_zero_words(mt)) return fail(GS_ERR_ARG, "gs_pinocchio_setup: tau collides with an interpolation node");
    fr_mul_words(RhoA, RhoB, rhoc);                       // :149
    fr_mul_words(Kbeta, Kgamma, kbg);                     // :172
    fr_mul_words(rhoc, zt, rhoczt);                       // :233-235
    DevBuf lag(n * 32), at(m * 32), bt(m * 32), ct(m * 32), sc[7], pw((m - 1) * 32);
    lagrange_at_dev(c, n, T, mt, lag.as<uint32_t>());
    eval_columns(c, ca, n, m, lag.as<uint32_t>(), at.as<uint32_t>());
    eval_columns(c, cb, n, m, lag.as<uint32_t>(), bt.as<uint32_t>());
    eval_columns(c, cc, n, m, lag.as<uint32_t>(), ct.as<uint32_t>());
    uint32_t* outs[7];
    for (int i = 0; i < 7; ++i) { sc[i].alloc(m * 32); outs[i] = sc[i].as<uint32_t>(); }
    pinocchio_scalars_dev(c, at.as<uint32_t>(), bt.as<uint32_t>(), ct.as<uint32_t>(), m, RhoA, RhoB, rhoc, Ka, Kb, Kc, Kbeta, outs);
    const uint64_t one[4] = {1, 0, 0, 0};
    scaled_powers_dev(c, T, one, m - 1, pw.as<uint32_t>());                          // G1T_i = tau^i G1      :239-247
    auto pk = std::make_unique<PinocchioPkObj>();
    pk->nvars = m; pk->npublic = npublic; pk->nz = m - 1; pk->ng1t = m - 1;
    pk->n_w = m; pk->n_h = m - 1;                                                    // a full key
    DevBuf* g1dst[7] = {&pk->a, nullptr, &pk->c, &pk->ap, &pk->bp, &pk->cp, &pk->kp};   // sb -> B lives in G2
    for (int i = 0; i < 7; ++i) {
      if (!g1dst[i]) continue;
      g1dst[i]->alloc(m * 64);
      fixed_base_g1(c, outs[i], (uint32_t)m, g1dst[i]->as<uint32_t>());
    }
    pk->b2.alloc(m * 128);
    fixed_base_g2(c, outs[1], (uint32_t)m, pk->b2.as<uint32_t>());                   // Pk.B                 :192-194
    pk->g1t.alloc((m - 1) * 64);
    fixed_base_g1(c, pw.as<uint32_t>(), (uint32_t)(m - 1), pk->g1t.as<uint32_t>());
    // evaluation-basis copy of G1T (prove.h): g1t_eval[j-1] = l_j(tau) * G over the nodes n+1 .. 2n
    DevBuf lag2(n * 32), qe(n * 32);
    if (eval_basis_scalars(c, n, T, one, lag2, qe)) {
      pk->g1t_eval.alloc(n * 64);
      fixed_base_g1(c, qe.as<uint32_t>(), (uint32_t)n, pk->g1t_eval.as<uint32_t>());
      pk->n_eval = n; pk->e_lo = 0; pk->n_e = n;
    }
    DevBuf zc((m - 1) * 32);
    zpoly_dev(c, m - 2, zc.as<uint32_t>());
    divisor_init(c, pk->z, zc.as<uint32_t>(), m - 1);
    if (vk_out) {
      // Vka (G2) | Vkb (G1) | Vkc (G2) | G1Kbg | G2Kbg | G2Kg | Vkz | IC[0..NPublic]       :162-175, 186-188, 236
      const size_t nic = npublic + 1;
      uint64_t s1h[8], s2h[20];
      memcpy(s1h, Kb, 32); memcpy(s1h + 4, kbg, 32);
      memcpy(s2h, Ka, 32); memcpy(s2h + 4, Kc, 32); memcpy(s2h + 8, kbg, 32); memcpy(s2h + 12, Kgamma, 32); memcpy(s2h + 16, rhoczt, 32);
      DevBuf s1(64), s2(160), p1((2 + nic) * 64), p2(5 * 128), j1((2 + nic) * 96), j2(5 * 192);
      GS_HIP(hipMemcpyAsync(s1.p, s1h, 64, hipMemcpyHostToDevice, c.stream));
      GS_HIP(hipMemcpyAsync(s2.p, s2h, 160, hipMemcpyHostToDevice, c.stream));
      fixed_base_g1(c, s1.as<uint32_t>(), 2, p1.as<uint32_t>());
      GS_HIP(hipMemcpyAsync(p1.as<uint32_t>() + 32, pk->a.p, nic * 64, hipMemcpyDeviceToDevice, c.stream));    // IC = A[0..NPublic]
      fixed_base_g2(c, s2.as<uint32_t>(), 5, p2.as<uint32_t>());
      affine_to_jacobian_std_g1(c, p1.as<uint32_t>(), (uint32_t)(2 + nic), j1.as<uint32_t>());
      affine_to_jacobian_std_g2(c, p2.as<uint32_t>(), 5, j2.as<uint32_t>());
      uint32_t* o = reinterpret_cast<uint32_t*>(vk_out);
      const uint32_t* J1 = j1.as<uint32_t>();
      const uint32_t* J2 = j2.as<uint32_t>();
      auto cp = [&](uint32_t* dst, const uint32_t* src, size_t words) { GS_HIP(hipMemcpyAsync(dst, src, words * 4, hipMemcpyDeviceToHost, c.stream)); };
      cp(o, J2, 48);                    // Vka
      cp(o + 48, J1, 24);               // Vkb
      cp(o + 72, J2 + 48, 48);          // Vkc
      cp(o + 120, J1 + 24, 24);         // G1Kbg
      cp(o + 144, J2 + 96, 48);         // G2Kbg
      cp(o + 192, J2 + 144, 48);        // G2Kg
      cp(o + 240, J2 + 192, 48);        // Vkz
      cp(o + 288, J1 + 48, nic * 24);   // IC
      GS_HIP(hipStreamSynchronize(c.stream));
    }
    force_infinity_points(c, pk->a, npublic + 1, 16);                               // the prover sums A, Ap over i > NPublic (snark.go:265)
    force_infinity_points(c, pk->ap, npublic + 1, 16);
    GS_HIP(hipStreamSynchronize(c.stream));
    pinocchio_pk_scan_sparsity(c, *pk);
    *pk_out = c.put(std::move(pk));
    return GS_OK;
  });
}

int gs_groth16_pk_export(gs_handle hpk, int which, uint64_t* jacobian, size_t count) {
  return guarded([&](Ctx& c) -> int {
    GrothPkObj* pk = c.get<GrothPkObj>(hpk, Kind::GrothPk);
    if (!pk) return fail(GS_ERR_ARG, "gs_groth16_pk_export: bad proving-key handle");
    const DevBuf* src = nullptr;
    size_t have = pk->n_w;            // a key slice exports the entries it holds
    bool g2 = false;
    if (which == 5) {           // the single elements, Jacobian: G1 alpha, beta, delta (3 x 12) then G2 beta, delta (2 x 24)
      if (count != 5 || !jacobian) return fail(GS_ERR_ARG, "gs_groth16_pk_export: which = 5 exports exactly 5 points (84 u64)");
      host_affine_to_jacobian_std<FqTag>(pk->alpha, jacobian);
      host_affine_to_jacobian_std<FqTag>(pk->beta, jacobian + 12);
      host_affine_to_jacobian_std<FqTag>(pk->delta, jacobian + 24);
      host_affine_to_jacobian_std<Fq2Tag>(pk->beta2, jacobian + 36);
      host_affine_to_jacobian_std<Fq2Tag>(pk->delta2, jacobian + 60);
      return GS_OK;
    }
    if (which == 6) {           // pk.Z: nz coefficients, 4 x u64 each
      if (count != pk->nz || !jacobian) return fail(GS_ERR_ARG, "gs_groth16_pk_export: Z has %zu coefficients, asked for %zu", pk->nz, count);
      GS_HIP(hipMemcpyAsync(jacobian, pk->z.b_std.p, count * 32, hipMemcpyDeviceToHost, c.stream));
      GS_HIP(hipStreamSynchronize(c.stream));
      return GS_OK;
    }
    switch (which) {
      case 0: src = &pk->at; break;
      case 1: src = &pk->bacgamma1; break;
      case 2: src = &pk->bacgamma2; g2 = true; break;
      case 3: src = &pk->bacdelta; break;
      case 4: src = &pk->ptd; have = pk->n_h; break;
      case 7: src = &pk->ptd_eval; have = pk->n_e; break;       // evaluation-basis copy of PowersTauDelta (0 points when the key has none)
      default: return fail(GS_ERR_ARG, "gs_groth16_pk_export: which must be 0..7");
    }
    if (count != have || (count && !jacobian)) return fail(GS_ERR_ARG, "gs_groth16_pk_export: array has %zu points, asked for %zu", have, count);
    if (!count) return GS_OK;
    const size_t words = g2 ? 48 : 24;
    DevBuf tmp(count * words * 4);
    if (g2) affine_to_jacobian_std_g2(c, src->as<uint32_t>(), (uint32_t)count, tmp.as<uint32_t>());
    else affine_to_jacobian_std_g1(c, src->as<uint32_t>(), (uint32_t)count, tmp.as<uint32_t>());
    GS_HIP(hipMemcpyAsync(jacobian, tmp.p, count * words * 4, hipMemcpyDeviceToHost, c.stream));
    GS_HIP(hipStreamSynchronize(c.stream));
    return GS_OK;
  });
}

int gs_pinocchio_pk_export(gs_handle hpk, int which, uint64_t* jacobian, size_t count) {
  return guarded([&](Ctx& c) -> int {
    PinocchioPkObj* pk = c.get<PinocchioPkObj>(hpk, Kind::PinocchioPk);
    if (!pk) return fail(GS_ERR_ARG, "gs_pinocchio_pk_export: bad proving-key handle");
    if (pk->shard_count != 1) return fail(GS_ERR_ARG, "gs_pinocchio_pk_export: the key is a slice (export the full key)");
    const DevBuf* arr[10] = {&pk->a, &pk->ap, &pk->b2, &pk->bp, &pk->c, &pk->cp, &pk->kp, &pk->g1t, nullptr, &pk->g1t_eval};
    if (which == 8) {           // pk.Z: nz coefficients, 4 x u64 each
      if (count != pk->nz || !jacobian) return fail(GS_ERR_ARG, "gs_pinocchio_pk_export: Z has %zu coefficients, asked for %zu", pk->nz, count);
      GS_HIP(hipMemcpyAsync(jacobian, pk->z.b_std.p, count * 32, hipMemcpyDeviceToHost, c.stream));
      GS_HIP(hipStreamSynchronize(c.stream));
      return GS_OK;
    }
    if (which < 0 || which > 9) return fail(GS_ERR_ARG, "gs_pinocchio_pk_export: which must be 0..9");
    const bool g2 = which == 2;
    const size_t have = which == 7 ? pk->ng1t : which == 9 ? pk->n_eval : pk->nvars;     // 9: evaluation-basis copy of G1T (0 points when there is none)
    if (count != have || (count && !jacobian)) return fail(GS_ERR_ARG, "gs_pinocchio_pk_export: array has %zu points, asked for %zu", have, count);
    if (!count) return GS_OK;
    const size_t words = g2 ? 48 : 24;
    DevBuf tmp(count * words * 4);
    if (g2) affine_to_jacobian_std_g2(c, arr[which]->as<uint32_t>(), (uint32_t)count, tmp.as<uint32_t>());
    else affine_to_jacobian_std_g1(c, arr[which]->as<uint32_t>(), (uint32_t)count, tmp.as<uint32_t>());
    GS_HIP(hipMemcpyAsync(jacobian, tmp.p, count * words * 4, hipMemcpyDeviceToHost, c.stream));
    GS_HIP(hipStreamSynchronize(c.stream));
    return GS_OK;
  });
}

// Attach an evaluation-basis array to a key that was not built here (a key file that carries one: utils.py's binary container,
// sections "PowersTauDeltaEval" / "G1TEval"): `bases` holds n_constraints G1 points,
//   Groth16:   l_j(tau) Z(tau) / delta * G,     Pinocchio:   l_j(tau) * G        (l_j: Lagrange basis over the nodes n+1 .. 2n).
// The library cannot check the points against tau (nobody knows tau any more); a wrong array gives proofs that do not verify,
// exactly as a wrong PowersTauDelta does.  n_constraints must be len(Z) - 1 or len(Z).
int gs_groth16_pk_set_eval(gs_handle hpk, gs_handle hbases) {
  return guarded([&](Ctx& c) -> int {
    GrothPkObj* pk = c.get<GrothPkObj>(hpk, Kind::GrothPk);
    Bases* b = c.get<Bases>(hbases, Kind::G1Bases);
    if (!pk || !b) return fail(GS_ERR_ARG, "gs_groth16_pk_set_eval: bad handle");
    if (pk->shard_count != 1) return fail(GS_ERR_ARG, "gs_groth16_pk_set_eval: the key is a slice");
    const size_t n = b->n;
    if (n < 2 || pk->nz == 0 || (pk->nz - 1 != n - 1 && pk->nz - 1 != n))
      return fail(GS_ERR_SHAPE, "gs_groth16_pk_set_eval: %zu points, but deg Z = %zu needs n = deg Z or deg Z + 1 constraints", n, pk->nz ? pk->nz - 1 : 0);
    table_settle(c, pk->t_ptd_eval, false);
    pk->t_ptd_eval.drop();
    pk->ptd_eval.alloc(n * 64);
    GS_HIP(hipMemcpyAsync(pk->ptd_eval.p, b->buf.p, n * 64, hipMemcpyDeviceToDevice, c.stream));
    GS_HIP(hipStreamSynchronize(c.stream));
    pk->n_eval = n; pk->e_lo = 0; pk->n_e = n;
    return GS_OK;
  }, true, false, hpk);
}
// number of evaluation-basis points a resident Groth16 or Pinocchio key holds (0 = none; a slice: its own share)
int gs_pk_eval_count(gs_handle hpk, size_t* count) {
  return guarded([&](Ctx& c) -> int {
    if (!count) return fail(GS_ERR_ARG, "gs_pk_eval_count: null output");
    if (GrothPkObj* g = c.get<GrothPkObj>(hpk, Kind::GrothPk)) { *count = g->n_e; return GS_OK; }
    if (PinocchioPkObj* p = c.get<PinocchioPkObj>(hpk, Kind::PinocchioPk)) { *count = p->n_e; return GS_OK; }
    return fail(GS_ERR_ARG, "gs_pk_eval_count: not a proving-key handle");
  }, true, true, hpk);
}
int gs_pinocchio_pk_set_eval(gs_handle hpk, gs_handle hbases) {
  return guarded([&](Ctx& c) -> int {
    PinocchioPkObj* pk = c.get<PinocchioPkObj>(hpk, Kind::PinocchioPk);
    Bases* b = c.get<Bases>(hbases, Kind::G1Bases);
    if (!pk || !b) return fail(GS_ERR_ARG, "gs_pinocchio_pk_set_eval: bad handle");
    if (pk->shard_count != 1) return fail(GS_ERR_ARG, "gs_pinocchio_pk_set_eval: the key is a slice (attach the array to the full key, then cut it)");
    const size_t n = b->n;
    if (n < 2 || pk->nz == 0 || (pk->nz - 1 != n - 1 && pk->nz - 1 != n))
      return fail(GS_ERR_SHAPE, "gs_pinocchio_pk_set_eval: %zu points, but deg Z = %zu needs n = deg Z or deg Z + 1 constraints", n, pk->nz ? pk->nz - 1 : 0);
    table_settle(c, pk->t_g1t_eval, false);
    pk->t_g1t_eval.drop();
    pk->g1t_eval.alloc(n * 64);
    GS_HIP(hipMemcpyAsync(pk->g1t_eval.p, b->buf.p, n * 64, hipMemcpyDeviceToDevice, c.stream));
    GS_HIP(hipStreamSynchronize(c.stream));
    pk->n_eval = n; pk->e_lo = 0; pk->n_e = n;
    return GS_OK;
  }, true, false, hpk);
}

}  // extern "C"
