// Internal C++ interface of the Fr polynomial engine (implemented in poly.hip).
// All pointers are DEVICE pointers to packed elements (8 x u32 words each, value < 2^256).
#pragma once
#include "fp29.h"
#include "runtime.h"

namespace gs {

enum class Form { Std, Mont };

// smallest power of two >= n (n >= 1) and its log2
int ceil_log2(size_t n);

// Forward NTT of `total` = 2^logt elements, as total / 2^logm independent transforms of size 2^logm
// (natural order in, bit-reversed out).  In place.
void ntt_forward(Ctx& c, uint32_t* data, int logt, int logm);
// Inverse of the above (bit-reversed in, natural out), WITHOUT the 1/N scaling.
void ntt_inverse_unscaled(Ctx& c, uint32_t* data, int logt, int logm);
// the same for any element count that is a multiple of 2^logm
void ntt_forward_n(Ctx& c, uint32_t* data, size_t total, int logm);
void ntt_inverse_unscaled_n(Ctx& c, uint32_t* data, size_t total, int logm);

// out (na + nb - 1 coefficients) = a * b.  fa/fb: form of the inputs; the output is Montgomery iff
// both inputs are, standard otherwise.  out may not alias the inputs.  out_cap >= na + nb - 1.
void poly_mul_dev(Ctx& c, const uint32_t* a, size_t na, Form fa, const uint32_t* b, size_t nb, Form fb, uint32_t* out);

// g = 1 / f mod x^k for f given in MONTGOMERY form (nf coefficients, f[0] != 0); g Montgomery, k coeffs.
void poly_inv_series_dev(Ctx& c, const uint32_t* f_mont, size_t nf, size_t k, uint32_t* g_mont);

// Cached divisor for repeated quotients by the same monic-or-not polynomial b (e.g. pk.Z):
struct Divisor {
  size_t nb = 0;
  DevBuf b_std;            // nb coefficients, standard form (for remainders)
  size_t k = 0;            // inverse series known to this many coefficients
  DevBuf inv_rev_mont;     // 1 / rev(b) mod x^k, Montgomery
  int logn_spec = 0;       // spectrum cached for NTT size 2^logn_spec (0 = none)
  size_t k_spec = 0;
  DevBuf inv_spec;         // NTT of inv_rev_mont[:k_spec] zero padded (bit-reversed order)
};
void divisor_init(Ctx& c, Divisor& d, const uint32_t* b_std_dev, size_t nb);
// quo (na - nb + 1 coefficients, standard form, values < 2r) = floor(a / b); a standard form.
void poly_quotient_dev(Ctx& c, Divisor& d, const uint32_t* a_std, size_t na, uint32_t* quo_std);

// Z(x) = prod_{i=1}^{deg} (x - i), deg + 1 canonical standard-form coefficients (subproduct tree of NTT products)
void zpoly_dev(Ctx& c, size_t deg, uint32_t* out_std);

// Lagrange interpolation on the nodes 1..n of nvec value vectors (nvec x n, standard form) -> nvec x n coefficients
// (standard form, values < 2r); O(n log^2 n) on a cached subproduct tree.
void interpolate_dev(Ctx& c, const uint32_t* values_std, size_t n, size_t nvec, uint32_t* coeffs_std);
// H(x) = (A(x) B(x) - C(x)) / Z(x) straight from the constraint values [A w | B w | C w] (n each, standard form) when the
// witness satisfies the constraints at the dz = deg Z roots of Z: node extension by one batched convolution, ONE tree
// interpolation, one Taylor shift.  false (nothing written) when a constraint is violated -- take the px route then.
bool hx_direct_dev(Ctx& c, const uint32_t* vals_std, size_t n, size_t dz, uint32_t* hx_out_std);
// The two halves of the above, for keys that carry an evaluation-basis copy of PowersTauDelta / G1T (prove.h): the prover then
// needs only H's VALUES at the nodes n+1..2n.  hx_values_dev: hv[k-1] = H(n+k), k = 1..n, canonical standard form (false: shape
// not served); r1cs_check_dev: *bad_dev = number of roots of Z at which a b != c (enqueue only -- the caller reads the word when
// it collects the proof, and takes the exact route if it is not zero); hx_from_values_dev: values -> coefficients.
bool hx_values_dev(Ctx& c, const uint32_t* vals_std, size_t n, size_t dz, uint32_t* hv_out_std);
void r1cs_check_dev(Ctx& c, const uint32_t* vals_std, size_t n, size_t dz, uint32_t* bad_dev);
void hx_from_values_dev(Ctx& c, const uint32_t* hv_std, size_t n, size_t dz, uint32_t* hx_out_std);
// CSR sparse matrix (standard-form values) times a Montgomery-form vector -> standard-form vector
void spmv_dev(Ctx& c, const uint32_t* rowptr, const uint32_t* col, const uint32_t* val_std, const uint32_t* x_mont, size_t nrows, size_t ncols,
              uint32_t* out_std);

void poly_addsub_dev(Ctx& c, const uint32_t* a, size_t na, const uint32_t* b, size_t nb, bool subtract, uint32_t* out);
void poly_canon_dev(Ctx& c, uint32_t* x, size_t n, int mode /* 0 canon, 1 to-Montgomery, 2 from-Montgomery */);
// out[0] = sum_i v[i] x^i (standard in, canonical standard out); x given as ABI words
void poly_eval_dev(Ctx& c, const uint32_t* v_std, size_t n, const uint64_t x[4], uint32_t* out_dev);

// host-side Fr helpers (same arithmetic as the kernels)
Fe<ModR, 2> fr_from_words_mont(const uint64_t w[4]);
void fr_words_from_mont(const Fe<ModR, 2>& a, uint64_t out[4]);
void fr_mul_words(const uint64_t a[4], const uint64_t b[4], uint64_t out[4]);
void fr_inv_words(const uint64_t a[4], uint64_t out[4]);
void fr_sub_words(const uint64_t a[4], const uint64_t b[4], uint64_t out[4]);
bool fr_is_zero_words(const uint64_t a[4]);
void fr_falling_product_words(const uint64_t x[4], size_t count, uint64_t out[4]);     // prod_{k=1}^{count} (x - k)

// trusted-setup building blocks (setup.hip)
const uint32_t* interpolation_weights_dev(Ctx& c, size_t n);                           // 1 / M'(j), nodes 1..n, Montgomery
void lagrange_at_dev(Ctx& c, size_t n, const uint64_t tau[4], const uint64_t mtau[4], uint32_t* out_mont);   // L_j(tau), j = 1..n
void setup_scalars_dev(Ctx& c, const uint32_t* at, const uint32_t* bt, const uint32_t* ct, size_t m, size_t npublic, const uint64_t kalpha[4],
                       const uint64_t kbeta[4], const uint64_t inv_delta[4], const uint64_t inv_gamma[4], uint32_t* cd, uint32_t* ic);
// out[0..6] = sa, sb, sc, sap, sbp, scp, skp (standard form), see k_pinocchio_scalars
void pinocchio_scalars_dev(Ctx& c, const uint32_t* at, const uint32_t* bt, const uint32_t* ct, size_t m, const uint64_t rhoa[4], const uint64_t rhob[4],
                           const uint64_t rhoc[4], const uint64_t ka[4], const uint64_t kb[4], const uint64_t kc[4], const uint64_t kbeta[4],
                           uint32_t* const out[7]);
void scaled_powers_dev(Ctx& c, const uint64_t base[4], const uint64_t scale_std[4], size_t count, uint32_t* out_std);
// out[i] = a_mont[i] * scale (scale in standard words), canonical standard form
void scale_mont_by_std_dev(Ctx& c, const uint32_t* a_mont, const uint64_t scale_std[4], size_t n, uint32_t* out_std);

}  // namespace gs
