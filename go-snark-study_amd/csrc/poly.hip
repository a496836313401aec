// Fr polynomial engine: host orchestration of poly_kernels.h.
#include "poly.h"

#include <algorithm>
#include <vector>

#include "poly_kernels.h"

namespace gs {

static inline dim3 grid1(size_t n, int block = 256) { return dim3((unsigned)((n + block - 1) / block)); }

int ceil_log2(size_t n) {
  int l = 0;
  while (((size_t)1 << l) < n) ++l;
  return l;
}

static FrConst to_const(const Fe<ModR, 2>& a) {
  FrConst c;
  for (int i = 0; i < NL; ++i) c.l[i] = a.l[i];
  return c;
}
Fe<ModR, 2> fr_from_words_mont(const uint64_t w[4]) {
  uint32_t u[8];
  for (int i = 0; i < 4; ++i) { u[2 * i] = (uint32_t)w[i]; u[2 * i + 1] = (uint32_t)(w[i] >> 32); }
  return to_mont(unpack32<ModR>(u));
}
static Fe<ModR, 2> fr_small_mont(uint64_t v) {
  uint64_t w[4] = {v, 0, 0, 0};
  return fr_from_words_mont(w);
}

// ---- twiddle tables ---------------------------------------------------------------------------------
struct Twiddles {
  int logn = 0;              // tables serve transforms up to 2^logn
  DevBuf fwd, inv;           // omega^i, omega^-i for i < 2^(logn-1)
};
// (node trees: see the subproduct-tree section below)
struct NodeTree {
  size_t n = 0, total = 0, pad = 0;
  int L = 0;
  std::vector<DevBuf> mspec;     // level j < L: spectra of the 2d-padded blocks, Montgomery, bit-reversed within each 2d block
  DevBuf root;                   // low 2^L coefficients of x^pad * prod_{i=1}^{n} (x - i), Montgomery
  DevBuf weights;                // n elements, Montgomery: 1 / prod_{k != j} (j - k), nodes j = 1..n
  uint64_t stamp = 0;
};

// Everything the engine keeps between calls, per context (device memory belongs to one device).
struct PolyState {
  Twiddles tw;
  DevBuf ws_a, ws_b, ws_c, ws_d;                 // product / quotient / series workspaces
  NodeTree trees[2];
  uint64_t tree_clock = 0;
  DevBuf cur_a, cur_b, wide, nxt;                // interpolation workspaces (grow-only: no allocation on the per-proof path)
  DevBuf long_rows;                              // k_spmv -> k_spmv_long hand-off
  // factorial tables fact[i] = i!, invfact[i] = 1 / i! for i <= fact_top (Montgomery), grown on demand
  size_t fact_top = 0;
  DevBuf fact, invfact;
  // tables of the direct H route (hx_direct_dev), per (n, deg Z).  A few circuit sizes stay cached side by side (a Go key cache
  // with several keys, a batch over two circuits): with ONE entry alternating sizes rebuilt the tables for every proof and the
  // rebuild freed buffers that queued kernels of earlier tickets still read.
  struct HxTables {
    size_t n = 0, dz = 0, N = 0;
    uint64_t stamp = 0;
    DevBuf inv_spec, shift_spec, t1, t2;         // spectra of 1/(t+1) and (-n)^t/t! (size N), value scalings (n each)
  };
  static constexpr int kHxSlots = 4;
  HxTables hx[kHxSlots];
  uint64_t hx_clock = 0;
  // The workspaces of the H stage are shared by the slots (one set per slot kept ~1.9 GB per cached size at 2^22
  // constraints): 3 N, n, n elements and the violated-constraint word, grown to the largest size seen.  Every user enqueues on the
  // polynomial stream (aux 1), so stream order keeps consecutive proofs of different sizes apart.
  DevBuf hx_conv, hx_hv, hx_g, hx_bad;
};
static PolyState& poly_state(Ctx& c) { return c.state<PolyState>(c.poly_state); }

static Twiddles& ensure_twiddles(Ctx& c, int logn) {
  Twiddles& tw = poly_state(c).tw;
  if (logn <= tw.logn) return tw;
  if (logn > ModR::kTwoAdicity) throw HipError{hipErrorInvalidValue, "NTT size exceeds the 2-adicity of Fr (2^28)", __LINE__};
  logn = std::max(logn, 16);
  const uint32_t half = 1u << (logn - 1);
  // omega_{2^logn} = omega_{2^28}^(2^(28-logn))
  Fe<ModR, 2> w, wi;
  for (int i = 0; i < NL; ++i) { w.l[i] = ModR::omega28_mont(i); wi.l[i] = ModR::omega28_inv_mont(i); }
  for (int i = 0; i < ModR::kTwoAdicity - logn; ++i) { w = sqr(w); wi = sqr(wi); }
  tw.fwd.alloc((size_t)half * kTwWords * 4);
  tw.inv.alloc((size_t)half * kTwWords * 4);
  hipLaunchKernelGGL(k_twiddle_gen, grid1(half), dim3(256), 0, c.stream, tw.fwd.as<uint32_t>(), half, to_const(w));
  hipLaunchKernelGGL(k_twiddle_gen, grid1(half), dim3(256), 0, c.stream, tw.inv.as<uint32_t>(), half, to_const(wi));
  GS_HIP(hipGetLastError());
  tw.logn = logn;
  return tw;
}

void ntt_forward(Ctx& c, uint32_t* data, int logt, int logm) { ntt_forward_n(c, data, (size_t)1 << logt, logm); }
void ntt_inverse_unscaled(Ctx& c, uint32_t* data, int logt, int logm) { ntt_inverse_unscaled_n(c, data, (size_t)1 << logt, logm); }

// pass schedule: the lowest min(7, logm) stages form the contiguous s_lo = 0 pass, the stages above are cut into
// passes of <= 7 from the top; every upper pass has s_lo >= 7, so its tiles can take C = 8..128 consecutive columns.
struct NttPass { int s_lo, k, clog; uint32_t ntiles; };
static std::vector<NttPass> ntt_schedule(size_t total, int logm) {
  std::vector<NttPass> v;                      // in DIF (forward) order: top stages first
  const int low = std::min(kNttMaxStages, logm);
  int hi = logm;
  while (hi > low) {
    const int k = std::min(kNttMaxStages, hi - low);
    const int s_lo = hi - k;
    const int clog = std::min(kNttTileLog - k, s_lo);
    v.push_back(NttPass{s_lo, k, clog, (uint32_t)(total >> (k + clog))});
    hi = s_lo;
  }
  int clog = kNttTileLog - low;                // s_lo = 0 pass: C adjacent contiguous groups of 2^low
  while (clog > 0 && ((total >> low) & (((size_t)1 << clog) - 1))) --clog;
  v.push_back(NttPass{0, low, clog, (uint32_t)(total >> (low + clog))});
  return v;
}

void ntt_forward_n(Ctx& c, uint32_t* data, size_t total, int logm) {
  if (logm == 0 || total == 0) return;
  Twiddles& tw = ensure_twiddles(c, logm);
  for (const NttPass& p : ntt_schedule(total, logm))
    hipLaunchKernelGGL((k_ntt_pass<false, kNttLoadPlain>), dim3(p.ntiles), dim3(256), 0, c.stream, data, tw.fwd.as<uint32_t>(), tw.logn, p.s_lo, p.k, p.clog, NttLoadAux{});
  GS_HIP(hipGetLastError());
}

void ntt_inverse_unscaled_n(Ctx& c, uint32_t* data, size_t total, int logm) {
  if (logm == 0 || total == 0) return;
  Twiddles& tw = ensure_twiddles(c, logm);
  const std::vector<NttPass> sched = ntt_schedule(total, logm);
  for (auto it = sched.rbegin(); it != sched.rend(); ++it)
    hipLaunchKernelGGL((k_ntt_pass<true, kNttLoadPlain>), dim3(it->ntiles), dim3(256), 0, c.stream, data, tw.inv.as<uint32_t>(), tw.logn, it->s_lo, it->k, it->clog, NttLoadAux{});
  GS_HIP(hipGetLastError());
}

// inverse transform of data[i] * spec[i & (2^logm - 1)] (un-scaled): the point-wise product of a convolution inside the first inverse pass
static void ntt_inverse_unscaled_of_product_n(Ctx& c, uint32_t* data, size_t total, int logm, const uint32_t* spec) {
  if (logm == 0 || total == 0) return;
  Twiddles& tw = ensure_twiddles(c, logm);
  const std::vector<NttPass> sched = ntt_schedule(total, logm);
  bool first = true;
  for (auto it = sched.rbegin(); it != sched.rend(); ++it) {
    if (first) hipLaunchKernelGGL((k_ntt_pass<true, kNttLoadScaled>), dim3(it->ntiles), dim3(256), 0, c.stream, data, tw.inv.as<uint32_t>(), tw.logn, it->s_lo, it->k, it->clog,
                                  NttLoadAux{spec, nullptr, 0u, (uint32_t)logm});
    else hipLaunchKernelGGL((k_ntt_pass<true, kNttLoadPlain>), dim3(it->ntiles), dim3(256), 0, c.stream, data, tw.inv.as<uint32_t>(), tw.logn, it->s_lo, it->k, it->clog, NttLoadAux{});
    first = false;
  }
  GS_HIP(hipGetLastError());
}

// The batched node-extension convolution of the H-values stage with its two point-wise steps inside the transforms (round 5):
//   conv[v] = IFFT( FFT(pad(vals[v] . weights)) . spec ),  v < nvec, transforms of size N = 2^logN, un-scaled inverse.
// The first forward pass builds its input from (vals, weights) -- the zero-padded half is never written or read --, the first inverse
// pass multiplies by the kernel's spectrum while it loads.
static void weighed_convolution_dev(Ctx& c, const uint32_t* vals, const uint32_t* weights, size_t n, uint32_t nvec, const uint32_t* spec, int logN,
                                    uint32_t* conv) {
  Twiddles& tw = ensure_twiddles(c, logN);
  const size_t total = (size_t)nvec << logN;
  const std::vector<NttPass> sched = ntt_schedule(total, logN);
  bool first = true;
  for (const NttPass& p : sched) {
    if (first) hipLaunchKernelGGL((k_ntt_pass<false, kNttLoadWeighed>), dim3(p.ntiles), dim3(256), 0, c.stream, conv, tw.fwd.as<uint32_t>(), tw.logn, p.s_lo, p.k, p.clog,
                                  NttLoadAux{vals, weights, (uint32_t)n, (uint32_t)logN});
    else hipLaunchKernelGGL((k_ntt_pass<false, kNttLoadPlain>), dim3(p.ntiles), dim3(256), 0, c.stream, conv, tw.fwd.as<uint32_t>(), tw.logn, p.s_lo, p.k, p.clog, NttLoadAux{});
    first = false;
  }
  first = true;
  for (auto it = sched.rbegin(); it != sched.rend(); ++it) {
    if (first) hipLaunchKernelGGL((k_ntt_pass<true, kNttLoadScaled>), dim3(it->ntiles), dim3(256), 0, c.stream, conv, tw.inv.as<uint32_t>(), tw.logn, it->s_lo, it->k, it->clog,
                                  NttLoadAux{spec, nullptr, 0u, (uint32_t)logN});
    else hipLaunchKernelGGL((k_ntt_pass<true, kNttLoadPlain>), dim3(it->ntiles), dim3(256), 0, c.stream, conv, tw.inv.as<uint32_t>(), tw.logn, it->s_lo, it->k, it->clog, NttLoadAux{});
    first = false;
  }
  GS_HIP(hipGetLastError());
}

// scale constant s such that mont_mul(y, s) = y * 2^-logn * R^extra   (extra = 0 or 1)
static FrConst inv_n_const(int logn, int extra_r) {
  Fe<ModR, 2> v = inv(fr_small_mont(1ull << logn));          // (1/N) R
  Fe<ModR, 1> r2;
  for (int i = 0; i < NL; ++i) r2.l[i] = ModR::r2(i);
  for (int i = 0; i < extra_r; ++i) v = mul(v, r2);          // * R
  return to_const(v);
}

static void copy_padded(Ctx& c, const uint32_t* src, size_t n, uint32_t* dst, size_t total) {
  GS_HIP(hipMemcpyAsync(dst, src, n * 32, hipMemcpyDeviceToDevice, c.stream));
  if (total > n) GS_HIP(hipMemsetAsync(dst + n * 8, 0, (total - n) * 32, c.stream));
}


void poly_mul_dev(Ctx& c, const uint32_t* a, size_t na, Form fa, const uint32_t* b, size_t nb, Form fb, uint32_t* out) {
  if (na == 0 || nb == 0) return;
  const size_t nr = na + nb - 1;
  const int logn = ceil_log2(nr);
  const size_t N = (size_t)1 << logn;
  // mont*mont -> mont needs 1/N ; std*mont -> std needs 1/N ; std*std -> (ab/R) needs R/N
  const int extra = (fa == Form::Std && fb == Form::Std) ? 1 : 0;
  poly_state(c).ws_a.ensure(N * 32);
  poly_state(c).ws_b.ensure(N * 32);
  uint32_t* A = poly_state(c).ws_a.as<uint32_t>();
  uint32_t* B = poly_state(c).ws_b.as<uint32_t>();
  copy_padded(c, a, na, A, N);
  copy_padded(c, b, nb, B, N);
  ntt_forward(c, A, logn, logn);
  ntt_forward(c, B, logn, logn);
  hipLaunchKernelGGL(k_pw_mul, grid1(N), dim3(256), 0, c.stream, A, B, A, (uint32_t)N);
  ntt_inverse_unscaled(c, A, logn, logn);
  hipLaunchKernelGGL(k_pw_mul_const, grid1(nr), dim3(256), 0, c.stream, A, inv_n_const(logn, extra), out, (uint32_t)nr);
  GS_HIP(hipGetLastError());
}

void poly_inv_series_dev(Ctx& c, const uint32_t* f, size_t nf, size_t k, uint32_t* g) {
  // Newton: g <- g (2 - f g) mod x^(2t), starting from g = 1/f0 mod x
  Fe<ModR, 6> f0v;
  uint32_t f0w[8];
  GS_HIP(hipMemcpyAsync(f0w, f, 32, hipMemcpyDeviceToHost, c.stream));
  GS_HIP(hipStreamSynchronize(c.stream));
  f0v = unpack32<ModR>(f0w);
  if (is_zero(f0v)) throw HipError{hipErrorInvalidValue, "series inverse of a polynomial with zero constant term", __LINE__};
  Fe<ModR, 2> g0 = inv(reduce2(f0v));                  // Montgomery in, Montgomery out
  uint32_t g0w[8];
  {
    Fe<ModR, 1> cg = canon(g0);
    pack32<ModR>(cg, g0w);
  }
  GS_HIP(hipMemcpyAsync(g, g0w, 32, hipMemcpyHostToDevice, c.stream));
  GS_HIP(hipStreamSynchronize(c.stream));
  const FrConst two = to_const(fr_small_mont(2));
  size_t t = 1;
  while (t < k) {
    const size_t t2 = std::min(2 * t, k);
    const size_t fl = std::min(nf, t2);
    // e = f[:t2] * g[:t] (length fl + t - 1), keep the first t2 coefficients
    poly_state(c).ws_c.ensure((std::max(fl, t2) + t) * 32);
    poly_state(c).ws_d.ensure((t2 + t) * 32);
    uint32_t* E = poly_state(c).ws_c.as<uint32_t>();
    uint32_t* U = poly_state(c).ws_d.as<uint32_t>();
    poly_mul_dev(c, f, fl, Form::Mont, g, t, Form::Mont, E);
    const size_t el = std::min(fl + t - 1, t2);
    hipLaunchKernelGGL(k_two_minus, grid1(t2), dim3(256), 0, c.stream, E, two, U, (uint32_t)el, (uint32_t)t2);
    // g_new = g * u mod x^t2
    poly_mul_dev(c, g, t, Form::Mont, U, t2, Form::Mont, E);
    GS_HIP(hipMemcpyAsync(g, E, t2 * 32, hipMemcpyDeviceToDevice, c.stream));
    t = t2;
  }
  GS_HIP(hipGetLastError());
}

void divisor_init(Ctx& c, Divisor& d, const uint32_t* b_std_dev, size_t nb) {
  d.nb = nb;
  d.b_std.alloc(nb * 32);
  GS_HIP(hipMemcpyAsync(d.b_std.p, b_std_dev, nb * 32, hipMemcpyDeviceToDevice, c.stream));
  d.k = 0;
  d.logn_spec = 0;
  d.k_spec = 0;
}

static void divisor_ensure(Ctx& c, Divisor& d, size_t k) {
  if (k <= d.k) return;
  // f = rev(b) in Montgomery form, only the first min(nb, k) coefficients matter
  const size_t fl = std::min(d.nb, k);
  DevBuf f(fl * 32);
  hipLaunchKernelGGL(k_copy_reversed, grid1(fl), dim3(256), 0, c.stream, d.b_std.as<uint32_t>(), (uint32_t)(d.nb - fl), (uint32_t)fl,
                     f.as<uint32_t>(), (uint32_t)fl);
  poly_canon_dev(c, f.as<uint32_t>(), fl, 1);
  d.inv_rev_mont.alloc(k * 32);
  poly_inv_series_dev(c, f.as<uint32_t>(), fl, k, d.inv_rev_mont.as<uint32_t>());
  GS_HIP(hipStreamSynchronize(c.stream));
  d.k = k;
  d.logn_spec = 0;
}

void poly_quotient_dev(Ctx& c, Divisor& d, const uint32_t* a, size_t na, uint32_t* quo) {
  if (na < d.nb) return;
  const size_t k = na - d.nb + 1;
  divisor_ensure(c, d, k);
  const int logn = ceil_log2(2 * k - 1);
  const size_t N = (size_t)1 << logn;
  if (d.logn_spec != logn || d.k_spec != k) {
    d.inv_spec.alloc(N * 32);
    copy_padded(c, d.inv_rev_mont.as<uint32_t>(), k, d.inv_spec.as<uint32_t>(), N);
    ntt_forward(c, d.inv_spec.as<uint32_t>(), logn, logn);
    d.logn_spec = logn;
    d.k_spec = k;
  }
  poly_state(c).ws_a.ensure(N * 32);
  uint32_t* A = poly_state(c).ws_a.as<uint32_t>();
  // A = rev(a)[:k] = a[na-1], a[na-2], ..., a[na-k]  zero padded to N   (standard form)
  hipLaunchKernelGGL(k_copy_reversed, grid1(N), dim3(256), 0, c.stream, a, (uint32_t)(na - k), (uint32_t)k, A, (uint32_t)N);
  ntt_forward(c, A, logn, logn);
  if (logn >= 1) ntt_inverse_unscaled_of_product_n(c, A, N, logn, d.inv_spec.as<uint32_t>());      // (r5: the spectrum product inside the first inverse pass)
  else hipLaunchKernelGGL(k_pw_mul, grid1(N), dim3(256), 0, c.stream, A, d.inv_spec.as<uint32_t>(), A, (uint32_t)N);
  // scale by 1/N and un-reverse the first k coefficients: quo[i] = A[k-1-i] / N
  poly_state(c).ws_b.ensure(k * 32);
  hipLaunchKernelGGL(k_copy_reversed, grid1(k), dim3(256), 0, c.stream, A, 0u, (uint32_t)k, poly_state(c).ws_b.as<uint32_t>(), (uint32_t)k);
  hipLaunchKernelGGL(k_pw_mul_const, grid1(k), dim3(256), 0, c.stream, poly_state(c).ws_b.as<uint32_t>(), inv_n_const(logn, 0), quo, (uint32_t)k);
  GS_HIP(hipGetLastError());
}

// ---- subproduct tree over the nodes 1..n (r1csqap.go's interpolation nodes) ---------------------------------------
// Level j holds 2^(L-j) monic polynomials of degree d = 2^j (low d coefficients, leading 1 implicit); the leaves are
// (x - i) for i = 1..n and the factor x for the padding up to 2^L.  Kept per n (two most recent): the spectra of every
// level (NTT of the zero-padded blocks, 2^(L+1) elements per level -- 1.3 GB at n = 2^20, which is what the HBM is for),
// the root, and the barycentric weights 1 / M'(j).

// fact / invfact up to `top` (>= 1): two tiled prefix-product scans and ONE field inversion (of top!)
static void prefix_products(Ctx& c, int mode, uint32_t top, uint32_t count, uint32_t* out, Fe<ModR, 2>* last_host) {
  const uint32_t ntiles = (count + kPpTile - 1) / kPpTile;
  DevBuf tiles((size_t)ntiles * 32);
  hipLaunchKernelGGL(k_pp_tiles, dim3(ntiles), dim3(kPpBlock), 0, c.stream, mode, top, count, out, tiles.as<uint32_t>());
  if (ntiles > 1) {
    hipLaunchKernelGGL(k_pp_tile_scan, dim3(1), dim3(1024), 0, c.stream, tiles.as<uint32_t>(), ntiles);
    hipLaunchKernelGGL(k_pp_apply, grid1(count), dim3(256), 0, c.stream, out, tiles.as<uint32_t>(), count);
  }
  GS_HIP(hipGetLastError());
  if (last_host) {
    uint32_t w[8];
    GS_HIP(hipMemcpyAsync(w, out + (size_t)(count - 1) * 8, 32, hipMemcpyDeviceToHost, c.stream));
    GS_HIP(hipStreamSynchronize(c.stream));
    *last_host = reduce2(unpack32<ModR>(w));         // stored Montgomery value (< 2r), as is
  }
  GS_HIP(hipStreamSynchronize(c.stream));            // `tiles` is released here
}
static void ensure_factorials(Ctx& c, size_t top) {
  PolyState& ps = poly_state(c);
  if (top <= ps.fact_top) return;
  top = std::max<size_t>(top, 2 * ps.fact_top);
  if (top >= (1ull << 31)) throw HipError{hipErrorInvalidValue, "factorial table too large", __LINE__};
  DevBuf fwd(top * 32), rev(top * 32);
  Fe<ModR, 2> top_fact;
  prefix_products(c, 0, (uint32_t)top, (uint32_t)top, fwd.as<uint32_t>(), &top_fact);       // fwd[i] = (i + 1)!
  prefix_products(c, 1, (uint32_t)top, (uint32_t)top, rev.as<uint32_t>(), nullptr);         // rev[t] = top! / (top - t - 1)!
  ps.fact.alloc((top + 1) * 32);
  ps.invfact.alloc((top + 1) * 32);
  hipLaunchKernelGGL(k_fact_tables, grid1(top + 1), dim3(256), 0, c.stream, fwd.as<uint32_t>(), rev.as<uint32_t>(), to_const(inv(top_fact)), (uint32_t)top,
                     ps.fact.as<uint32_t>(), ps.invfact.as<uint32_t>());
  GS_HIP(hipGetLastError());
  GS_HIP(hipStreamSynchronize(c.stream));
  ps.fact_top = top;
}

static void build_weights(Ctx& c, NodeTree& t) {
  // 1 / M'(j) = (-1)^(n-j) / ((j-1)! (n-j)!)        [cf. NewPolZeroAt's divisor, r1csqap.go:130-136, without its int overflow]
  ensure_factorials(c, std::max<size_t>(t.n, 1));
  t.weights.alloc(std::max<size_t>(t.n, 1) * 32);
  if (t.n) hipLaunchKernelGGL(k_bary_weights, grid1(t.n), dim3(256), 0, c.stream, poly_state(c).invfact.as<uint32_t>(), (uint32_t)t.n, t.weights.as<uint32_t>());
  GS_HIP(hipGetLastError());
  GS_HIP(hipStreamSynchronize(c.stream));
}

static NodeTree& ensure_tree(Ctx& c, size_t n, bool need_weights) {
  NodeTree* t = nullptr;
  PolyState& ps = poly_state(c);
  for (auto& x : ps.trees) if (x.n == n && x.root.p) t = &x;
  if (!t) {
    t = (ps.trees[0].stamp <= ps.trees[1].stamp) ? &ps.trees[0] : &ps.trees[1];
    *t = NodeTree{};
    t->n = n;
    t->L = ceil_log2(std::max<size_t>(n, 1));
    t->total = (size_t)1 << t->L;
    t->pad = t->total - n;
    const size_t total = t->total;
    DevBuf cur(total * 32), nxt(total * 32);
    hipLaunchKernelGGL(k_zp_leaves, grid1(total), dim3(256), 0, c.stream, cur.as<uint32_t>(), (uint32_t)n, (uint32_t)total);
    t->mspec.resize(t->L);
    for (int j = 0; j < t->L; ++j) {                           // blocks of d = 2^j low coefficients -> blocks of 2d
      const uint32_t d = 1u << j;
      DevBuf& wide = t->mspec[j];
      wide.alloc(2 * total * 32);
      hipLaunchKernelGGL(k_expand_blocks, grid1(2 * total), dim3(256), 0, c.stream, cur.as<uint32_t>(), wide.as<uint32_t>(), d, (uint32_t)(2 * total));
      ntt_forward(c, wide.as<uint32_t>(), t->L + 1, j + 1);
      hipLaunchKernelGGL(k_pw_mul_pairs, grid1(total), dim3(256), 0, c.stream, wide.as<uint32_t>(), nxt.as<uint32_t>(), 2 * d, (uint32_t)total);
      ntt_inverse_unscaled(c, nxt.as<uint32_t>(), t->L, j + 1);
      // (x^d + a)(x^d + b): low 2d coefficients = a b + x^d (a + b)
      DevBuf out(total * 32);
      hipLaunchKernelGGL(k_monic_combine, grid1(total), dim3(256), 0, c.stream, nxt.as<uint32_t>(), cur.as<uint32_t>(), inv_n_const(j + 1, 0),
                         out.as<uint32_t>(), d, (uint32_t)total);
      GS_HIP(hipStreamSynchronize(c.stream));
      cur = std::move(out);
    }
    GS_HIP(hipGetLastError());
    GS_HIP(hipStreamSynchronize(c.stream));
    t->root = std::move(cur);
  }
  if (need_weights && !t->weights.p) build_weights(c, *t);
  t->stamp = ++ps.tree_clock;
  return *t;
}

// Z(x) = prod_{i=1}^{deg} (x - i): deg + 1 coefficients, canonical standard form.
void zpoly_dev(Ctx& c, size_t deg, uint32_t* out_std) {
  const uint32_t one[8] = {1, 0, 0, 0, 0, 0, 0, 0};
  if (deg > 0) {
    NodeTree& t = ensure_tree(c, deg, false);
    // root = low 2^L coefficients of x^pad * Z (Montgomery); Z[k] = root[k + pad] for k < deg, Z[deg] = 1
    GS_HIP(hipMemcpyAsync(out_std, t.root.as<uint32_t>() + t.pad * 8, deg * 32, hipMemcpyDeviceToDevice, c.stream));
    poly_canon_dev(c, out_std, deg, 2);
  }
  GS_HIP(hipMemcpyAsync(out_std + deg * 8, one, 32, hipMemcpyHostToDevice, c.stream));
  GS_HIP(hipStreamSynchronize(c.stream));
}

// Lagrange interpolation on the nodes 1..n of `nvec` value vectors at once (values: nvec x n, standard form) ->
// coefficients (nvec x n, standard form, < 2r).  p(x) = sum_j v_j / M'(j) * M(x) / (x - j), evaluated bottom-up on the
// node tree:  P_node = P_left * M_right + P_right * M_left   (r1csqap.go:150-158 computes the same polynomial in O(n^3)).
void interpolate_dev(Ctx& c, const uint32_t* values_std, size_t n, size_t nvec, uint32_t* coeffs_std) {
  if (n == 0 || nvec == 0) return;
  NodeTree& t = ensure_tree(c, n, true);
  const size_t total = t.total, all = total * nvec;
  PolyState& ps = poly_state(c);
  DevBuf &cur_a = ps.cur_a, &cur_b = ps.cur_b, &wide = ps.wide, &nxt = ps.nxt;
  cur_a.ensure(all * 32); cur_b.ensure(all * 32); wide.ensure(2 * all * 32); nxt.ensure(all * 32);
  uint32_t* cur = cur_a.as<uint32_t>();
  uint32_t* other = cur_b.as<uint32_t>();
  hipLaunchKernelGGL(k_interp_leaves, grid1(all), dim3(256), 0, c.stream, values_std, t.weights.as<uint32_t>(), (uint32_t)n, (uint32_t)total,
                     (uint32_t)nvec, cur, wide.as<uint32_t>());
  for (int j = 0; j < t.L; ++j) {
    const uint32_t d = 1u << j;
    // `wide` already holds this level's blocks zero-padded to 2d (written by the previous combine / the leaves kernel)
    ntt_forward_n(c, wide.as<uint32_t>(), 2 * all, j + 1);
    hipLaunchKernelGGL(k_pw_cross, grid1(all), dim3(256), 0, c.stream, wide.as<uint32_t>(), t.mspec[j].as<uint32_t>(), nxt.as<uint32_t>(), 2 * d,
                       (uint32_t)(total / (2 * d)), (uint32_t)all);
    ntt_inverse_unscaled_n(c, nxt.as<uint32_t>(), all, j + 1);
    // P_L (x^d + m_R) + P_R (x^d + m_L) = [P_L m_R + P_R m_L] + x^d (P_L + P_R): compact into `other`, padded into `wide`
    hipLaunchKernelGGL(k_interp_combine, grid1(all), dim3(256), 0, c.stream, nxt.as<uint32_t>(), cur, inv_n_const(j + 1, 0), other,
                       (j + 1 < t.L) ? wide.as<uint32_t>() : nullptr, d, (uint32_t)all);
    std::swap(cur, other);
  }
  GS_HIP(hipGetLastError());
  // cur[k] = x^pad * p_k(x): coefficients pad .. pad + n - 1
  for (size_t k = 0; k < nvec; ++k)
    GS_HIP(hipMemcpyAsync(coeffs_std + k * n * 8, cur + (k * total + t.pad) * 8, n * 32, hipMemcpyDeviceToDevice, c.stream));
  // (no host wait: every workspace above is persistent, and callers that read the result on the host synchronise themselves)
}

// ---- H = (A B - C) / Z from the constraint values -----------------------------------------------------------------------
// vals: [A w | B w | C w], n each, standard form.  For a witness that satisfies the constraints at every root of Z
// (deg Z = dz in {n - 1, n}) P = A B - C is an exact multiple of Z and H is fixed by its values at the n nodes n+1..2n.
// See poly_kernels.h for the derivation.
static bool hx_shape_ok(size_t n, size_t dz) { return n >= 2 && (dz == n - 1 || dz == n); }

// tables of one (n, deg Z): once (synchronises the stream when it builds)
static PolyState::HxTables& hx_tables(Ctx& c, size_t n, size_t dz) {
  PolyState& ps = poly_state(c);
  PolyState::HxTables* slot = &ps.hx[0];
  for (auto& e : ps.hx) {
    if (e.n == n && e.dz == dz) { e.stamp = ++ps.hx_clock; return e; }
    if (e.n == 0 ? slot->n != 0 : (slot->n != 0 && e.stamp < slot->stamp)) slot = &e;      // an empty slot, else the least recently used
  }
  PolyState::HxTables& hx = *slot;
  // Every user of these tables enqueues on the polynomial stream (c.stream here): before a victim's buffers go, whatever is queued
  // on it must have run -- stated here instead of leaning on hipFree's device-wide synchronisation.
  if (hx.n != 0) {
    GS_HIP(hipStreamSynchronize(c.stream));
    if (c.aux_stream[1] && c.aux_stream[1] != c.stream) GS_HIP(hipStreamSynchronize(c.aux_stream[1]));   // (whoever the caller is)
  }
  const int logN = ceil_log2(2 * n);
  const size_t N = (size_t)1 << logN;
  Fe<ModR, 1> r2;
  for (int i = 0; i < NL; ++i) r2.l[i] = ModR::r2(i);
  const FrConst r2c = to_const(relax<2>(r2));
  const FrConst inv_N = inv_n_const(logN, 0);
  ensure_tree(c, n, true);
  ensure_factorials(c, 2 * n);
  hx = PolyState::HxTables{};
  hx.inv_spec.alloc(N * 32); hx.shift_spec.alloc(N * 32); hx.t1.alloc(n * 32); hx.t2.alloc(n * 32);
  if (ps.hx_conv.p && (3 * N * 32 > ps.hx_conv.bytes || n * 32 > ps.hx_hv.bytes)) {      // growing: earlier proofs' H stages may still use the old ones
    GS_HIP(hipStreamSynchronize(c.stream));
    if (c.aux_stream[1] && c.aux_stream[1] != c.stream) GS_HIP(hipStreamSynchronize(c.aux_stream[1]));
  }
  ps.hx_conv.ensure(3 * N * 32); ps.hx_hv.ensure(n * 32); ps.hx_g.ensure(n * 32); ps.hx_bad.ensure(16);
  hipLaunchKernelGGL(k_hx_tables, grid1(N), dim3(256), 0, c.stream, ps.fact.as<uint32_t>(), ps.invfact.as<uint32_t>(), (uint32_t)n, (uint32_t)dz, (uint32_t)N,
                     inv_N, r2c, hx.inv_spec.as<uint32_t>(), hx.t1.as<uint32_t>(), hx.t2.as<uint32_t>());
  // (-n)^t, t < n, in standard form (scale 1), then / t!
  DevBuf pw(n * 32);
  uint64_t negn[4], one[4] = {1, 0, 0, 0};
  {
    uint64_t nn[4] = {(uint64_t)n, 0, 0, 0};
    fr_words_from_mont(reduce2(neg(fr_from_words_mont(nn))), negn);
  }
  scaled_powers_dev(c, negn, one, n, pw.as<uint32_t>());
  hipLaunchKernelGGL(k_hx_shift_table, grid1(N), dim3(256), 0, c.stream, ps.invfact.as<uint32_t>(), pw.as<uint32_t>(), (uint32_t)n, (uint32_t)N,
                     hx.shift_spec.as<uint32_t>());
  ntt_forward(c, hx.inv_spec.as<uint32_t>(), logN, logN);
  ntt_forward(c, hx.shift_spec.as<uint32_t>(), logN, logN);
  GS_HIP(hipGetLastError());
  GS_HIP(hipStreamSynchronize(c.stream));              // `pw` is released here
  hx.n = n; hx.dz = dz; hx.N = N;
  hx.stamp = ++ps.hx_clock;
  return hx;
}

// *bad_dev (one device word) = number of roots j of Z at which a_j b_j != c_j.  Enqueue only.
void r1cs_check_dev(Ctx& c, const uint32_t* vals_std, size_t n, size_t dz, uint32_t* bad_dev) {
  Fe<ModR, 1> r2;
  for (int i = 0; i < NL; ++i) r2.l[i] = ModR::r2(i);
  GS_HIP(hipMemsetAsync(bad_dev, 0, 4, c.stream));
  if (n) hipLaunchKernelGGL(k_r1cs_check, grid1(n), dim3(256), 0, c.stream, vals_std, (uint32_t)n, (uint32_t)dz, to_const(relax<2>(r2)), bad_dev);
  GS_HIP(hipGetLastError());
}

// hv[k - 1] = H(n + k), k = 1..n (canonical standard form): the values of A, B, C at the nodes n+1..2n by three cyclic
// convolutions with 1 / (t + 1) (batched), then (a b - c) / Z point-wise.  No host wait once the tables of (n, dz) exist.
bool hx_values_dev(Ctx& c, const uint32_t* vals_std, size_t n, size_t dz, uint32_t* hv_out) {
  if (!hx_shape_ok(n, dz)) return false;
  PolyState::HxTables& hx = hx_tables(c, n, dz);
  NodeTree& t = ensure_tree(c, n, true);
  const int logN = ceil_log2(2 * n);
  const size_t N = hx.N;
  uint32_t* conv = poly_state(c).hx_conv.as<uint32_t>();
  static const bool unfused = dev_flag("GS_HX_UNFUSED");          // development builds: the separate point-wise kernels of rounds 3-4, for A/B
  if (unfused) {
    hipLaunchKernelGGL(k_hx_weigh, grid1(3 * N), dim3(256), 0, c.stream, vals_std, t.weights.as<uint32_t>(), (uint32_t)n, (uint32_t)N, 3u, conv);
    ntt_forward_n(c, conv, 3 * N, logN);
    hipLaunchKernelGGL(k_pw_mul_bcast, grid1(3 * N), dim3(256), 0, c.stream, conv, hx.inv_spec.as<uint32_t>(), (uint32_t)N, 3u);
    ntt_inverse_unscaled_n(c, conv, 3 * N, logN);
  } else {
    weighed_convolution_dev(c, vals_std, t.weights.as<uint32_t>(), n, 3u, hx.inv_spec.as<uint32_t>(), logN, conv);
  }
  hipLaunchKernelGGL(k_hx_values, grid1(n), dim3(256), 0, c.stream, conv, hx.t1.as<uint32_t>(), hx.t2.as<uint32_t>(), (uint32_t)n, (uint32_t)N, hv_out);
  GS_HIP(hipGetLastError());
  return true;
}

// H's nh = 2n - 1 - dz coefficients (canonical standard form) from those values: G(y) = H(y + n) by ONE tree interpolation at
// y = 1..n, then the Taylor shift H(x) = G(x - n) (one more convolution).
void hx_from_values_dev(Ctx& c, const uint32_t* hv_std, size_t n, size_t dz, uint32_t* hx_out) {
  PolyState& ps = poly_state(c);
  PolyState::HxTables& hx = hx_tables(c, n, dz);
  const int logN = ceil_log2(2 * n);
  const size_t N = hx.N, nh = 2 * n - 1 - dz;
  const FrConst inv_N = inv_n_const(logN, 0);
  uint32_t* conv = poly_state(c).hx_conv.as<uint32_t>();
  interpolate_dev(c, hv_std, n, 1, poly_state(c).hx_g.as<uint32_t>());
  hipLaunchKernelGGL(k_hx_shift_in, grid1(N), dim3(256), 0, c.stream, poly_state(c).hx_g.as<uint32_t>(), ps.fact.as<uint32_t>(), (uint32_t)n, (uint32_t)N, conv);
  ntt_forward(c, conv, logN, logN);
  hipLaunchKernelGGL(k_pw_mul, grid1(N), dim3(256), 0, c.stream, conv, hx.shift_spec.as<uint32_t>(), conv, (uint32_t)N);
  ntt_inverse_unscaled(c, conv, logN, logN);
  hipLaunchKernelGGL(k_hx_shift_out, grid1(nh), dim3(256), 0, c.stream, conv, ps.invfact.as<uint32_t>(), inv_N, (uint32_t)n, (uint32_t)nh, hx_out);
  GS_HIP(hipGetLastError());
}

// Both steps, for keys that only have the monomial PowersTauDelta / G1T.  Returns false -- and writes nothing -- when a constraint
// is violated at a root of Z: the caller then takes the exact route (px, then the floor quotient the reference computes,
// r1csqap.go:70-84).  hx_out: nh = 2n - 1 - dz coefficients, canonical standard form.
bool hx_direct_dev(Ctx& c, const uint32_t* vals_std, size_t n, size_t dz, uint32_t* hx_out) {
  if (!hx_shape_ok(n, dz)) return false;
  (void)hx_tables(c, n, dz);                            // makes the cached spectra and the shared workspaces (hx_bad, hx_hv) exist
  uint32_t nbad = 0;                                    // the flag word lives with the tables: no allocation on the per-proof path
  r1cs_check_dev(c, vals_std, n, dz, poly_state(c).hx_bad.as<uint32_t>());
  GS_HIP(hipMemcpyAsync(&nbad, poly_state(c).hx_bad.p, 4, hipMemcpyDeviceToHost, c.stream));
  GS_HIP(hipStreamSynchronize(c.stream));
  if (nbad) return false;
  hx_values_dev(c, vals_std, n, dz, poly_state(c).hx_hv.as<uint32_t>());
  hx_from_values_dev(c, poly_state(c).hx_hv.as<uint32_t>(), n, dz, hx_out);
  return true;
}

const uint32_t* interpolation_weights_dev(Ctx& c, size_t n) { return ensure_tree(c, n, true).weights.as<uint32_t>(); }

static FrConst fr_const_from_words(const uint64_t w[4]) { return to_const(fr_from_words_mont(w)); }

void lagrange_at_dev(Ctx& c, size_t n, const uint64_t tau[4], const uint64_t mtau[4], uint32_t* out_mont) {
  const uint32_t* wts = interpolation_weights_dev(c, n);
  if (n) hipLaunchKernelGGL(k_lagrange_at, grid1(n), dim3(256), 0, c.stream, wts, (uint32_t)n, fr_const_from_words(tau), fr_const_from_words(mtau), out_mont);
  GS_HIP(hipGetLastError());
}

void setup_scalars_dev(Ctx& c, const uint32_t* at, const uint32_t* bt, const uint32_t* ct, size_t m, size_t npublic, const uint64_t kalpha[4],
                       const uint64_t kbeta[4], const uint64_t inv_delta[4], const uint64_t inv_gamma[4], uint32_t* cd, uint32_t* ic) {
  const uint64_t one[4] = {1, 0, 0, 0};
  if (m) hipLaunchKernelGGL(k_setup_scalars, grid1(m), dim3(256), 0, c.stream, at, bt, ct, (uint32_t)m, (uint32_t)npublic, fr_const_from_words(kalpha),
                            fr_const_from_words(kbeta), fr_const_from_words(inv_delta), fr_const_from_words(inv_gamma), fr_const_from_words(one), cd, ic);
  GS_HIP(hipGetLastError());
}

void pinocchio_scalars_dev(Ctx& c, const uint32_t* at, const uint32_t* bt, const uint32_t* ct, size_t m, const uint64_t rhoa[4], const uint64_t rhob[4],
                           const uint64_t rhoc[4], const uint64_t ka[4], const uint64_t kb[4], const uint64_t kc[4], const uint64_t kbeta[4],
                           uint32_t* const out[7]) {
  PinoConsts k{fr_const_from_words(rhoa), fr_const_from_words(rhob), fr_const_from_words(rhoc), fr_const_from_words(ka), fr_const_from_words(kb),
               fr_const_from_words(kc), fr_const_from_words(kbeta)};
  if (m) hipLaunchKernelGGL(k_pinocchio_scalars, grid1(m), dim3(256), 0, c.stream, at, bt, ct, (uint32_t)m, k, out[0], out[1], out[2], out[3], out[4], out[5], out[6]);
  GS_HIP(hipGetLastError());
}

// out[i] = a[i] * scale: a Montgomery, scale given in STANDARD words (used raw) -> canonical standard form
void scale_mont_by_std_dev(Ctx& c, const uint32_t* a_mont, const uint64_t scale_std[4], size_t n, uint32_t* out_std) {
  FrConst sc;
  uint32_t u[8];
  for (int i = 0; i < 4; ++i) { u[2 * i] = (uint32_t)scale_std[i]; u[2 * i + 1] = (uint32_t)(scale_std[i] >> 32); }
  const Fe<ModR, 6> raw = unpack32<ModR>(u);
  for (int i = 0; i < NL; ++i) sc.l[i] = raw.l[i];
  if (n) hipLaunchKernelGGL(k_pw_mul_const, grid1(n), dim3(256), 0, c.stream, a_mont, sc, out_std, (uint32_t)n);
  GS_HIP(hipGetLastError());
  poly_canon_dev(c, out_std, n, 0);
}

void scaled_powers_dev(Ctx& c, const uint64_t base[4], const uint64_t scale_std[4], size_t count, uint32_t* out_std) {
  // scale is passed in STANDARD limbs (not converted): acc starts as the raw value, Montgomery products by base keep it standard
  FrConst sc;
  uint32_t u[8];
  for (int i = 0; i < 4; ++i) { u[2 * i] = (uint32_t)scale_std[i]; u[2 * i + 1] = (uint32_t)(scale_std[i] >> 32); }
  const Fe<ModR, 6> raw = unpack32<ModR>(u);
  for (int i = 0; i < NL; ++i) sc.l[i] = raw.l[i];
  if (count) hipLaunchKernelGGL(k_scaled_powers, grid1(count), dim3(256), 0, c.stream, out_std, (uint32_t)count, fr_const_from_words(base), sc);
  GS_HIP(hipGetLastError());
}

// ---- host-side Fr arithmetic on ABI words (setup constants, toxic-value bookkeeping) ---------------------------------
void fr_words_from_mont(const Fe<ModR, 2>& a, uint64_t out[4]) {
  uint32_t w[8];
  pack32<ModR>(from_mont(a), w);
  for (int i = 0; i < 4; ++i) out[i] = (uint64_t)w[2 * i] | ((uint64_t)w[2 * i + 1] << 32);
}
void fr_mul_words(const uint64_t a[4], const uint64_t b[4], uint64_t out[4]) { fr_words_from_mont(mul(fr_from_words_mont(a), fr_from_words_mont(b)), out); }
void fr_inv_words(const uint64_t a[4], uint64_t out[4]) { fr_words_from_mont(inv(fr_from_words_mont(a)), out); }
void fr_sub_words(const uint64_t a[4], const uint64_t b[4], uint64_t out[4]) { fr_words_from_mont(reduce2(sub(fr_from_words_mont(a), fr_from_words_mont(b))), out); }
bool fr_is_zero_words(const uint64_t a[4]) { return is_zero(fr_from_words_mont(a)); }
// prod_{k=1}^{count} (x - k)
void fr_falling_product_words(const uint64_t x[4], size_t count, uint64_t out[4]) {
  const Fe<ModR, 2> xm = fr_from_words_mont(x);
  Fe<ModR, 2> acc = relax<2>(fe_one<ModR>());
  Fe<ModR, 2> k = fe_zero<ModR, 2>();
  const Fe<ModR, 1> one = fe_one<ModR>();
  for (size_t i = 1; i <= count; ++i) {
    k = reduce2(add(k, one));
    acc = mul(acc, sub(xm, k));
  }
  fr_words_from_mont(acc, out);
}

// out[row] = sum_k val[k] * x[col[k]] over Fr; CSR with values and x in standard form; out standard (< 2r)
void spmv_dev(Ctx& c, const uint32_t* rowptr, const uint32_t* col, const uint32_t* val_std, const uint32_t* x_mont, size_t nrows, size_t ncols,
              uint32_t* out_std) {
  if (nrows) {
    DevBuf& long_rows = poly_state(c).long_rows;   // [count | row indices]: filled by k_spmv, consumed by k_spmv_long (stream order)
    long_rows.ensure((1 + kSpmvLongCap) * 4);
    GS_HIP(hipMemsetAsync(long_rows.p, 0, 4, c.stream));
    hipLaunchKernelGGL(k_spmv, grid1(nrows), dim3(256), 0, c.stream, rowptr, col, val_std, x_mont, (uint32_t)nrows, (uint32_t)ncols, out_std,
                       long_rows.as<uint32_t>());
    hipLaunchKernelGGL(k_spmv_long, dim3(64), dim3(256), 0, c.stream, rowptr, col, val_std, x_mont, (uint32_t)ncols, out_std,
                       long_rows.as<uint32_t>());
  }
  GS_HIP(hipGetLastError());
}

void poly_addsub_dev(Ctx& c, const uint32_t* a, size_t na, const uint32_t* b, size_t nb, bool subtract, uint32_t* out) {
  const size_t n = std::max(na, nb);
  if (n) hipLaunchKernelGGL(k_addsub, grid1(n), dim3(256), 0, c.stream, a, (uint32_t)na, b, (uint32_t)nb, subtract ? 1 : 0, out, (uint32_t)n);
  GS_HIP(hipGetLastError());
}

void poly_canon_dev(Ctx& c, uint32_t* x, size_t n, int mode) {
  if (n) hipLaunchKernelGGL(k_convert, grid1(n), dim3(256), 0, c.stream, x, (uint32_t)n, mode);
  GS_HIP(hipGetLastError());
}

void poly_eval_dev(Ctx& c, const uint32_t* v, size_t n, const uint64_t x[4], uint32_t* out_dev) {
  const uint32_t nchunks = (uint32_t)((n + kEvalChunk - 1) / kEvalChunk);
  if (nchunks == 0) { GS_HIP(hipMemsetAsync(out_dev, 0, 32, c.stream)); return; }
  poly_state(c).ws_c.ensure((size_t)nchunks * 32);
  hipLaunchKernelGGL(k_eval_chunks, grid1(nchunks), dim3(256), 0, c.stream, v, (uint32_t)n, to_const(fr_from_words_mont(x)),
                     poly_state(c).ws_c.as<uint32_t>(), nchunks);
  hipLaunchKernelGGL(k_sum_block, dim3(1), dim3(256), 0, c.stream, poly_state(c).ws_c.as<uint32_t>(), nchunks, out_dev);
  GS_HIP(hipGetLastError());
}

}  // namespace gs
