// MSM engine: host orchestration of the kernels in msm_kernels.h + the O(1) serial combination.
#include "msm.h"
#include "hostcopy.h"

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <future>
#include <type_traits>

#include "msm_kernels.h"

namespace gs {

static_assert(sizeof(G1Xyzz) == PointIO<FqTag>::kXyzzWords * 4, "G1 XYZZ must be raw limbs");
static_assert(sizeof(G2Xyzz) == PointIO<Fq2Tag>::kXyzzWords * 4, "G2 XYZZ must be raw limbs");

static inline dim3 grid1(size_t n, int block = 256) { return dim3((unsigned)((n + block - 1) / block)); }

// Window width.  With window tables an MSM costs n * W(c) mixed additions (W = floor(254 / c) + 1 digit positions, all feeding
// ONE bucket set) plus two complete additions per bucket for the reduction (a complete XYZZ addition is ~1.4 mixed ones) plus
// the chunk partials, so wide windows pay as soon as n is large against the 2^(c-1) buckets: c = 20 at n = 2^20 is 13
// additions per term instead of 16.  GS_WINDOW_COST_BUCKET (in mixed additions per bucket) tunes the model for experiments.
int choose_window_bits(uint32_t n, int forced) {
  if (forced >= 8 && forced <= kMaxWindowBits) return forced;
  static const double per_bucket = dev_knob_f("GS_WINDOW_COST_BUCKET", 6.0, 0.0, 64.0);
  // The model stops at 17 bits: wider windows need 4+ bucket ranges in the sort and a deeper reduce, and measured end to end they
  // lose what the shorter accumulation wins (2^22-constraint proof: c = 17 37.8 ms, 18 41.3 ms, 20 41.6 ms; 2^20: 9.1 / 9.6 / 9.9).
  static const int auto_max = (int)dev_knob("GS_AUTO_MAX_C", 17, 8, kMaxWindowBits);
  int best = 8;
  double best_cost = 1e300;
  for (int c = 8; c <= std::min(kMaxWindowBits, std::max(8, auto_max)); ++c) {
    const int W = 254 / c + 1;
    const double cost = (double)n * W + per_bucket * (double)(1u << (c - 1));
    if (cost < best_cost) { best_cost = cost; best = c; }
  }
  return best;
}

// Table-free route: every window has its own bucket set, so a bucket costs per WINDOW: cost(c) = n W + k W 2^(c-1) with k ~ 3 mixed
// additions per bucket (one combine, two complete additions of the reduce).  2^20 terms: c = 16 (16.8 M + 1.6 M additions against the
// 15.7 M + 0.2 M of the 17-bit window tables: ~1.15x); 2^16 terms: c = 13.
int choose_window_bits_free(uint32_t n, int forced) {
  if (forced) return std::min(kMaxFreeWindowBits, std::max(kMinFreeWindowBits, forced));
  static const double per_bucket = dev_knob_f("GS_WINDOW_COST_BUCKET_FREE", 3.0, 0.0, 64.0);
  int best = kMinFreeWindowBits;
  double best_cost = 1e300;
  for (int c = kMinFreeWindowBits; c <= kMaxFreeWindowBits; ++c) {
    const int W = 254 / c + 1;
    const double cost = (double)W * ((double)n + per_bucket * (double)(1u << (c - 1)));
    if (cost < best_cost) { best_cost = cost; best = c; }
  }
  return best;
}

static void exclusive_scan(Ctx& c, PlanBuffers& pb, const uint32_t* in, uint32_t* out, uint32_t n) {
  const uint32_t ntiles = (n + kScanTile - 1) / kScanTile;
  pb.tiles.ensure((size_t)ntiles * 4);
  pb.total.ensure(4);
  hipLaunchKernelGGL(k_scan_tiles, dim3(ntiles), dim3(kScanBlock), 0, c.stream, in, out, pb.tiles.as<uint32_t>(), n);
  hipLaunchKernelGGL(k_scan_tile_sums, dim3(1), dim3(1024), 0, c.stream, pb.tiles.as<uint32_t>(), ntiles, pb.total.as<uint32_t>());
  hipLaunchKernelGGL(k_scan_add, dim3(ntiles), dim3(kScanBlock), 0, c.stream, out, pb.tiles.as<uint32_t>(), n);
}

struct MsmState {                                  // per context (device memory belongs to one device)
  PlanBuffers plan_slots[3 * Ctx::kSlots];         // (w, h) x slots, then one more per slot: the masked plan over w of keys with sparse B arrays
  DevBuf table_scratch;                            // slab of the batched window-table builder
  DevBuf table_scratch_bg;                         // ... and of the builds that run in the background on the table stream
  DevBuf upload_scratch;                           // Jacobian triples of a base array on their way in (upload_jacobian), kept up to kUploadScratchKeep
  DevBuf upload_flag;                              // {points off the curve, first such index} of the conversion kernel
  bool lds_attr_set = false;
};
static MsmState& msm_state(Ctx& c) { return c.state<MsmState>(c.msm_state); }

// Chunk size of a plan = additions one accumulate thread performs back to back.  Small chunks mean more threads and shorter
// serial chains (a 2^16-term MSM with 32-entry chunks is only 512 waves of 32 dependent additions), large chunks mean fewer
// bucket flushes and fewer chunk partials for the tails.  Measured (pipelined prove / single G1 MSM, ms):
//   2^16 terms: chunk 8 1.16 / 0.60, 16 1.17 / 0.59, 32 1.39 / 0.65, 64 1.92
//   2^18 terms: chunk 8 2.86, 16 2.85, 32 2.96, 64 3.48
//   2^20 terms: chunk 16 9.97 / 2.20, 32 9.94 / 2.01, 64 10.26 / 1.99, 128 11.0
//   2^22 terms: chunk 32 48.7, 64 40.7 -- a bucket should not be cut into more than ~32 chunks (its partials are added up
//   serially by one combine thread, and past kHeavySpan by the block-wide tree)
static uint32_t choose_chunk(uint64_t entries, uint32_t nbuckets, const std::vector<LaunchShape>& users) {
  static const int forced = (int)dev_knob("GS_CHUNK", 0, 4, 1024, 4);                     // development builds only (a multiple of 4)
  if (forced >= 4 && forced % 4 == 0) return (uint32_t)forced;
  // (development builds: another chunk size for plans whose only user is ONE G1 launch of one base array -- the sum over h)
  static const int forced_h = (int)dev_knob("GS_CHUNK_H", 0, 4, 1024, 4);
  if (forced_h && users.size() == 1 && users[0].njobs == 1 && !users[0].g2) return (uint32_t)forced_h;
  // (round 6: 64 from 2^27 entries on -- 2^24 constraints table-free.  Every chunk owns a head and a tail partial, 288 B for G1 and 576 B
  //  for G2: with 32-entry chunks the partials of ONE ticket slot were 17 GB at 2^24, half of what stood between the library and 2^25)
  uint32_t chunk = entries >= (1ull << 27) ? 64u : entries >= (1ull << 23) ? 32u : 16u;
  while (chunk < 1024u && entries / std::max<uint32_t>(nbuckets, 1u) > 32ull * chunk) chunk *= 2;
  return chunk;
}

// R >= 4 bucket ranges: partition by (window, range) first (msm_kernels.h); with two ranges reading the digit matrix twice is
// cheaper than writing and re-reading 8-byte records
static uint32_t part_min_ranges() {
  static const uint32_t v = (uint32_t)dev_knob("GS_PART_MIN_R", 4, 2, 64);
  return v;
}
bool plan_partitions_first(int cbits) {
  if (cbits < 1) return false;
  return std::max<uint32_t>(1u, (1u << (cbits - 1)) >> kRangeLog) >= part_min_ranges();
}

uint32_t finite_mask_dev(Ctx& c, const uint32_t* g1_pts, const uint32_t* g2_pts, uint32_t n, uint32_t* mask_dev) {
  const size_t words = ((size_t)n + 31) / 32;
  GS_HIP(hipMemsetAsync(mask_dev, 0, std::max<size_t>(words, 1) * 4, c.stream));
  DevBuf cnt(4);
  GS_HIP(hipMemsetAsync(cnt.p, 0, 4, c.stream));
  if (n) hipLaunchKernelGGL(k_finite_mask, grid1(n), dim3(256), 0, c.stream, g1_pts, (uint32_t)PointIO<FqTag>::kAffineWords, g2_pts,
                            (uint32_t)PointIO<Fq2Tag>::kAffineWords, n, mask_dev, cnt.as<uint32_t>());
  GS_HIP(hipGetLastError());
  uint32_t host = 0;
  GS_HIP(hipMemcpyAsync(&host, cnt.p, 4, hipMemcpyDeviceToHost, c.stream));
  GS_HIP(hipStreamSynchronize(c.stream));
  return host;
}

void build_plan(Ctx& c, int slot, const uint32_t* scalars_dev, uint32_t n, MsmPlan& plan, const std::vector<LaunchShape>& users, int cbits,
                bool table_free, const uint32_t* term_index, uint32_t index_bias) {
  MsmState& ms = msm_state(c);
  PlanBuffers& pb = ms.plan_slots[slot % (3 * Ctx::kSlots)];
  plan.n = n;
  plan.table_free = table_free;
  plan.c = cbits ? cbits : (table_free ? choose_window_bits_free(n, c.window_bits) : choose_window_bits(n, c.window_bits));
  if (table_free && (plan.c < kMinFreeWindowBits || plan.c > kMaxFreeWindowBits))
    throw HipError{hipErrorInvalidValue, "table-free plan outside 9..16 window bits", __LINE__};
  plan.W = 254 / plan.c + 1;
  plan.B = 1u << (plan.c - 1);
  plan.nbuckets = table_free ? (uint32_t)plan.W * plan.B : plan.B;      // one bucket set for all windows (window tables), or one per window
  plan.chunk = choose_chunk((uint64_t)n * plan.W, plan.nbuckets, users);
  plan.maxchunks = (uint32_t)(((size_t)n * plan.W + plan.chunk - 1) / plan.chunk) + 1;
  const size_t ncount = (size_t)plan.nbuckets + 1;
  PlanParams pp{};
  pp.n = n; pp.c = plan.c; pp.W = plan.W; pp.B = plan.B;
  pp.R = std::max<uint32_t>(1u, plan.B >> kRangeLog);      // bucket ranges of 2^15 counters (one LDS histogram each)
  pp.tf = table_free ? 1u : 0u;
  // slices: enough (window, slice, range) workgroups to cover the 256 CUs, never less than 16384 scalars per slice, and a
  // histogram matrix hist[W * S][B] of at most 64 MiB
  {
    // (round 4: FLOOR, not ceiling -- a histogram / scatter workgroup holds 128 KiB of LDS, so exactly one fits a CU, and at c = 17
    //  the ceiling gave 15 windows x 2 ranges x 9 slices = 270 workgroups for 256 CUs: fourteen stragglers ran a second round of the
    //  whole kernel, k_scatter 0.26 ms instead of ~0.15; 8 slices = 240 workgroups finish in one)
    const uint32_t want = std::max<uint32_t>(1u, 256u / (plan.W * pp.R));
    const uint32_t by_size = std::max<uint32_t>(1u, (n + 16383u) / 16384u);
    const uint32_t by_mem = std::max<uint32_t>(1u, (uint32_t)((64ull << 20) / ((uint64_t)plan.B * plan.W * 4)));
    pp.S = std::max<uint32_t>(1u, std::min(std::min<uint32_t>(16u, std::max<uint32_t>(want, 1u)), std::min(by_size, by_mem)));
    if (pp.R == 1 && !table_free) pp.S = std::max<uint32_t>(1u, std::min<uint32_t>(16u, by_size));     // narrow windows: as before
  }
  pp.slice = (n + pp.S - 1) / pp.S;
  pp.stride = (n + 63u) & ~63u;
  const bool wide = plan_partitions_first(plan.c);         // (= pp.R >= GS_PART_MIN_R)
  if (wide && term_index) throw HipError{hipErrorInvalidValue, "plans over a term list cannot be sorted partition-first (plan_partitions_first)", __LINE__};
  const uint32_t nparts = (uint32_t)plan.W * pp.R;
  if (wide && nparts > kMaxParts) throw HipError{hipErrorInvalidValue, "too many (window, range) partitions", __LINE__};
  if (!wide) pb.digits.ensure((size_t)std::max<uint32_t>(pp.stride, 64u) * plan.W * sizeof(digit_t));
  else {
    pb.recs.ensure(std::max<size_t>((size_t)n * plan.W, 1) * 8);
    pb.parts.ensure((size_t)(3 * kMaxParts + 4) * 4);
  }
  pb.hist.ensure((size_t)plan.B * plan.W * pp.S * 4);
  pb.totals.ensure(ncount * 4);
  pb.offsets.ensure(ncount * 4);
  pb.entries.ensure(((size_t)plan.maxchunks + 1) * plan.chunk * 4);
  pb.chunk_bucket.ensure((size_t)plan.maxchunks * 4);
  pb.heavy_list.ensure((size_t)plan.nbuckets * 4);         // one slot per bucket: the list cannot overflow
  pb.counters.ensure(16);
  const size_t lds = (size_t)std::min<uint32_t>(plan.B, kRangeBuckets) * 4;
  static const int sort_block = (int)dev_knob("GS_SORT_BLOCK", kSortBlock, 64, kSortBlock, 64);
  if (!ms.lds_attr_set) {       // B <= 2^15 counters = 128 KiB of the CU's 160 KiB LDS
    GS_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_hist), hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024));
    GS_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_scatter), hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024));
    GS_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_hist_part), hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024));
    GS_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_scatter_part), hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024));
    ms.lds_attr_set = true;
  }
  GS_HIP(hipMemsetAsync(pb.totals.as<uint32_t>() + plan.nbuckets, 0, 4, c.stream));
  GS_HIP(hipMemsetAsync(pb.counters.p, 0, 16, c.stream));
  uint32_t* part_count = pb.parts.as<uint32_t>();
  uint32_t* part_base = part_count ? part_count + kMaxParts : nullptr;
  uint32_t* part_cursor = part_count ? part_base + kMaxParts + 4 : nullptr;
  if (n > 0 && wide) {
    GS_HIP(hipMemsetAsync(part_count, 0, (size_t)nparts * 4, c.stream));
    hipLaunchKernelGGL(k_part_count, grid1(n, kPartBlock), dim3(kPartBlock), 0, c.stream, scalars_dev, pp, part_count);
    hipLaunchKernelGGL(k_part_scan, dim3(1), dim3(64), 0, c.stream, part_count, nparts, part_base, part_cursor);
    hipLaunchKernelGGL(k_part_scatter, grid1(n, kPartBlock * kPartPerThread), dim3(kPartBlock), 0, c.stream, scalars_dev, pp, part_base, part_cursor,
                       pb.recs.as<uint2>());
    hipLaunchKernelGGL(k_hist_part, dim3(8 * pp.S * ((nparts + 7) / 8)), dim3(sort_block), lds, c.stream, pb.recs.as<uint2>(), part_base, pp, pb.hist.as<uint32_t>());
    hipLaunchKernelGGL(k_colscan, grid1(plan.B), dim3(256), 0, c.stream, pb.hist.as<uint32_t>(), pp, pb.totals.as<uint32_t>());
  } else if (n > 0) {
    hipLaunchKernelGGL(k_digits, grid1(n), dim3(256), 0, c.stream, scalars_dev, pp, pb.digits.as<digit_t>(), term_index, index_bias);
    hipLaunchKernelGGL(k_hist, dim3(plan.W, pp.S, pp.R), dim3(sort_block), lds, c.stream, pb.digits.as<digit_t>(), pp, pb.hist.as<uint32_t>());
    if (table_free) hipLaunchKernelGGL(k_colscan_windows, grid1(plan.nbuckets), dim3(256), 0, c.stream, pb.hist.as<uint32_t>(), pp, pb.totals.as<uint32_t>());
    else hipLaunchKernelGGL(k_colscan, grid1(plan.B), dim3(256), 0, c.stream, pb.hist.as<uint32_t>(), pp, pb.totals.as<uint32_t>());
  } else {
    GS_HIP(hipMemsetAsync(pb.totals.p, 0, ncount * 4, c.stream));
  }
  exclusive_scan(c, pb, pb.totals.as<uint32_t>(), pb.offsets.as<uint32_t>(), (uint32_t)ncount);
  if (n > 0) {
    if (wide) hipLaunchKernelGGL(k_scatter_part, dim3(8 * pp.S * ((nparts + 7) / 8)), dim3(sort_block), lds, c.stream, pb.recs.as<uint2>(), part_base, pp,
                                 pb.hist.as<uint32_t>(), pb.offsets.as<uint32_t>(), pb.entries.as<uint32_t>());
    else hipLaunchKernelGGL(k_scatter, dim3(plan.W, pp.S, pp.R), dim3(sort_block), lds, c.stream, pb.digits.as<digit_t>(), pp, pb.hist.as<uint32_t>(),
                            pb.offsets.as<uint32_t>(), pb.entries.as<uint32_t>(), term_index, index_bias);
    hipLaunchKernelGGL(k_chunk_map, grid1(plan.nbuckets), dim3(256), 0, c.stream, pb.offsets.as<uint32_t>(), plan.nbuckets, plan.chunk,
                       pb.chunk_bucket.as<uint32_t>(), pb.heavy_list.as<uint32_t>(), pb.counters.as<uint32_t>());
  }
  GS_HIP(hipGetLastError());
  plan.offsets = pb.offsets.as<uint32_t>();
  plan.entries = pb.entries.as<uint32_t>();
  plan.chunk_bucket = pb.chunk_bucket.as<uint32_t>();
  plan.heavy_list = pb.heavy_list.as<uint32_t>();
  plan.heavy_count = pb.counters.as<uint32_t>();
}

// Enqueue the kernels that fill rows [first, last) of every row of `rows` (W rows of n points, allocated) from row 0 on `stream`.
// slab: points per launch (2^18 = 1024 workgroups).
template <class T>
static void enqueue_table_slabs(hipStream_t stream, DevBuf& scratch, const uint32_t* src, size_t n, int cbits, DevBuf& rows, size_t first, size_t last,
                                size_t slab) {
  const int W = 254 / cbits + 1;
  if (first >= last) return;
  static const bool per_row = dev_flag("GS_TABLE_PER_ROW");               // the one-inversion-per-row builder, for comparison
  if (per_row || W <= 2) {
    // (whole table at once: this builder has no range form; only reached with W <= 2, i.e. never for BN254 widths <= 20)
    if (first == 0) hipLaunchKernelGGL(k_build_table<T>, grid1(n), dim3(256), 0, stream, src, (uint32_t)n, cbits, W, rows.as<uint32_t>());
  } else {
    // slabs: (W - 1) rows of [XYZZ | running product] raw limbs per point (<= 1.4 GiB for G2 at 2^18 points), reused per slab
    constexpr size_t sw = PointIO<T>::kXyzzWords + PointIO<T>::kXyzzWords / 4;
    slab = std::min<size_t>(std::max<size_t>(slab, 1), n);
    scratch.ensure(slab * (size_t)(W - 1) * sw * 4);
    for (size_t at = first; at < last; at += slab) {
      const size_t count = std::min(slab, last - at);
      hipLaunchKernelGGL(k_build_table_batched<T>, grid1(count), dim3(256), 0, stream, src, (uint32_t)n, (uint32_t)at, (uint32_t)count, cbits, W,
                         rows.as<uint32_t>(), scratch.as<uint32_t>());
    }
  }
  GS_HIP(hipGetLastError());
}
template <class T>
static void enqueue_table_build(Ctx& c, hipStream_t stream, DevBuf& scratch, const uint32_t* src, size_t n, int cbits, DevBuf& fresh,
                                size_t slab_max = (size_t)1 << 18) {
  constexpr size_t aw = PointIO<T>::kAffineWords;
  const int W = 254 / cbits + 1;
  fresh.alloc(std::max<size_t>(n, 1) * W * aw * 4);
  enqueue_table_slabs<T>(stream, scratch, src, n, cbits, fresh, 0, n, slab_max);
}

template <class T>
static void ensure_table(Ctx& c, BaseTable& t, const uint32_t* row0, size_t n, int cbits) {
  if (t.c == cbits && t.n == n && t.rows.p) return;
  table_settle(c, t, false);                    // a pending build of another width: superseded
  const uint32_t* src = row0 ? row0 : t.rows.as<uint32_t>();
  if (!src || (!row0 && t.n != n)) throw HipError{hipErrorInvalidValue, "window table rebuild without its points", __LINE__};
  DevBuf fresh;
  enqueue_table_build<T>(c, c.stream, msm_state(c).table_scratch, src, n, cbits, fresh);
  GS_HIP(hipStreamSynchronize(c.stream));       // the old rows (possibly the source) are released below
  t.rows = std::move(fresh);
  t.n = n; t.c = cbits; t.W = 254 / cbits + 1;
  t.uses = 0;
}
void ensure_table_g1(Ctx& c, BaseTable& t, const uint32_t* row0, size_t n, int cbits) { ensure_table<FqTag>(c, t, row0, n, cbits); }
void ensure_table_g2(Ctx& c, BaseTable& t, const uint32_t* row0, size_t n, int cbits) { ensure_table<Fq2Tag>(c, t, row0, n, cbits); }

// ---- when a base array gets its table (gs_set_table_policy) ---------------------------------------------------------------------
// The reference proves ONCE per key load (cli/main.go:330-349); 15-row tables cost ~140 ms and 15x the key's memory at 2^20 -- fifteen
// proofs' worth of work before the first one.  Under `auto` a base array is summed table-free until it has been used
// kTableAfterUses times; from then on every call that finds it without a table builds a few more slabs of it.
//
// Round 6: IN INSTALMENTS, in front of the call's own accumulations, on the accumulation stream.  Round 5 enqueued
// all five builds of a key at once on a lowest-priority stream and let the hardware share the chip: proofs 2-10 of a fresh 2^20 key took
// 19-35 ms each, unpredictably, and a proof could not use a table the moment it was complete.  The work is what it is -- ~125 ms of
// full-chip time per 2^20 Groth16 key, and whatever runs beside it is slowed by what it takes ("what overlap can and
// cannot buy") -- so the only choice is how it is spread over the calls: each call gets a build CREDIT proportional to its own work
// (msm.h, kBuildCreditPerUnitTerm: ~+80 % of a table-free call), spends it on whole slabs of the pending table (the balance, at most one
// slab either way, stays on the context), and the call that enqueues the last slab installs the table and uses it -- stream order makes
// that safe without a host wait.  A fresh 2^20 key: 24 ms, then ~16 proofs of <= 2x the steady time, then steady; a proof never waits for
// more than its own instalment.  GS_TABLE_BUDGET_PCT scales the credit (100 = as described; 100000 = everything inside the second call).
constexpr uint32_t kTableAfterUses = 2;

template <class T>
static void enqueue_pending_slabs(Ctx& c, BaseTable& t, size_t first, size_t last, size_t slab) {
  enqueue_table_slabs<T>(c.main_stream, msm_state(c).table_scratch_bg, t.pending_src, t.pending_n, t.pending_c, t.pending, first, last, slab);
}
static size_t instalment_slab(size_t n, bool g2) {
  // As LARGE as possible: a builder thread walks its point through 14 x 17 dependent doublings, ~1.3 ms for a lone wave whatever the
  // slab holds, so a slab must fill the chip to be worth its launch.  2^18 points for G1 (4 waves per SIMD: 5 ms) and 2^17 for G2
  // (5.7 ms), the whole table when it is smaller (GS_TABLE_BG_SLAB_LOG2 caps the G1 figure).  Measured on fresh keys
  // (profiles/r06_auto_instalments.txt): with slabs of n / 8 points a 2^16 key's tables took 75 ms of instalments instead of 9, and
  // 2^17-point G1 slabs cost a 2^20 key 168 ms against the 130 ms `always` spends in 2^18-point launches.
  static const size_t cap = (size_t)1 << run_knob("GS_TABLE_BG_SLAB_LOG2", 18, 10, 18);
  return std::min(g2 ? std::max<size_t>(cap / 2, (size_t)1 << 12) : cap, std::max<size_t>(n, 1));
}
// all streams of the context wait for the table's last slab (accumulations run on the main stream, where the slabs are: this is for
// whatever else may come to read a table)
static void order_streams_behind(Ctx& c, hipEvent_t ev) {
  for (auto a : c.aux_stream) if (a && a != c.main_stream) GS_HIP(hipStreamWaitEvent(a, ev, 0));
}

void table_settle(Ctx& c, BaseTable& t, bool install) {
  if (!t.pending.p) return;
  if (install && !t.pending_complete()) {              // finish the missing instalments now (gs_build_tables on a key that was warming up)
    const size_t slab = (size_t)1 << 18;
    if (t.pending_g2) enqueue_pending_slabs<Fq2Tag>(c, t, t.pending_next, t.pending_n, slab);
    else enqueue_pending_slabs<FqTag>(c, t, t.pending_next, t.pending_n, slab);
    t.pending_next = t.pending_n;
    GS_HIP(hipEventRecord(t.pending_done, c.main_stream));
  }
  // this table's own build only
  if (t.pending_done && t.pending_next > 0) GS_HIP(hipEventSynchronize(t.pending_done));
  else if (t.pending_next > 0) GS_HIP(hipStreamSynchronize(c.main_stream));
  if (install) {
    t.rows = std::move(t.pending);
    t.n = t.pending_n; t.c = t.pending_c; t.W = 254 / t.pending_c + 1;
    t.uses = 0;
  } else t.pending.release();
  t.pending_c = 0; t.pending_n = 0; t.pending_next = 0; t.pending_src = nullptr;
}

// install a table whose last slab has been ENQUEUED (not necessarily executed): everything that reads it is ordered behind the slab
static void install_enqueued(Ctx& c, BaseTable& t) {
  if (hipEventQuery(t.pending_done) != hipSuccess) order_streams_behind(c, t.pending_done);
  (void)hipGetLastError();                                                     // hipErrorNotReady is not an error
  // (the old rows, if any, are released by the move: hipFree waits for the device, i.e. for every reader of them)
  t.rows = std::move(t.pending);
  t.n = t.pending_n; t.c = t.pending_c; t.W = 254 / t.pending_c + 1;
  t.uses = 0;
  t.pending_c = 0; t.pending_n = 0; t.pending_next = 0; t.pending_src = nullptr;
}

static bool begin_pending(Ctx& c, BaseTable& t, const TableRef& r, int cbits) {
  if (!t.pending_done) GS_HIP(hipEventCreateWithFlags(&t.pending_done, hipEventDisableTiming));
  const int W = 254 / cbits + 1;
  const size_t aw = r.g2 ? PointIO<Fq2Tag>::kAffineWords : PointIO<FqTag>::kAffineWords;
  try {
    // the slab scratch at its G2 size BEFORE the first launch: growing it later releases the old buffer, and hipFree waits for the
    // device -- for the slabs just enqueued (round 5: that, not the builds' share of the chip, was most of a second proof's 88 ms)
    constexpr size_t sw2 = PointIO<Fq2Tag>::kXyzzWords + PointIO<Fq2Tag>::kXyzzWords / 4;
    if (W > 2) {      // (a G1 slab may be twice a G2 slab's points, and a G1 point's scratch is half a G2 point's: the same bytes)
      constexpr size_t sw1 = PointIO<FqTag>::kXyzzWords + PointIO<FqTag>::kXyzzWords / 4;
      const size_t s2 = instalment_slab(r.n, true), s1 = instalment_slab(r.n, false);
      const size_t bytes = std::max((s2 + s2 / 4) * sw2, (s1 + s1 / 4) * sw1);
      msm_state(c).table_scratch_bg.ensure(bytes * (size_t)(W - 1) * 4);
    }
    t.pending.alloc(std::max<size_t>(r.n, 1) * W * aw * 4);
  } catch (const HipError& e) {
    if (e.e != hipErrorOutOfMemory) throw;      // no room for a table: keep summing table-free
    t.pending.release();
    return false;
  }
  t.pending_c = cbits; t.pending_n = r.n; t.pending_next = 0; t.pending_src = r.row0; t.pending_g2 = r.g2;
  return true;
}

// spend credit on the next slabs of t's pending table; true when the last slab has been enqueued
static bool advance_pending(Ctx& c, BaseTable& t, double& credit) {
  const double unit = t.pending_g2 ? kG2BuildCost : 1.0;
  const size_t slab = instalment_slab(t.pending_n, t.pending_g2);
  for (;;) {
    if (t.pending_next >= t.pending_n) break;
    // (a remainder of up to a quarter slab rides with the last one: a key of 2^k + 1 variables must not end on a one-point launch)
    const size_t last = t.pending_n - t.pending_next <= slab + slab / 4 ? t.pending_n : t.pending_next + slab;
    // a slab is bought when the credit covers at least half of it (the call overdraws by at most half a slab, ~2.5 ms at 2^20, and the
    // next call's grant is that much smaller): the instalments of consecutive calls then differ by one slab at most
    if (credit < 0.5 * (double)(last - t.pending_next) * unit) break;
    if (t.pending_g2) enqueue_pending_slabs<Fq2Tag>(c, t, t.pending_next, last, last - t.pending_next);
    else enqueue_pending_slabs<FqTag>(c, t, t.pending_next, last, last - t.pending_next);
    credit -= (double)(last - t.pending_next) * unit;
    t.pending_next = last;
  }
  if (t.pending_next < t.pending_n) return false;
  GS_HIP(hipEventRecord(t.pending_done, c.main_stream));
  return true;
}

void stamp_tables(Ctx& c, std::initializer_list<BaseTable*> tables) {
  for (BaseTable* t : tables) if (t) t->last_use = c.call_clock;
}

bool prepare_tables(Ctx& c, const std::vector<TableRef>& group, uint32_t nterms, int* cbits, double* credit) {
  int cb = choose_window_bits(std::max<uint32_t>(nterms, 1u), c.window_bits);
  // A group whose tables are all resident at ONE width next to the model's serves at that width (gs_build_tables sizes a
  // key's tables for the arrays it holds, a proof asks for the width of its term range -- near a boundary of the model the warmed
  // table was silently ignored or rebuilt, and callers alternating two MSM lengths over one array flip-flopped it).
  if (c.window_bits == 0 && !group.empty() && group[0].t->rows.p) {
    const int tc = group[0].t->c;
    bool same = tc >= cb - 1 && tc <= cb + 1;
    for (const TableRef& r : group) same = same && r.t->ready(r.n, tc);
    if (same) cb = tc;
  }
  bool all_ready = true;
  for (const TableRef& r : group) {
    r.t->last_use = c.call_clock;
    all_ready = all_ready && r.t->ready(r.n, cb);
  }
  if (all_ready) {
    for (const TableRef& r : group) r.t->last_table_use = c.call_clock;
    *cbits = cb;
    return true;
  }
  if (c.table_policy == 1) {                                                   // always: inside the call, as rounds 1-4 did
    for (const TableRef& r : group) {
      if (r.g2) ensure_table_g2(c, *r.t, r.row0, r.n, cb); else ensure_table_g1(c, *r.t, r.row0, r.n, cb);
      r.t->last_table_use = c.call_clock;
    }
    *cbits = cb;
    return true;
  }
  if (c.table_policy == 0) {
    static const double scale = (double)run_knob("GS_TABLE_BUDGET_PCT", 100, 1, 1000000) / 100.0;
    // what this call may still enqueue: its grant (what an earlier group of the same call left of it) minus the last call's overdraft
    const double grant = credit ? *credit : 0.0;
    double avail = grant * scale + c.build_balance;
    // G2 arrays first: the G2 sum is the longest of a proof, so its table is the one that pays most per call
    std::vector<const TableRef*> order;
    for (const TableRef& r : group) if (r.g2) order.push_back(&r);
    for (const TableRef& r : group) if (!r.g2) order.push_back(&r);
    for (const TableRef* pr : order) {
      const TableRef& r = *pr;
      BaseTable& t = *r.t;
      if (t.ready(r.n, cb)) continue;
      if (t.pending.p && (t.pending_c != cb || t.pending_n != r.n || t.pending_src != r.row0)) table_settle(c, t, false);   // superseded
      if (!t.pending.p) {
        t.uses += 1;
        // a table of another width that still serves other calls stays (callers alternating two lengths over one array)
        const bool serving = t.rows.p && c.call_clock - t.last_table_use <= 8;
        if (t.uses < kTableAfterUses || !r.n || serving) continue;
        if (!begin_pending(c, t, r, cb)) continue;
      }
      if (advance_pending(c, t, avail)) install_enqueued(c, t);
    }
    c.build_balance = std::min(avail, 0.0);                       // an overdraft (less than one slab) comes off the next call's grant
    if (credit) *credit = std::max(avail, 0.0) / scale;           // the rest of the grant is the call's next group's
    bool now_ready = true;
    for (const TableRef& r : group) now_ready = now_ready && r.t->ready(r.n, cb);
    if (now_ready) {                                              // this call enqueued the last slab: it is the first to use the tables
      for (const TableRef& r : group) r.t->last_table_use = c.call_clock;
      *cbits = cb;
      return true;
    }
  }
  *cbits = choose_window_bits_free(std::max<uint32_t>(nterms, 1u), c.window_bits);
  return false;
}

template <class T>
static void msm_enqueue(Ctx& c, const MsmPlan& plan, const std::vector<MsmBase>& bases, int ws_base, int slot, MsmPending& p,
                        hipStream_t tail_stream) {
  const int njobs = (int)bases.size();
  p = MsmPending{};
  p.njobs = njobs; p.slot = slot; p.n = plan.n; p.g2 = PointIO<T>::kAffineWords == 32;
  p.c = plan.c; p.W = plan.W;
  if (njobs == 0 || plan.n == 0) { p.njobs = plan.n == 0 ? -njobs : 0; return; }
  if (njobs > kMaxJobs) throw HipError{hipErrorInvalidValue, "too many MSM jobs", __LINE__};
  constexpr size_t pw = PointIO<T>::kXyzzWords;
  constexpr size_t aw = PointIO<T>::kAffineWords;
  // buckets per reduce thread: 4, or as many as it takes (up to 32) to stay at <= 32 workgroup pairs, which the host folds while
  // the device moves on (3 host additions per pair) -- the on-device fold (k_pair_reduce) is one more ~35-deep chain of dependent
  // additions in the tail.  The thread's chain is 2 L additions deep (+ 16 for the scan and the tree): L = 8 at 65536 buckets.
  // Measured (profiles/r02_ab_reduce_depth.txt): tails of a 2^20 proof alone 2.44 -> 1.9 ms against (16 pairs, L = 16).
  // (32 pairs only from 2^19 terms on: a 2^17 / 2^18 proof takes 1.6 / 2.7 ms, and three host additions per pair for five jobs then
  // make the host the pace-setter in some repetitions -- median 2.0 vs 1.6 ms at 2^17 with the same best case.)
  static const uint32_t fold_env = (uint32_t)dev_knob("GS_FOLD_MAX", 0, 1, 256);
  const uint32_t fold_max = fold_env ? fold_env : (plan.n >= (1u << 19) ? 32u : 16u);
  // (a power of two: the reduce multiplies by L through log2(L) doublings, and the host fold by 256 L likewise)
  static const uint32_t l_min = (uint32_t)dev_knob("GS_REDUCE_L", 4, 1, 32, 1, true);
  int L = (int)std::max<uint32_t>(1u, std::min<uint32_t>(l_min, plan.B / kReduceBlock));
  while (L < 32 && plan.B / ((uint32_t)kReduceBlock * (uint32_t)L) > fold_max) L *= 2;
  uint32_t nblk = (plan.B + kReduceBlock * L - 1) / (kReduceBlock * L);
  if (plan.table_free) {
    // Per-window bucket sets (B >= 256 = one reduce workgroup, a power of two: a workgroup never straddles two windows).  The same
    // kernel reduces all W * B buckets; workgroup blk belongs to window blk / (B / (256 L)) and the host folds each window's pairs,
    // then recombines the window sums by Horner.  L = 16 keeps the pairs of a 6-job Pinocchio group inside the pinned slot.
    L = (int)std::max<uint32_t>(1u, std::min<uint32_t>(16u, plan.B / kReduceBlock));
    p.nblk_window = plan.B / ((uint32_t)kReduceBlock * (uint32_t)L);
    nblk = p.nblk_window * (uint32_t)plan.W;
    p.table_free = true;
  }
  p.L = L; p.nblk = nblk;
  p.folded = !plan.table_free && nblk > fold_max;         // wide windows: the pairs are folded on the device (k_pair_reduce)
  if (p.folded && nblk > (uint32_t)kReduceBlock) throw HipError{hipErrorInvalidValue, "too many reduce workgroups for one fold", __LINE__};
  // result staging: [pairs | finals | stats], downloaded in ONE copy: pairs .. stats (host fold) or finals .. stats (device fold).
  // stats = {bucket entries of the plan, buckets combined by the heavy tree} (k_bucket_combine writes them; gs_timing reports them)
  const size_t pair_bytes = (size_t)njobs * nblk * 2 * pw * 4, final_bytes = (size_t)njobs * pw * 4, stats_bytes = 16;
  const size_t out_bytes = (p.folded ? final_bytes : pair_bytes + final_bytes) + stats_bytes;
  if (out_bytes > Ctx::kPinnedBytes) throw HipError{hipErrorInvalidValue, "MSM result staging too small", __LINE__};
  p.stats_off = out_bytes - stats_bytes;
  AccJobs jobs{};
  DevBuf& outb = c.ws_out[ws_base % Ctx::kWsSets];
  outb.ensure(pair_bytes + final_bytes + stats_bytes);
  uint32_t* finals = outb.as<uint32_t>() + pair_bytes / 4;
  uint32_t* stats = finals + final_bytes / 4;
  for (int j = 0; j < njobs; ++j) {
    const BaseTable* t = bases[j].table;
    const uint32_t* points = nullptr;
    uint32_t row_stride = 0;
    if (plan.table_free) {               // every window adds the base point itself: "row" w of a table with stride 0
      if (!bases[j].points || bases[j].off + plan.n > bases[j].npoints)
        throw HipError{hipErrorInvalidValue, "MSM base array does not cover the plan", __LINE__};
      points = bases[j].points + bases[j].off * aw;
    } else {
      if (!t || t->c != plan.c || !t->rows.p || bases[j].off + plan.n > t->n)
        throw HipError{hipErrorInvalidValue, "MSM base table does not match the plan", __LINE__};
      points = t->rows.as<uint32_t>() + bases[j].off * aw;
      row_stride = (uint32_t)t->n;
    }
    DevBuf& bk = c.ws_buckets[(ws_base + j) % Ctx::kWsSets];
    DevBuf& mg = c.ws_chunks[(ws_base + j) % Ctx::kWsSets];
    DevBuf& pt = c.ws_partials[(ws_base + j) % Ctx::kWsSets];
    bk.ensure((size_t)plan.nbuckets * pw * 4);
    mg.ensure((size_t)plan.nbuckets * pw * 4);
    pt.ensure((size_t)plan.maxchunks * 2 * pw * 4);
    jobs.j[j] = AccJob{points, row_stride, bk.as<uint32_t>(), pt.as<uint32_t>(),
                       pt.as<uint32_t>() + (size_t)plan.maxchunks * pw, mg.as<uint32_t>(),
                       outb.as<uint32_t>() + (size_t)j * nblk * 2 * pw, finals + (size_t)j * pw};
  }
  p.tacc = std::make_shared<PhaseTimer>(c.stream);
  p.tker = std::make_shared<PhaseTimer>(c.stream);
  // (round 6 tried one launch per base array instead of grid.y = njobs -- the proof stream runs at sclk 2.10 GHz where an MSM stream, one
  //  array per launch, runs at 2.31: no change in time or clock, profiles/r06_power_clock_streams.txt)
  hipLaunchKernelGGL(k_bucket_accumulate<T>, dim3((plan.maxchunks + kAccBlock - 1) / kAccBlock, njobs), dim3(kAccBlock), 0, c.stream,
                     jobs, plan.offsets, plan.entries, plan.chunk_bucket, plan.nbuckets, plan.chunk);
  p.tker->stop();
  p.tacc->stop();
  // the latency-bound tail may run on another stream, in the shadow of the next group's accumulation
  hipStream_t ts = tail_stream ? tail_stream : c.stream;
  if (ts != c.stream) {
    hipEvent_t ev;
    GS_HIP(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
    GS_HIP(hipEventRecord(ev, c.stream));
    GS_HIP(hipStreamWaitEvent(ts, ev, 0));
    GS_HIP(hipEventDestroy(ev));       // released by the runtime once it has fired
  }
  p.tred = std::make_shared<PhaseTimer>(ts);
  // (the block-wide tree for buckets cut into very many chunks belongs to the tail too: with uniform scalars it finds nothing
  // to do, and on the accumulation stream even an empty launch waited ~0.5 ms for register space)
  // small MSMs: one tail wave per SIMD (msm_kernels.h, kAlone); from 2^19 terms on the tails share their SIMDs with accumulation waves
  static const long alone_below_log2 = dev_knob("GS_TAIL_ALONE_LOG2", 19, 0, 32);
  const bool alone = plan.n < (1ull << alone_below_log2);
  auto launch_tails = [&](auto alone_tag) {
    constexpr bool kAlone = decltype(alone_tag)::value;
    // (round 6 tried the two heavy-bucket kernels always in their sharing form -- with uniform scalars they find nothing to do, and the
    //  one-wave-per-SIMD form must wait for empty SIMDs: no change at 2^16-2^18 with uniform scalars, +2 % on a realistic 2^18 witness,
    //  where they do have work: profiles/r06_ab_heavy_kernels_sharing.txt.  Not adopted.)
    hipLaunchKernelGGL((k_heavy_combine<T, kAlone>), dim3(64, njobs), dim3(kHeavyBlock), 0, ts,
                       jobs, plan.offsets, plan.heavy_list, plan.heavy_count, plan.chunk);
    hipLaunchKernelGGL((k_heavy_finish<T, kAlone>), dim3(16, njobs), dim3(kHeavyBlock), 0, ts,
                       jobs, plan.offsets, plan.heavy_list, plan.heavy_count, plan.chunk);
    hipLaunchKernelGGL((k_bucket_combine<T, kAlone>), dim3((plan.nbuckets + 255) / 256, njobs), dim3(256), 0, ts, jobs, plan.offsets, plan.nbuckets,
                       plan.chunk, plan.heavy_count, stats);
    hipLaunchKernelGGL((k_block_reduce<T, kAlone>), dim3(nblk, njobs), dim3(kReduceBlock), 0, ts, jobs, plan.nbuckets, L);
    if (p.folded) {
      int log2_span = 0;
      while ((1u << log2_span) < (uint32_t)kReduceBlock * (uint32_t)L) ++log2_span;
      hipLaunchKernelGGL((k_pair_reduce<T, kAlone>), dim3(1, njobs), dim3(kReduceBlock), 0, ts, jobs, nblk, log2_span);
    }
  };
  if (alone) launch_tails(std::true_type{}); else launch_tails(std::false_type{});
  GS_HIP(hipGetLastError());
  GS_HIP(hipMemcpyAsync(c.pinned[slot], p.folded ? (const void*)finals : outb.p, out_bytes, hipMemcpyDeviceToHost, ts));
  p.pinned_slot = c.pinned[slot];
  p.tred->stop();
}

// host side of an MSM group: books the device timings (serial), then adds the <= 16 workgroup pairs of every job --
// result = sum_blk A_blk + (256 L) * sum_blk blk * S_blk -- one host core per job.
template <class T>
static Xyzz<T> sum_pairs(const Xyzz<T>* pr, uint32_t nblk, int L) {
  Xyzz<T> run = xyzz_inf<T>(), tot = xyzz_inf<T>(), sumA = xyzz_inf<T>();
  for (uint32_t blk = nblk; blk-- > 0;) {
    xyzz_add(sumA, pr[2 * blk]);
    if (blk >= 1) { xyzz_add(run, pr[2 * blk + 1]); xyzz_add(tot, run); }
  }
  for (uint32_t s2 = (uint32_t)kReduceBlock * (uint32_t)L; s2 > 1; s2 >>= 1) xyzz_dbl(tot);
  xyzz_add(sumA, tot);
  return sumA;
}

// Books the device timings of a group.  HIP calls (event queries) stay on the thread that drives the device; only the pure
// host arithmetic of msm_finish runs on worker threads.
void msm_book_timing(Ctx& c, const MsmPending& p) {
  if (p.njobs <= 0) return;
  std::lock_guard<std::mutex> lk(c.timing_mu);
  c.timing.accumulate_ms += p.tacc->ms();
  const uint64_t terms = (uint64_t)p.n * p.njobs;
  c.timing.window_bits = (uint32_t)p.c;
  c.timing.reduce_ms += p.tred->ms();
  // the plan's own counts travel behind the group's results (the download has completed: tred's stop event follows it): a term
  // has one digit per window, a zero digit costs nothing, every other one is exactly one mixed addition of the accumulation kernel
  const uint32_t* st = p.pinned_slot ? reinterpret_cast<const uint32_t*>(static_cast<const char*>(p.pinned_slot) + p.stats_off) : nullptr;
  const uint64_t adds = st ? (uint64_t)st[0] * p.njobs : terms * p.W;
  if (!p.g2) { c.timing.acc_g1_ms += p.tker->ms(); c.timing.acc_g1_launches += 1; c.timing.acc_g1_terms += terms; c.timing.acc_g1_adds += adds; }
  else { c.timing.acc_g2_ms += p.tker->ms(); c.timing.acc_g2_launches += 1; c.timing.acc_g2_terms += terms; c.timing.acc_g2_adds += adds; }
  c.timing.plan_digits += terms * p.W;
  c.timing.plan_entries += adds;
  // (per MSM GROUP: the G2 and the G1 group over w share one plan, so a proof books that plan's heavy buckets twice -- what the
  //  heavy-bucket kernels really processed, once per group; include/gosnark_hip.h says so)
  if (st) c.timing.heavy_buckets += st[1];
}

// table-free route: the pairs of a job come window by window (nblk_window each); S_w = sum_pairs of window w, and the result is
// sum_w 2^(c w) S_w by Horner from the top window down (W c doublings + W additions on a host core).
template <class T>
static Xyzz<T> sum_pairs_windows(const Xyzz<T>* pr, const MsmPending& p) {
  Xyzz<T> acc = xyzz_inf<T>();
  for (int w = p.W; w-- > 0;) {
    for (int k = 0; k < p.c; ++k) xyzz_dbl(acc);
    xyzz_add(acc, sum_pairs<T>(pr + (size_t)w * p.nblk_window * 2, p.nblk_window, p.L));
  }
  return acc;
}

template <class T>
static void msm_finish(Ctx& c, const MsmPending& p, std::vector<Xyzz<T>>& out) {
  if (p.njobs <= 0) { out.assign(-p.njobs, xyzz_inf<T>()); return; }
  out.assign(p.njobs, xyzz_inf<T>());
  const Xyzz<T>* pairs = static_cast<const Xyzz<T>*>(c.pinned[p.slot]);
  if (p.folded) {                                          // one point per job came back
    for (int j = 0; j < p.njobs; ++j) out[j] = pairs[j];
    return;
  }
  auto one = [pairs, &p](int j) {
    const Xyzz<T>* pr = pairs + (size_t)j * p.nblk * 2;
    return p.table_free ? sum_pairs_windows<T>(pr, p) : sum_pairs<T>(pr, p.nblk, p.L);
  };
  std::vector<std::future<Xyzz<T>>> fut;
  for (int j = 1; j < p.njobs; ++j) fut.push_back(std::async(std::launch::async, [one, j] { return one(j); }));
  out[0] = one(0);
  for (int j = 1; j < p.njobs; ++j) out[j] = fut[j - 1].get();
}

void msm_enqueue_g1(Ctx& c, const MsmPlan& plan, const std::vector<MsmBase>& bases, int ws_base, int slot, MsmPending& p, hipStream_t tail) {
  msm_enqueue<FqTag>(c, plan, bases, ws_base, slot, p, tail);
}
void msm_enqueue_g2(Ctx& c, const MsmPlan& plan, const std::vector<MsmBase>& bases, int ws_base, int slot, MsmPending& p, hipStream_t tail) {
  msm_enqueue<Fq2Tag>(c, plan, bases, ws_base, slot, p, tail);
}
void msm_finish_g1(Ctx& c, const MsmPending& p, std::vector<G1Xyzz>& out) { msm_finish<FqTag>(c, p, out); }
void msm_finish_g2(Ctx& c, const MsmPending& p, std::vector<G2Xyzz>& out) { msm_finish<Fq2Tag>(c, p, out); }
void msm_run_g1(Ctx& c, const MsmPlan& plan, const std::vector<MsmBase>& bases, std::vector<G1Xyzz>& out) {
  MsmPending p;
  msm_enqueue<FqTag>(c, plan, bases, 8 * c.blocking_slot(), 3 * c.blocking_slot(), p, nullptr);
  GS_HIP(hipStreamSynchronize(c.stream));
  msm_book_timing(c, p);
  msm_finish<FqTag>(c, p, out);
}
void msm_run_g2(Ctx& c, const MsmPlan& plan, const std::vector<MsmBase>& bases, std::vector<G2Xyzz>& out) {
  MsmPending p;
  msm_enqueue<Fq2Tag>(c, plan, bases, 8 * c.blocking_slot() + 4, 3 * c.blocking_slot(), p, nullptr);
  GS_HIP(hipStreamSynchronize(c.stream));
  msm_book_timing(c, p);
  msm_finish<Fq2Tag>(c, p, out);
}

// Returns the number of points that are not on their curve (and the first such index); synchronises the stream.
template <class T>
static uint32_t jacobian_to_affine_checked(Ctx& c, const uint32_t* jac, uint32_t n, uint32_t* out, uint32_t* first_bad) {
  if (!n) return 0;
  DevBuf& flag = msm_state(c).upload_flag;          // (the context's: a hipMalloc and a hipFree -- a device-wide wait -- per call otherwise)
  flag.ensure(8);
  const uint32_t init[2] = {0u, 0xffffffffu};
  GS_HIP(hipMemcpyAsync(flag.p, init, 8, hipMemcpyHostToDevice, c.stream));
  hipLaunchKernelGGL(k_jacobian_to_affine<T>, grid1(n), dim3(256), 0, c.stream, jac, n, out, flag.as<uint32_t>());
  GS_HIP(hipGetLastError());
  uint32_t res[2] = {0, 0};
  GS_HIP(hipMemcpyAsync(res, flag.p, 8, hipMemcpyDeviceToHost, c.stream));
  GS_HIP(hipStreamSynchronize(c.stream));
  if (first_bad) *first_bad = res[1];
  return res[0];
}
// gs_g1_upload / gs_g2_upload: n Jacobian triples (standard form) in caller memory -> packed affine Montgomery points at `out`.  The
// triples come in through the pinned staging buffers and the copy threads (hostcopy.h) into a scratch the context keeps while it is small:
// a 2^20 key is five such uploads, and each used to pay three hipMalloc and two hipFree (each a device-wide wait) around 2 ms of copying.
constexpr size_t kUploadScratchKeep = (size_t)256 << 20;
template <class T>
static uint32_t upload_jacobian(Ctx& c, const uint64_t* jac_host, uint32_t n, uint32_t* out, uint32_t* first_bad) {
  if (!n) return 0;
  const size_t bytes = (size_t)n * 3 * T::kWords * 4;
  DevBuf once;
  DevBuf& scratch = bytes <= kUploadScratchKeep ? msm_state(c).upload_scratch : once;
  scratch.ensure(bytes);
  staged_h2d(c, scratch.p, jac_host, bytes, c.stream);
  return jacobian_to_affine_checked<T>(c, scratch.as<uint32_t>(), n, out, first_bad);       // synchronises the stream: `once` may go
}
uint32_t upload_jacobian_g1(Ctx& c, const uint64_t* jac_host, uint32_t n, uint32_t* out, uint32_t* first_bad) { return upload_jacobian<FqTag>(c, jac_host, n, out, first_bad); }
uint32_t upload_jacobian_g2(Ctx& c, const uint64_t* jac_host, uint32_t n, uint32_t* out, uint32_t* first_bad) { return upload_jacobian<Fq2Tag>(c, jac_host, n, out, first_bad); }
uint32_t jacobian_to_affine_g1(Ctx& c, const uint32_t* jac, uint32_t n, uint32_t* out, uint32_t* first_bad) {
  return jacobian_to_affine_checked<FqTag>(c, jac, n, out, first_bad);
}
uint32_t jacobian_to_affine_g2(Ctx& c, const uint32_t* jac, uint32_t n, uint32_t* out, uint32_t* first_bad) {
  return jacobian_to_affine_checked<Fq2Tag>(c, jac, n, out, first_bad);
}
void affine_to_jacobian_std_g1(Ctx& c, const uint32_t* aff, uint32_t n, uint32_t* out) {
  if (n) hipLaunchKernelGGL(k_affine_to_jacobian_std<FqTag>, grid1(n), dim3(256), 0, c.stream, aff, n, out);
  GS_HIP(hipGetLastError());
}
void affine_to_jacobian_std_g2(Ctx& c, const uint32_t* aff, uint32_t n, uint32_t* out) {
  if (n) hipLaunchKernelGGL(k_affine_to_jacobian_std<Fq2Tag>, grid1(n), dim3(256), 0, c.stream, aff, n, out);
  GS_HIP(hipGetLastError());
}

template <class T>
static void fixed_base(Ctx& c, DevBuf& table, const uint32_t* scalars, uint32_t n, uint32_t* out) {
  if (table.p == nullptr) {
    // 2^j * G for j < 256, then the 32 x 256 window table d * 2^(8 w) * G that the batch kernel reads
    table.alloc((size_t)32 * 256 * PointIO<T>::kAffineWords * 4);
    DevBuf pow2((size_t)256 * PointIO<T>::kAffineWords * 4), chain((size_t)256 * PointIO<T>::kXyzzWords * 4);
    hipLaunchKernelGGL(k_build_pow2_table<T>, dim3(1), dim3(256), 0, c.stream, pow2.as<uint32_t>(), chain.as<uint32_t>());
    hipLaunchKernelGGL(k_build_fixed_window_table<T>, dim3(32), dim3(256), 0, c.stream, pow2.as<uint32_t>(), table.as<uint32_t>());
    GS_HIP(hipStreamSynchronize(c.stream));       // `pow2` and `chain` are released here
  }
  if (n) hipLaunchKernelGGL(k_fixed_base_mul<T>, grid1(n), dim3(256), 0, c.stream, scalars, n, table.as<uint32_t>(), out);
  GS_HIP(hipGetLastError());
}
void fixed_base_g1(Ctx& c, const uint32_t* s, uint32_t n, uint32_t* out) { fixed_base<FqTag>(c, c.g1_pow2, s, n, out); }
void fixed_base_g2(Ctx& c, const uint32_t* s, uint32_t n, uint32_t* out) { fixed_base<Fq2Tag>(c, c.g2_pow2, s, n, out); }

// ---- host-side serial helpers ---------------------------------------------------------------------------
void fr_canon_words(const uint64_t k[4], uint32_t out[8]) {
  uint32_t t[8];
  for (int i = 0; i < 4; ++i) { t[2 * i] = (uint32_t)k[i]; t[2 * i + 1] = (uint32_t)(k[i] >> 32); }
  scalar_canon(t);
  for (int i = 0; i < 8; ++i) out[i] = t[i];
}

template <class T>
static Affine<T> affine_from_jac_std(const uint64_t* jac) {
  constexpr int cw = PointIO<T>::kCoordWords;
  const uint32_t* p = reinterpret_cast<const uint32_t*>(jac);
  auto X = PointIO<T>::load_std(p), Y = PointIO<T>::load_std(p + cw), Z = PointIO<T>::load_std(p + 2 * cw);
  return jacobian_to_affine<T>(X, Y, Z);
}
G1Affine g1_affine_from_jacobian_std(const uint64_t jac[12]) { return affine_from_jac_std<FqTag>(jac); }
G2Affine g2_affine_from_jacobian_std(const uint64_t jac[24]) { return affine_from_jac_std<Fq2Tag>(jac); }

template <class T>
static bool to_affine_std(const Xyzz<T>& p, uint64_t* out) {
  constexpr int cw = PointIO<T>::kCoordWords;
  uint32_t* o = reinterpret_cast<uint32_t*>(out);
  memset(o, 0, 2 * cw * 4);
  if (is_inf(p)) return true;
  Affine<T> a = xyzz_to_affine(p);
  PointIO<T>::store_std(o, a.x);
  PointIO<T>::store_std(o + cw, a.y);
  return false;
}
bool g1_to_affine_std(const G1Xyzz& p, uint64_t out[8]) { return to_affine_std<FqTag>(p, out); }
bool g2_to_affine_std(const G2Xyzz& p, uint64_t out[16]) { return to_affine_std<Fq2Tag>(p, out); }

G1Xyzz g1_mul_scalar(const G1Xyzz& p, const uint64_t k[4]) {
  uint32_t w[8];
  fr_canon_words(k, w);
  return xyzz_mul_words_w4(p, w);
}
G2Xyzz g2_mul_scalar(const G2Xyzz& p, const uint64_t k[4]) {
  uint32_t w[8];
  fr_canon_words(k, w);
  return xyzz_mul_words_w4(p, w);
}

template <class T>
void HostFixedBase<T>::build(const Affine<T>& p) {
  win.assign(64 * 15, xyzz_inf<T>());
  Xyzz<T> base = xyzz_from_affine(p);
  for (int w = 0; w < 64; ++w) {
    Xyzz<T> run = base;
    for (int d = 1; d <= 15; ++d) {
      win[15 * w + d - 1] = run;
      if (d < 15) xyzz_add(run, base);
    }
    for (int j = 0; j < 4; ++j) xyzz_dbl(base);          // 16^(w+1) * P
  }
}
template <class T>
Xyzz<T> HostFixedBase<T>::mul(const uint64_t k[4]) const {
  uint32_t w[8];
  fr_canon_words(k, w);
  Xyzz<T> acc = xyzz_inf<T>();
  for (int nib = 0; nib < 64; ++nib) {
    const uint32_t v = (w[nib >> 3] >> ((nib & 7) * 4)) & 15u;
    if (v) xyzz_add(acc, win[15 * nib + v - 1]);
  }
  return acc;
}
template struct HostFixedBase<FqTag>;
template struct HostFixedBase<Fq2Tag>;


}  // namespace gs
