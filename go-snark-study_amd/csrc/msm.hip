// MSM engine: host orchestration of the kernels in msm_kernels.h + the O(1) serial combination.
#include "msm.h"

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
  uint32_t chunk = entries >= (1ull << 23) ? 32u : 16u;
  while (chunk < 1024u && entries / std::max<uint32_t>(nbuckets, 1u) > 32ull * chunk) chunk *= 2;
  return chunk;
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
  // R >= 4 bucket ranges: partition by (window, range) first (msm_kernels.h); with two ranges reading the digit matrix twice is
  // cheaper than writing and re-reading 8-byte records
  static const uint32_t part_min_r = (uint32_t)dev_knob("GS_PART_MIN_R", 4, 2, 64);
  const bool wide = pp.R >= part_min_r;
  if (wide && term_index) throw HipError{hipErrorInvalidValue, "plans over a term list need a window width below 19", __LINE__};
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

// Enqueue the kernels that fill `fresh` (W rows of n points) from row 0 on `stream`.
// slab_max: points per launch (2^18 = 1024 workgroups; a background build passes 2^17, see start_background_build).
template <class T>
static void enqueue_table_build(Ctx& c, hipStream_t stream, DevBuf& scratch, const uint32_t* src, size_t n, int cbits, DevBuf& fresh,
                                size_t slab_max = (size_t)1 << 18) {
  constexpr size_t aw = PointIO<T>::kAffineWords;
  const int W = 254 / cbits + 1;
  fresh.alloc(std::max<size_t>(n, 1) * W * aw * 4);
  if (n) {
    static const bool per_row = dev_flag("GS_TABLE_PER_ROW");               // the one-inversion-per-row builder, for comparison
    if (per_row || W <= 2) {
      hipLaunchKernelGGL(k_build_table<T>, grid1(n), dim3(256), 0, stream, src, (uint32_t)n, cbits, W, fresh.as<uint32_t>());
    } else {
      // slabs of 2^18 points: (W - 1) rows of [XYZZ | running product] raw limbs per point (<= 1.4 GiB for G2), reused per slab
      constexpr size_t sw = PointIO<T>::kXyzzWords + PointIO<T>::kXyzzWords / 4;
      const size_t slab = std::min<size_t>(n, slab_max);
      scratch.ensure(slab * (size_t)(W - 1) * sw * 4);
      for (size_t first = 0; first < n; first += slab) {
        const size_t count = std::min(slab, n - first);
        hipLaunchKernelGGL(k_build_table_batched<T>, grid1(count), dim3(256), 0, stream, src, (uint32_t)n, (uint32_t)first, (uint32_t)count, cbits, W,
                           fresh.as<uint32_t>(), scratch.as<uint32_t>());
      }
    }
  }
  GS_HIP(hipGetLastError());
}

template <class T>
static void ensure_table(Ctx& c, BaseTable& t, const uint32_t* row0, size_t n, int cbits) {
  if (t.c == cbits && t.n == n && t.rows.p) return;
  table_settle(c, t, false);                    // a background build of another width: superseded
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
// kTableAfterUses times; then its table is built on the table stream (lowest priority, behind and beside the proofs that keep
// running table-free) and the first call that finds the build complete switches over.
constexpr uint32_t kTableAfterUses = 2;

void table_settle(Ctx& c, BaseTable& t, bool install) {
  if (!t.pending.p) return;
  // this table's own build only: the stream may still be busy with the key's other arrays (a proof that found ONE table complete used
  // to wait here for all five)
  if (t.pending_done) GS_HIP(hipEventSynchronize(t.pending_done));
  else if (c.table_stream) GS_HIP(hipStreamSynchronize(c.table_stream));
  if (install) {
    t.rows = std::move(t.pending);
    t.n = t.pending_n; t.c = t.pending_c; t.W = 254 / t.pending_c + 1;
    t.uses = 0;
  } else t.pending.release();
  t.pending_c = 0; t.pending_n = 0;
}

static void start_background_build(Ctx& c, BaseTable& t, const TableRef& r, int cbits) {
  if (!c.table_stream) {
    int least = 0, greatest = 0;
    GS_HIP(hipDeviceGetStreamPriorityRange(&least, &greatest));
    static const bool low = run_knob("GS_TABLE_STREAM_LOW", 1, 0, 1) != 0;
    GS_HIP(hipStreamCreateWithPriority(&c.table_stream, hipStreamNonBlocking, low ? least : 0));
  }
  if (!t.pending_done) GS_HIP(hipEventCreateWithFlags(&t.pending_done, hipEventDisableTiming));
  DevBuf& scratch = msm_state(c).table_scratch_bg;
  // Points per launch of a background build.  Measured on a fresh 2^20 key, blocking proofs back to back
  // (profiles/r05_background_build_slabs.txt): slabs of 2^18 / 2^17 / 2^16 / 2^15 points -> the tables serve from proof #10 / #13 / #19 /
  // #35 on (245 / 288 / 357 / 580 ms after the key arrived), the slowest proof on the way takes 35 / 28 / 26 / 24 ms (table-free and
  // alone: 10-12).  The table stream has the lowest priority, so smaller slabs mostly starve the build; 2^17 it is.
  static const size_t slab = (size_t)1 << run_knob("GS_TABLE_BG_SLAB_LOG2", 17, 10, 18);      // (same tables whatever the slab)
  try {
    // The slab scratch at its G2 size BEFORE the first launch: the stream's builds share it, and growing it for the fourth array (the
    // G2 one) released the old buffer -- hipFree waits for the device, i.e. for the three G1 builds just enqueued: that, not the
    // builds' share of the chip, was most of the 88 ms a fresh 2^20 key's second proof took (now 28).
    {
      constexpr size_t sw2 = PointIO<Fq2Tag>::kXyzzWords + PointIO<Fq2Tag>::kXyzzWords / 4;
      const int W = 254 / cbits + 1;
      if (W > 2) scratch.ensure(std::min<size_t>(std::max<size_t>(r.n, 1), slab) * (size_t)(W - 1) * sw2 * 4);
    }
    if (r.g2) enqueue_table_build<Fq2Tag>(c, c.table_stream, scratch, r.row0, r.n, cbits, t.pending, slab);
    else enqueue_table_build<FqTag>(c, c.table_stream, scratch, r.row0, r.n, cbits, t.pending, slab);
  } catch (const HipError& e) {
    if (e.e != hipErrorOutOfMemory) throw;      // no room for a table: keep summing table-free
    t.pending.release();
    return;
  }
  t.pending_c = cbits; t.pending_n = r.n;
  GS_HIP(hipEventRecord(t.pending_done, c.table_stream));
}

bool prepare_tables(Ctx& c, const std::vector<TableRef>& group, uint32_t nterms, int* cbits) {
  const int cb = choose_window_bits(std::max<uint32_t>(nterms, 1u), c.window_bits);
  bool all_ready = true;
  for (const TableRef& r : group) {
    BaseTable& t = *r.t;
    t.last_use = c.call_clock;
    if (t.pending.p && hipEventQuery(t.pending_done) == hipSuccess) {          // a background build came through
      if (t.pending_c == cb && t.pending_n == r.n) table_settle(c, t, true); else table_settle(c, t, false);
    }
    (void)hipGetLastError();                                                   // hipErrorNotReady is not an error
    all_ready = all_ready && t.ready(r.n, cb);
  }
  if (all_ready) { *cbits = cb; return true; }
  if (c.table_policy == 1) {                                                   // always: inside the call, as rounds 1-4 did
    for (const TableRef& r : group) {
      if (r.g2) ensure_table_g2(c, *r.t, r.row0, r.n, cb); else ensure_table_g1(c, *r.t, r.row0, r.n, cb);
    }
    *cbits = cb;
    return true;
  }
  if (c.table_policy == 0) {
    for (const TableRef& r : group) {
      BaseTable& t = *r.t;
      if (t.ready(r.n, cb)) continue;
      t.uses += 1;
      if (t.uses >= kTableAfterUses && !t.pending.p && r.n) start_background_build(c, t, r, cb);
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
  hipLaunchKernelGGL(k_bucket_accumulate<T>, dim3((plan.maxchunks + 255) / 256, njobs), dim3(256), 0, c.stream,
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
  msm_enqueue<FqTag>(c, plan, bases, 8 * Ctx::kBlockingSlot, 3 * Ctx::kBlockingSlot, p, nullptr);
  GS_HIP(hipStreamSynchronize(c.stream));
  msm_book_timing(c, p);
  msm_finish<FqTag>(c, p, out);
}
void msm_run_g2(Ctx& c, const MsmPlan& plan, const std::vector<MsmBase>& bases, std::vector<G2Xyzz>& out) {
  MsmPending p;
  msm_enqueue<Fq2Tag>(c, plan, bases, 8 * Ctx::kBlockingSlot + 4, 3 * Ctx::kBlockingSlot, p, nullptr);
  GS_HIP(hipStreamSynchronize(c.stream));
  msm_book_timing(c, p);
  msm_finish<Fq2Tag>(c, p, out);
}

// Returns the number of points that are not on their curve (and the first such index); synchronises the stream.
template <class T>
static uint32_t jacobian_to_affine_checked(Ctx& c, const uint32_t* jac, uint32_t n, uint32_t* out, uint32_t* first_bad) {
  if (!n) return 0;
  DevBuf flag(8);
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
