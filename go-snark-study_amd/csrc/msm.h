// Internal C++ interface of the MSM engine (implemented in msm.hip).
#pragma once
#include <algorithm>
#include <vector>

#include "ec.h"
#include "runtime.h"

namespace gs {

struct PlanBuffers {             // grow-only device workspaces owned by a plan slot
  DevBuf digits, hist, totals, offsets, entries, tiles, total;
  DevBuf chunk_bucket, heavy_list, counters;
  DevBuf recs, parts;              // wide windows: (term, bucket) records partitioned by (window, range); counts | bases | cursors
};

struct MsmPlan {                 // digits of one scalar vector, bucket-sorted (device resident)
  uint32_t n = 0;
  int c = 0, W = 0;
  uint32_t B = 0;                // buckets per window
  // Bucket sets.  Window-table route: ONE set of B buckets for all windows (digit d of window j adds row j of the table, i.e.
  // 2^(c j) P_i, into bucket d).  Table-free route: a set per window, bucket id = window * B + (|d| - 1), every window adds the
  // base point itself; the window sums are recombined by Horner on the host.  nbuckets = B or W * B accordingly.
  bool table_free = false;
  uint32_t nbuckets = 0;
  uint32_t chunk = 32;           // entries per accumulate thread (16 or 32; always a multiple of 4)
  uint32_t maxchunks = 0;        // upper bound of the number of chunks (the exact count stays on the device)
  const uint32_t* offsets = nullptr;        // nbuckets + 1 (offsets[nbuckets] = number of entries)
  const uint32_t* entries = nullptr;
  const uint32_t* chunk_bucket = nullptr;   // maxchunks
  const uint32_t* heavy_list = nullptr;
  const uint32_t* heavy_count = nullptr;
};

constexpr int kMaxWindowBits = 20;
int choose_window_bits(uint32_t n, int forced);
// the table-free route's width: every window owns a bucket set, so the optimum is narrower; 9 .. 16 (one LDS histogram range, and
// a reduce workgroup never straddles two windows)
constexpr int kMinFreeWindowBits = 9, kMaxFreeWindowBits = 16;
int choose_window_bits_free(uint32_t n, int forced);

// Window table of a base array: rows[j][i] = 2^(c j) * P_i (packed affine), j < W = 254 / c + 1.
struct BaseTable {
  DevBuf rows;
  size_t n = 0;
  int c = 0, W = 0;
  uint64_t last_use = 0;           // Ctx::call_clock of the last call that used the array (LRU order of evict_tables_for)
  uint64_t last_table_use = 0;     // ... of the last call that summed it over its TABLE: under policy auto a table that serves is not
                                   //     replaced by one of another width just because one call wanted that width
  uint32_t uses = 0;               // proofs / MSMs that found no table since it last had one (table policy auto)
  // A build under policy auto (round 6: in instalments).  `pending` holds the rows of the table to be; every call that finds the
  // array without its table enqueues the next slabs of points [pending_next, ...) on the accumulation stream, as many as its build
  // credit pays for (msm.hip, prepare_tables), and the call that enqueues the last slab installs the table: its accumulation kernel
  // runs behind the slabs on the same stream.
  DevBuf pending;
  int pending_c = 0;
  size_t pending_n = 0, pending_next = 0;
  const uint32_t* pending_src = nullptr;      // row 0 (the base array: lives as long as the object that owns this table)
  bool pending_g2 = false;
  hipEvent_t pending_done = nullptr;          // recorded behind the last slab
  bool pending_complete() const { return pending.p != nullptr && pending_next >= pending_n; }
  bool ready(size_t n_, int c_) const { return rows.p != nullptr && n == n_ && c == c_; }
  void drop() { rows.release(); n = 0; c = 0; W = 0; }
  BaseTable() = default;
  BaseTable(const BaseTable&) = delete;
  BaseTable& operator=(const BaseTable&) = delete;
  ~BaseTable() { if (pending_done && !process_exiting()) (void)hipEventDestroy(pending_done); }
};
// (Re)build `t` for window width c from row 0 (`row0` = packed affine points; pass nullptr to rebuild from the
// table's own row 0).  No-op when the table already matches.
void ensure_table_g1(Ctx& c, BaseTable& t, const uint32_t* row0, size_t n, int cbits);
void ensure_table_g2(Ctx& c, BaseTable& t, const uint32_t* row0, size_t n, int cbits);

// The base arrays one plan will be multiplied with (a proof's At / BACGamma / BACDelta / BACGamma2 over w; one array for an MSM).
struct TableRef { BaseTable* t; const uint32_t* row0; size_t n; bool g2; };
// Decides how the group is summed THIS time and prepares it: true = window tables, all of width *cbits and resident (built now
// under policy `always`, or found); false = table-free with *cbits from choose_window_bits_free.  Under policy `auto` an array that
// has been used twice gets its table IN INSTALMENTS: every call enqueues, in front of its own accumulations, as many slabs of the
// pending table as `*credit` pays for (in G1-point builds; a G2 point costs kG2BuildCost of them; the caller grants ~0.055 points per
// (job-unit x term) of its own work, i.e. a proof pays ~60 % of its own table-free time on top) and takes what it spent off
// *credit.  Stamps the tables for the LRU.
constexpr double kG2BuildCost = 2.3;              // a G2 row costs 2.3 G1 rows (40 vs 17.5 ms per 2^20 points)
constexpr double kBuildCreditPerUnitTerm = 0.055; // 17.5 ms per 2^20 G1 points built (Jacobian doublings) / 1.5 ms per 2^20 (job-unit x term) summed table-free: 0.055 -> +60 % (6.5 ms per 2^20 proof)
// The credit a call of `units` job-units (a G1 sum = 1, a G2 sum = 2.76) over n terms grants itself; never less than 2^18 G1 points
// (~5 ms of building): the tables of a 2^16 key cost 9 ms in all while its table-free proofs take 5.5 ms instead of 2, so small keys are
// warm after two or three calls instead of sixteen.
inline double build_credit(double units, size_t n) { return std::max(kBuildCreditPerUnitTerm * units * (double)n, (double)((size_t)1 << 18)); }
bool prepare_tables(Ctx& c, const std::vector<TableRef>& group, uint32_t nterms, int* cbits, double* credit = nullptr);
// the tables one call will use, stamped for the LRU BEFORE any of them is built: an allocation made while the first group's tables are
// built must not evict the second group's (prepare_tables only stamped the group it was called for)
void stamp_tables(Ctx& c, std::initializer_list<BaseTable*> tables);
// wait for a pending build and install it (finishing it first if instalments are missing) or drop it (gs_release_tables, gs_build_tables)
void table_settle(Ctx& c, BaseTable& t, bool install);

struct MsmBase {                 // one job of an MSM launch
  const BaseTable* table;        // window-table route: the table ...
  size_t off;                    // ... and the first term's offset in it (or in `points`)
  const uint32_t* points = nullptr;   // table-free route: the base array itself (packed affine), `npoints` of them
  size_t npoints = 0;
};

// A launch that will consume a plan: how many base arrays it sums at once and whether they are G2.
struct LaunchShape { int njobs; bool g2; };
// scalars_dev: n x 8 u32 words (standard form, any 256-bit value).  slot: 0..3 (four plans may be alive).  `users`: the
// launches that will run on this plan (decides the chunk size: whole wave rounds for every one of them).
// cbits: the width prepare_tables chose; table_free: its route.
// term_index / index_bias: the plan's n terms are terms term_index[i] - index_bias of the scalar vector and of the base arrays
// (see k_digits); window widths below 19 only.
void build_plan(Ctx& c, int slot, const uint32_t* scalars_dev, uint32_t n, MsmPlan& plan, const std::vector<LaunchShape>& users, int cbits = 0,
                bool table_free = false, const uint32_t* term_index = nullptr, uint32_t index_bias = 0);
// true when build_plan sorts a plan of this width partition-first (>= GS_PART_MIN_R bucket ranges of 2^15, i.e. c >= 18 as shipped):
// that sort takes no term list, so callers that would hand build_plan one (the sparse-B plans of prove.hip) must ask first
bool plan_partitions_first(int cbits);
// which points of one or two packed-affine arrays are finite: mask (ceil(n / 32) words, written) and their number (synchronises the stream)
uint32_t finite_mask_dev(Ctx& c, const uint32_t* g1_pts, const uint32_t* g2_pts, uint32_t n, uint32_t* mask_dev);

// One launch sequence for up to 8 base arrays sharing a plan (their tables must have been built for plan.c).
// msm_enqueue_* only ENQUEUES on c.stream (kernels + the async download of the <= 16 workgroup pairs per
// job into pinned slot `slot`), so several MSM groups can be in flight on different streams; after that
// stream has been synchronised, msm_finish_* adds the pairs on the host core and books the timings.
// ws_base: first of the 8 workspace sets to use (groups in flight together must not share sets).
struct MsmPending {
  int njobs = 0, L = 1, slot = 0, c = 0, W = 0;
  bool table_free = false;         // per-window bucket sets: the pairs come grouped by window, the host recombines by Horner
  uint32_t nblk_window = 0;        // reduce workgroups per window on that route
  uint32_t nblk = 0, n = 0;
  bool g2 = false;
  bool folded = false;             // the device folded the workgroup pairs: one XYZZ point per job in the pinned slot
  const void* pinned_slot = nullptr;   // where this group's results land; the plan's statistics follow them at stats_off
  size_t stats_off = 0;
  std::shared_ptr<PhaseTimer> tacc, tker, tred;
};
// tail_stream: where the window merge / reduction / download go (nullptr = c.stream)
void msm_enqueue_g1(Ctx& c, const MsmPlan& plan, const std::vector<MsmBase>& bases, int ws_base, int slot, MsmPending& p, hipStream_t tail_stream = nullptr);
void msm_enqueue_g2(Ctx& c, const MsmPlan& plan, const std::vector<MsmBase>& bases, int ws_base, int slot, MsmPending& p, hipStream_t tail_stream = nullptr);
void msm_book_timing(Ctx& c, const MsmPending& p);        // HIP event queries: call on the thread that drives the device
void msm_finish_g1(Ctx& c, const MsmPending& p, std::vector<G1Xyzz>& out);   // pure host arithmetic (worker threads allowed)
void msm_finish_g2(Ctx& c, const MsmPending& p, std::vector<G2Xyzz>& out);
// enqueue + synchronise + finish on c.stream
void msm_run_g1(Ctx& c, const MsmPlan& plan, const std::vector<MsmBase>& bases, std::vector<G1Xyzz>& out);
void msm_run_g2(Ctx& c, const MsmPlan& plan, const std::vector<MsmBase>& bases, std::vector<G2Xyzz>& out);

// base-array helpers (device)
// -> number of points off their curve (first such index in *first_bad); synchronises the stream
uint32_t jacobian_to_affine_g1(Ctx& c, const uint32_t* jac_dev, uint32_t n, uint32_t* out_dev, uint32_t* first_bad = nullptr);
uint32_t jacobian_to_affine_g2(Ctx& c, const uint32_t* jac_dev, uint32_t n, uint32_t* out_dev, uint32_t* first_bad = nullptr);
// the same from caller memory (gs_g1_upload / gs_g2_upload): staged through pinned buffers into a scratch the context keeps
uint32_t upload_jacobian_g1(Ctx& c, const uint64_t* jac_host, uint32_t n, uint32_t* out_dev, uint32_t* first_bad = nullptr);
uint32_t upload_jacobian_g2(Ctx& c, const uint64_t* jac_host, uint32_t n, uint32_t* out_dev, uint32_t* first_bad = nullptr);
void affine_to_jacobian_std_g1(Ctx& c, const uint32_t* aff_dev, uint32_t n, uint32_t* out_dev);
void affine_to_jacobian_std_g2(Ctx& c, const uint32_t* aff_dev, uint32_t n, uint32_t* out_dev);
void fixed_base_g1(Ctx& c, const uint32_t* scalars_dev, uint32_t n, uint32_t* out_dev);
void fixed_base_g2(Ctx& c, const uint32_t* scalars_dev, uint32_t n, uint32_t* out_dev);

// ---- host-side serial helpers (same __host__ __device__ arithmetic as the kernels) ----------------
// standard-form words <-> Montgomery host values
G1Affine g1_affine_from_jacobian_std(const uint64_t jac[12]);
G2Affine g2_affine_from_jacobian_std(const uint64_t jac[24]);
// XYZZ -> affine standard words ([x,y] = 8 words / [x0,x1,y0,y1] = 16 words); returns is_inf
bool g1_to_affine_std(const G1Xyzz& p, uint64_t out[8]);
bool g2_to_affine_std(const G2Xyzz& p, uint64_t out[16]);
// k * P with a 256-bit scalar in ABI words (reduced mod r first)
G1Xyzz g1_mul_scalar(const G1Xyzz& p, const uint64_t k[4]);
G2Xyzz g2_mul_scalar(const G2Xyzz& p, const uint64_t k[4]);
// Host-side fixed-base multiplication for points that stay the same for the life of a key (delta, delta2 of groth16.Pk): a table
// of d * 16^w * P (w < 64, d = 1..15; 960 points, built once in ~1.5 ms for G1 / ~5 ms for G2) turns k * P into at most 64 additions
// instead of 256 doublings + 64 additions -- the prover tail has four such products per proof (groth16.go:254-264, 274-275).
template <class T>
struct HostFixedBase {
  std::vector<Xyzz<T>> win;        // win[15 w + d - 1] = d * 16^w * P
  void build(const Affine<T>& p);
  Xyzz<T> mul(const uint64_t k[4]) const;
};
extern template struct HostFixedBase<FqTag>;
extern template struct HostFixedBase<Fq2Tag>;

void fr_canon_words(const uint64_t k[4], uint32_t out[8]);

}  // namespace gs
