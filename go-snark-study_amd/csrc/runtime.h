// Host runtime of libgosnark_hip.so: device context, error reporting, handle table, workspace pool.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <atomic>
#include <condition_variable>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <mutex>
#include <string>
#include <unordered_map>
#include <vector>

#include "../../include/gosnark_hip.h"
#include "knobs.h"

namespace gs {

// ---- errors -------------------------------------------------------------------------------------
inline std::string& last_error_ref() {
  static thread_local std::string e;
  return e;
}
inline int fail(int code, const char* fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof buf, fmt, ap);
  va_end(ap);
  last_error_ref() = buf;
  return code;
}
struct HipError {
  hipError_t e;
  const char* what;
  int line;
};
#define GS_HIP(x)                                              \
  do {                                                         \
    hipError_t _e = (x);                                       \
    if (_e != hipSuccess) throw ::gs::HipError{_e, #x, __LINE__}; \
  } while (0)

// set by an atexit handler registered in gs_init: static DevBuf destructors that run during process teardown must
// not call into a HIP runtime that may already be gone (seen as a crash in __cxa_finalize under rocprofv3)
inline bool& process_exiting() {
  static bool v = false;
  return v;
}

// ---- device buffers -------------------------------------------------------------------------------
// every device byte this library holds (all contexts): gs_memory_query reports it
inline std::atomic<uint64_t>& devbuf_bytes() {
  static std::atomic<uint64_t> v{0};
  return v;
}

// Allocation bookkeeping (gs_alloc_counters): how often the library went to hipMalloc / hipFree.  The steady state of a prover that
// streams proofs -- resident or host-buffer tickets -- performs neither (tests/test_gpu_stream_host.py).
inline std::atomic<uint64_t>& devbuf_allocs() { static std::atomic<uint64_t> v{0}; return v; }
inline std::atomic<uint64_t>& devbuf_frees() { static std::atomic<uint64_t> v{0}; return v; }
// gs_set_memory_limit: a cap on devbuf_bytes() (0 = none) -- a development / test hook that makes "out of device memory" reachable
// without filling 288 GB; an allocation that would exceed it is treated exactly like hipErrorOutOfMemory.
inline std::atomic<uint64_t>& devbuf_limit() { static std::atomic<uint64_t> v{0}; return v; }
// Called when hipMalloc reports out of memory (or the cap is hit): drop least-recently-used window tables of the calling thread's
// context that no ticket and not the running call hold, until `need` bytes were freed or nothing is left (capi_mem.hip).
bool evict_tables_for(size_t need);

inline void* dev_malloc(size_t n) {
  for (int attempt = 0;; ++attempt) {
    void* q = nullptr;
    const uint64_t cap = devbuf_limit().load();
    hipError_t e = (cap && devbuf_bytes().load() + n > cap) ? hipErrorOutOfMemory : hipMalloc(&q, n);
    if (e == hipSuccess) { devbuf_allocs() += 1; return q; }
    (void)hipGetLastError();
    if (e != hipErrorOutOfMemory || attempt > 0 || !evict_tables_for(n)) throw HipError{e, "hipMalloc", __LINE__};
  }
}

struct DevBuf {
  void* p = nullptr;
  size_t bytes = 0;
  DevBuf() = default;
  explicit DevBuf(size_t n) { alloc(n); }
  DevBuf(const DevBuf&) = delete;
  DevBuf& operator=(const DevBuf&) = delete;
  DevBuf(DevBuf&& o) noexcept : p(o.p), bytes(o.bytes) { o.p = nullptr; o.bytes = 0; }
  DevBuf& operator=(DevBuf&& o) noexcept {
    if (this != &o) { release(); p = o.p; bytes = o.bytes; o.p = nullptr; o.bytes = 0; }
    return *this;
  }
  ~DevBuf() { release(); }
  // The new buffer exists before the old one goes (a failed allocation leaves the object as it was).  Only when even eviction
  // cannot make room for both is the old one given up first -- its contents are scratch for every caller of alloc / ensure.
  void alloc(size_t n) {
    if (n == 0) n = 16;
    void* q = nullptr;
    try { q = dev_malloc(n); }
    catch (const HipError&) {
      if (!p) throw;
      release();
      q = dev_malloc(n);
    }
    release();
    p = q;
    bytes = n;
    devbuf_bytes() += n;
  }
  void ensure(size_t n) { if (n > bytes) alloc(n + n / 8); }      // grow-only workspace
  void release() {
    if (p) {
      if (!process_exiting()) { (void)hipFree(p); devbuf_frees() += 1; }
      devbuf_bytes() -= bytes; p = nullptr; bytes = 0;
    }
  }
  template <class U> U* as() const { return reinterpret_cast<U*>(p); }
};

// ---- handle table ---------------------------------------------------------------------------------
enum class Kind : uint32_t { G1Bases = 1, G2Bases, Scalars, GrothPk, PinocchioPk, R1cs };

struct Object {
  Kind kind;
  virtual ~Object() = default;
  explicit Object(Kind k) : kind(k) {}
};

struct Bases : Object {          // packed affine Montgomery points, resident (+ their lazily built window table)
  DevBuf buf;
  size_t n = 0;
  std::shared_ptr<void> table;   // gs::BaseTable (msm.h), created on first MSM use
  explicit Bases(Kind k) : Object(k) {}
};
struct R1csObj : Object {        // sparse R1CS resident on the device: A, B, C in CSR (values standard form) + per-proof workspaces
  size_t n = 0, m = 0, nnz[3] = {0, 0, 0};
  DevBuf rowptr[3], col[3], val[3];
  DevBuf w_mont, vals, coef, prod;       // grow-once workspaces of gs_r1cs_px
  R1csObj() : Object(Kind::R1cs) {}
};
struct Scalars : Object {        // n x 8 u32 words, standard form, resident
  DevBuf buf;
  size_t n = 0;
  // gs_scalars_update overwrites the vector IN PLACE, so it must come after every device read enqueued before it.  A pipelined
  // operation that reads the vector leaves, per stream it read it on, an event behind its last reader (in-order streams: a later
  // record on the same stream supersedes the earlier one); the update makes its copy wait for them.  Blocking entry points have
  // finished reading when they return and leave nothing.
  struct ReadMark { hipStream_t stream; hipEvent_t ev; };
  std::vector<ReadMark> reads;
  void mark_read(hipStream_t s) {
    for (auto& r : reads) if (r.stream == s) { GS_HIP(hipEventRecord(r.ev, s)); return; }
    hipEvent_t ev;
    GS_HIP(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
    GS_HIP(hipEventRecord(ev, s));
    reads.push_back(ReadMark{s, ev});
  }
  Scalars() : Object(Kind::Scalars) {}
  ~Scalars() override { if (!process_exiting()) for (auto& r : reads) (void)hipEventDestroy(r.ev); }
};

// a proof whose device work has been enqueued but not collected yet (prove.hip)
struct InFlightBase {
  uint64_t ticket = 0;
  std::vector<std::shared_ptr<Object>> keep;   // the key and scalar vectors this operation reads: gs_free on them is deferred
  // Block until the operation's device work is through (its completion events; HIP event waits are thread-safe).  Called WITHOUT the
  // context's lock by wait_ticket_unlocked below, so that a thread that collects a ticket does not keep every other thread's _begin
  // (and the _end of finished tickets) out for the ~8 ms a 2^20 proof still has to run (round 6).
  virtual void wait_device() const {}
  virtual ~InFlightBase() = default;
};

// Handles carry the logical device they live on in their top byte (logical device 0 handles are the small integers they
// always were); every context owns its objects, streams, workspaces and lock.
constexpr int kHandleDevShift = 56;
constexpr int kMaxLogicalDevices = 64;
inline int handle_device(gs_handle h) { return (int)(h >> kHandleDevShift); }

// Handle and ticket numbers are unique for the LIFETIME OF THE PROCESS, not of a gs_init session: after gs_shutdown + gs_init a
// stale handle of the earlier session (a Python object's destructor, a cached Go key) must fail with GS_ERR_ARG, never alias -- and
// gs_free -- an object of the new session.
inline std::atomic<uint64_t>& handle_counter() { static std::atomic<uint64_t> v{1}; return v; }
inline std::atomic<uint64_t>& ticket_counter() { static std::atomic<uint64_t> v{1}; return v; }

// The context's lock: first come, first served.  std::mutex is not fair -- the thread that releases it and asks again at once beats
// every waiter that first has to be woken -- and entry points hold a context for as long as a blocking proof: two threads calling
// gs_groth16_prove in a loop locked three pipelined producers, a canceller and two bystanders out for a whole minute (round 6,
// tests/c/stream_stress.c: 27 434 blocking proofs against ONE operation of every other thread).  Tickets are served in the order of
// arrival; uncontended it costs two uncontended std::mutex round trips.
class FairMutex {
 public:
  void lock() {
    std::unique_lock<std::mutex> lk(m_);
    const uint64_t mine = next_++;
    while (mine != serving_) cv_.wait(lk);
  }
  bool try_lock() {
    std::lock_guard<std::mutex> lk(m_);
    if (next_ != serving_) return false;
    next_ += 1;
    return true;
  }
  void unlock() {
    bool waiters;
    { std::lock_guard<std::mutex> lk(m_); serving_ += 1; waiters = next_ != serving_; }
    if (waiters) cv_.notify_all();
  }
 private:
  std::mutex m_;
  std::condition_variable cv_;
  uint64_t next_ = 0, serving_ = 0;
};

struct Ctx {
  int logical = 0;                // index in gs_init's device list (several entries may name the same physical device)
  int device = -1;                // HIP ordinal
  bool ready = false;
  hipStream_t stream = nullptr;   // the stream every engine function enqueues on (switched by StreamScope)
  hipStream_t main_stream = nullptr;
  // 0 / 2: reduction tails, 1: plans and H(x).  (A fourth stream for plan(w) was tried and LOST 5 %: beyond four streams two of
  // them share a hardware queue, and the sort of the next proof then queues behind a reduction tail.)
  static constexpr int kAuxStreams = 3;
  hipStream_t aux_stream[kAuxStreams] = {nullptr, nullptr, nullptr};
#ifndef GS_MAX_IN_FLIGHT
#define GS_MAX_IN_FLIGHT 3                           // (a build-time constant: profiles/r06_ab_four_in_flight.txt measured 4)
#endif
  static constexpr int kMaxInFlight = GS_MAX_IN_FLIGHT;   // pipelined operations (tickets); each owns one set of workspaces
  static constexpr int kSlots = kMaxInFlight + 1;    // + one set for the blocking entry points (serialised by `mu`), so a blocking
  static constexpr int kBlockingSlot = kMaxInFlight; //   call made while tickets are outstanding never touches their result staging
  void* pinned[4 * kSlots] = {};                     // host staging of the result downloads: 3 per slot, then one more per slot
                                                     //   (the B1 group of a proof whose B sums run over their own plan)
  // violated-constraint counters of the witness routes, one word per slot: device words + their pinned host copies (an async copy
  // behind the check kernel writes them; the collector reads them).  Context-owned, so they outlive gs_trim while tickets are out.
  DevBuf bad_dev;
  uint32_t* bad_host = nullptr;
  static constexpr int kStageBuffers = 4;            // pinned staging of uploads from pageable caller memory (hostcopy.h), lazy: at most 4,
  static constexpr size_t kStageBytes = 64u << 20;   // of at most 64 MiB (hostcopy.h decides how many and how large)
  void* stage[kStageBuffers] = {};
  hipEvent_t stage_ev[kStageBuffers] = {};
  hipStream_t copy_stream = nullptr;                 // gs_scalars_upload (lazy)
  std::shared_ptr<InFlightBase> inflight[kMaxInFlight];   // shared: a collector waits for the device on its own reference, outside the lock
  // Consecutive pipelined operations swap the two tail streams: the reduction tails are chains of dependent point additions
  // (latency, not throughput), so the tails of operation k + 1 may run beside those of operation k instead of queueing behind
  // them -- at 2^16 the G2 tail (1.2 ms) was longer than the accumulations of a whole proof (0.9 ms) and set the pace.
  unsigned tail_flip = 0;
  // (Which stream carries plan(w) and which the tails is decided once per proof by prove.hip's proof_streams(); the context only
  //  keeps the alternation state of the two tail streams.)
  hipStream_t tail_stream(int which) { return aux_stream[((which ^ (int)(tail_flip & 1u)) & 1) ? 2 : 0]; }
  void next_tails(uint32_t n) {                     // called once per pipelined operation of n terms
    static const int mode = (int)run_knob("GS_TAIL_FLIP", 1, 0, 2);                        // 0 never, 1 always, 2 by size (same results)
    if (mode == 1 || (mode == 2 && n <= kTailFlipMaxTerms)) tail_flip ^= 1u;
  }
  static constexpr uint32_t kTailFlipMaxTerms = 1u << 18;
  uint64_t new_ticket() { return ((uint64_t)logical << kHandleDevShift) | ticket_counter()++; }
  int free_parity() const { for (int p = 0; p < kMaxInFlight; ++p) if (!inflight[p]) return p; return -1; }
  bool any_inflight() const { for (int p = 0; p < kMaxInFlight; ++p) if (inflight[p]) return true; return false; }
  // The slot a BLOCKING entry point works in (round 6): a ticket slot that is free -- the call holds the context's lock
  // until it returns, so no ticket can take the slot under it, and a free slot has no device work or uncollected result -- and only
  // with three tickets outstanding the fourth set of workspaces.  A caller that mixes tickets and blocking calls used to hold four
  // sets of bucket / partial workspaces (17 GB each at 2^24 constraints); now the fourth exists only if it is ever needed.
  int blocking_slot() const { const int p = free_parity(); return p >= 0 ? p : kBlockingSlot; }
  static constexpr size_t kPinnedBytes = 256 * 1024;
  FairMutex mu;
  std::unordered_map<uint64_t, std::shared_ptr<Object>> objs;     // in-flight operations hold references: gs_free defers
  uint64_t call_clock = 0;       // one tick per entry-point call on this context: window tables stamp it when a call uses them (LRU
                                 // order for evict_tables_for; a table stamped with the running call's tick is never its victim)
  uint64_t evictions = 0;        // window tables dropped by evict_tables_for so far
  int window_bits = 0;           // 0 = auto
  // gs_set_table_policy -- when a base array gets its 15-row window table (msm.h, prepare_tables): 0 auto (sum table-free until the
  // array has been used twice, then build in the background and switch when the build is through), 1 always (build on first use,
  // inside the call: rounds 1-4), 2 never
  int table_policy = 0;
  hipStream_t table_stream = nullptr;   // (round 5: background table builds; round 6 builds in instalments on the main stream, msm.hip)
  double build_balance = 0;             // build credit a call overdrew (less than one slab): taken off the next call's (msm.hip, prepare_tables)
  bool eval_basis = true;        // gs_set_eval_basis: witness route over H's values when the key has an evaluation-basis array
  gs_timing timing{};
  std::mutex timing_mu;          // msm_finish of several groups may run on different host threads
  // reusable workspaces (grow-only)
  static constexpr int kWsSets = 8 * kSlots;         // 8 workspace sets per slot
  DevBuf ws_buckets[kWsSets], ws_chunks[kWsSets], ws_partials[kWsSets], ws_out[kWsSets];
  // per-context state of the engines (plan buffers, NTT twiddles, node trees, staging buffers): device memory belongs to
  // one device, so nothing of this may be a process-wide static
  std::shared_ptr<void> msm_state, poly_state, prove_state;
  DevBuf ws_misc;
  DevBuf g1_pow2, g2_pow2;       // fixed-base window tables d * 2^(8 w) * G, 32 x 256 entries (lazy)
  std::vector<hipEvent_t> events;

  template <class O> O* get(gs_handle h, Kind k) {
    auto it = objs.find(h);
    if (it == objs.end() || it->second->kind != k) return nullptr;
    return static_cast<O*>(it->second.get());
  }
  template <class O> std::shared_ptr<O> share(gs_handle h, Kind k) {
    auto it = objs.find(h);
    if (it == objs.end() || it->second->kind != k) return nullptr;
    return std::static_pointer_cast<O>(it->second);
  }
  gs_handle put(std::shared_ptr<Object> o) {
    uint64_t h = ((uint64_t)logical << kHandleDevShift) | handle_counter()++;
    objs[h] = std::move(o);
    return h;
  }
  // lazily created engine state bags
  template <class S> S& state(std::shared_ptr<void>& slot) {
    if (!slot) slot = std::make_shared<S>();
    return *static_cast<S*>(slot.get());
  }
  // wait for everything enqueued on this context (outstanding tickets keep their results in their pinned slots)
  void drain() {
    if (main_stream) GS_HIP(hipStreamSynchronize(main_stream));
    for (auto a : aux_stream) if (a && a != main_stream) GS_HIP(hipStreamSynchronize(a));
  }
};

// ---- the contexts of this process: one per entry of gs_init's device list ------------------------------------------
// `mu` guards the vector (gs_init / gs_shutdown write it).  Entry points copy the shared_ptr of their context under `mu` and hold it
// for the duration of the call, so a concurrent gs_shutdown can neither free a context under a running call nor race with the
// lookup: the call then finds ready = false under the context's own lock.  Lock order: Registry::mu before Ctx::mu, and never
// Registry::mu while holding a Ctx::mu.
struct Registry {
  std::mutex mu;
  std::vector<std::shared_ptr<Ctx>> ctxs;
};
inline Registry& registry() {
  static Registry r;
  return r;
}
// the logical device the calling host thread creates objects on (gs_set_device; like hipSetDevice, per thread)
inline int& current_logical() {
  static thread_local int v = 0;
  return v;
}
// the context of a logical device, or null; *count (optional) = number of logical devices
inline std::shared_ptr<Ctx> ctx_ref(int logical, size_t* count = nullptr) {
  Registry& r = registry();
  std::lock_guard<std::mutex> lk(r.mu);
  if (count) *count = r.ctxs.size();
  if (logical < 0 || (size_t)logical >= r.ctxs.size()) return nullptr;
  return r.ctxs[logical];
}
inline size_t logical_device_count() {
  Registry& r = registry();
  std::lock_guard<std::mutex> lk(r.mu);
  return r.ctxs.size();
}

// Every entry point: pick the context (the one `route` lives on when a handle is given, else the calling thread's current
// logical device), lock it, check init, translate exceptions into status codes.
// allow_inflight = false: the call uses workspaces that outstanding tickets also use, so it first waits for their device work
// (it QUEUES behind them; their results stay in their own pinned slots until gs_*_end collects them).
// What an entry point sees before gs_init (ready = false; only its lock is used).  ONE object for every instantiation of guarded<F>:
// as a function-local static of the template each of the ~100 lambdas had a full Ctx of its own, and calls made before gs_init
// locked different mutexes depending on the entry point.
inline Ctx& none_ctx() {
  static Ctx none;
  return none;
}

// the context the calling host thread is inside of (set by guarded / guarded_pair for the duration of the call): what
// evict_tables_for() may take tables from -- its lock is held, so its handle table and tickets cannot change under the evictor
inline Ctx*& current_ctx() {
  static thread_local Ctx* c = nullptr;
  return c;
}
struct CurrentCtxScope {
  Ctx* saved;
  explicit CurrentCtxScope(Ctx* c) : saved(current_ctx()) { current_ctx() = c; }
  ~CurrentCtxScope() { current_ctx() = saved; }
};

template <class F>
int guarded(F&& f, bool need_init = true, bool allow_inflight = false, gs_handle route = 0) {
  size_t ndev = 0;
  const int logical = route ? handle_device(route) : current_logical();
  std::shared_ptr<Ctx> pc = ctx_ref(logical, &ndev);
  Ctx& c = pc ? *pc : none_ctx();
  std::lock_guard<FairMutex> lk(c.mu);
  if (need_init && !c.ready) {
    if (ndev == 0) return fail(GS_ERR_NOT_INIT, "gs_init has not been called (or failed)");
    return fail(GS_ERR_ARG, "no logical device %d (gs_init listed %zu)", logical, ndev);
  }
  // the HIP current device is per host thread: callers (goroutines, worker threads) may arrive on any thread
  if (c.ready) (void)hipSetDevice(c.device);
  CurrentCtxScope scope(c.ready ? &c : nullptr);
  c.call_clock += 1;
  try {
    if (need_init && !allow_inflight && c.any_inflight()) c.drain();
    return f(c);
  } catch (const HipError& e) {
    return fail(GS_ERR_HIP, "HIP error %d (%s) at %s line %d", (int)e.e, hipGetErrorString(e.e), e.what, e.line);
  } catch (const std::bad_alloc&) {
    return fail(GS_ERR_HIP, "host allocation failed");
  } catch (const std::exception& e) {
    return fail(GS_ERR_ARG, "%s", e.what());
  }
}
inline void reset_timing(Ctx& c) { c.timing = gs_timing{}; }

// First phase of every gs_*_end: wait for the ticket's device work WITHOUT holding the context's lock (the second phase, under the
// lock as before, then finds every event complete and only folds the downloaded partial sums).  Round 5's _end waited for the device
// inside the lock: with two producer threads on one device (tests/c/stream_producer.c) one of them spent the whole run locked out --
// std::mutex is not fair, and the thread that held it for 8 of every 10 ms won every race for it -- and a goroutine collecting one
// ticket kept all others from submitting.  An unknown ticket is left for the locked phase to report.
inline void wait_ticket_unlocked(uint64_t ticket) {
  std::shared_ptr<Ctx> pc = ctx_ref(handle_device(ticket));
  if (!pc) return;
  std::shared_ptr<InFlightBase> op;
  {
    std::lock_guard<FairMutex> lk(pc->mu);
    if (!pc->ready) return;
    for (auto& f : pc->inflight) if (f && f->ticket == ticket) op = f;
  }
  if (!op) return;
  (void)hipSetDevice(pc->device);
  try { op->wait_device(); } catch (const HipError&) { (void)hipGetLastError(); }      // the locked phase meets the same error and reports it
}

// Entry points that move an object from the context `route` lives on to logical device `target`: both contexts locked
// (std::lock: no ordering deadlock), the source drained, the HIP current device set to the target's.
template <class F>
int guarded_pair(gs_handle route, int target, F&& f) {
  size_t ndev = 0;
  std::shared_ptr<Ctx> ps = ctx_ref(handle_device(route), &ndev), pd = ctx_ref(target);
  if (ndev == 0) return fail(GS_ERR_NOT_INIT, "gs_init has not been called (or failed)");
  if (!ps) return fail(GS_ERR_ARG, "bad handle (no logical device %d)", handle_device(route));
  if (!pd) return fail(GS_ERR_ARG, "no logical device %d (gs_init listed %zu)", target, ndev);
  Ctx& src = *ps;
  Ctx& dst = *pd;
  std::unique_lock<FairMutex> l1(src.mu, std::defer_lock), l2(dst.mu, std::defer_lock);
  if (&src == &dst) l1.lock(); else std::lock(l1, l2);
  if (!src.ready || !dst.ready) return fail(GS_ERR_NOT_INIT, "the library was shut down");
  CurrentCtxScope scope(&dst);                  // allocations of a pair call land on the target
  dst.call_clock += 1;
  try {
    (void)hipSetDevice(src.device);
    src.drain();
    (void)hipSetDevice(dst.device);
    if (&src != &dst && dst.any_inflight()) dst.drain();
    return f(src, dst);
  } catch (const HipError& e) {
    return fail(GS_ERR_HIP, "HIP error %d (%s) at %s line %d", (int)e.e, hipGetErrorString(e.e), e.what, e.line);
  } catch (const std::bad_alloc&) {
    return fail(GS_ERR_HIP, "host allocation failed");
  } catch (const std::exception& e) {
    return fail(GS_ERR_ARG, "%s", e.what());
  }
}

// Device-to-device copy between two contexts (key slices, clones).  Same physical GPU: a plain copy; different GPUs:
// hipMemcpyPeerAsync, which is correct with or without peer access (gs_init enables it where the driver allows; without it the
// runtime stages through the host) -- a plain hipMemcpyDeviceToDevice between devices without peer access is unspecified.
inline void copy_between(Ctx& dst, void* d, const Ctx& src, const void* s, size_t bytes) {
  if (!bytes) return;
  if (dst.device == src.device) GS_HIP(hipMemcpyAsync(d, s, bytes, hipMemcpyDeviceToDevice, dst.stream));
  else GS_HIP(hipMemcpyPeerAsync(d, dst.device, s, src.device, bytes, dst.stream));
}

// run a section of engine calls on another stream of the context
struct StreamScope {
  Ctx& c;
  hipStream_t saved;
  StreamScope(Ctx& ctx_, hipStream_t s) : c(ctx_), saved(ctx_.stream) { c.stream = s; }
  ~StreamScope() { c.stream = saved; }
};

// RAII event timer on the library stream
struct PhaseTimer {
  hipEvent_t a, b;
  hipStream_t s;
  explicit PhaseTimer(hipStream_t st) : s(st) {
    GS_HIP(hipEventCreate(&a));
    GS_HIP(hipEventCreate(&b));
    GS_HIP(hipEventRecord(a, s));
  }
  void stop() { GS_HIP(hipEventRecord(b, s)); }
  float ms() {
    float m = 0;
    GS_HIP(hipEventSynchronize(b));
    GS_HIP(hipEventElapsedTime(&m, a, b));
    return m;
  }
  ~PhaseTimer() { (void)hipEventDestroy(a); (void)hipEventDestroy(b); }
};

}  // namespace gs
