// extern "C" surface of libgosnark_hip.so (declared in include/gosnark_hip.h).
#include <algorithm>
#include <cstdlib>
#include <vector>

#include "msm.h"
#include "hostcopy.h"
#include "point_io.h"
#include "poly.h"
#include "prove.h"
#include "runtime.h"

using namespace gs;

namespace gs { void multi_shutdown(); }

namespace {


template <class T>
int upload_bases(Ctx& c, Kind kind, const uint64_t* jac, size_t n, gs_handle* out) {
  if (!out || (n && !jac)) return fail(GS_ERR_ARG, "null argument");
  if (n >= (1ull << 31)) return fail(GS_ERR_ARG, "too many points");
  constexpr size_t cw = T::kWords;
  auto b = std::make_unique<Bases>(kind);
  b->n = n;
  b->buf.alloc(std::max<size_t>(n, 1) * 2 * cw * 4);
  if (n) {
    uint32_t first_bad = 0;
    const uint32_t bad = kind == Kind::G1Bases ? upload_jacobian_g1(c, jac, (uint32_t)n, b->buf.as<uint32_t>(), &first_bad)
                                               : upload_jacobian_g2(c, jac, (uint32_t)n, b->buf.as<uint32_t>(), &first_bad);
    if (bad) return fail(GS_ERR_ARG, "%u of the %zu points are not on the curve (first at index %u)", bad, n, first_bad);
  }
  *out = c.put(std::move(b));
  return GS_OK;
}

template <class T>
int download_bases(Ctx& c, Kind kind, gs_handle h, uint64_t* jac, size_t n) {
  Bases* b = c.get<Bases>(h, kind);
  if (!b) return fail(GS_ERR_ARG, "bad base handle");
  if (n != b->n || (n && !jac)) return fail(GS_ERR_ARG, "size mismatch");
  if (!n) return GS_OK;
  constexpr size_t cw = T::kWords;
  DevBuf tmp(n * 3 * cw * 4);
  if (kind == Kind::G1Bases) affine_to_jacobian_std_g1(c, b->buf.as<uint32_t>(), (uint32_t)n, tmp.as<uint32_t>());
  else affine_to_jacobian_std_g2(c, b->buf.as<uint32_t>(), (uint32_t)n, tmp.as<uint32_t>());
  GS_HIP(hipMemcpyAsync(jac, tmp.p, n * 3 * cw * 4, hipMemcpyDeviceToHost, c.stream));
  GS_HIP(hipStreamSynchronize(c.stream));
  return GS_OK;
}

template <class T>
int fixed_base_api(Ctx& c, Kind kind, const uint64_t* scalars, size_t n, gs_handle* out) {
  if (!out || (n && !scalars)) return fail(GS_ERR_ARG, "null argument");
  if (n >= (1ull << 31)) return fail(GS_ERR_ARG, "too many scalars");
  constexpr size_t cw = T::kWords;
  auto b = std::make_unique<Bases>(kind);
  b->n = n;
  b->buf.alloc(std::max<size_t>(n, 1) * 2 * cw * 4);
  if (n) {
    DevBuf tmp(n * 32);
    GS_HIP(hipMemcpyAsync(tmp.p, scalars, n * 32, hipMemcpyHostToDevice, c.stream));
    if (kind == Kind::G1Bases) fixed_base_g1(c, tmp.as<uint32_t>(), (uint32_t)n, b->buf.as<uint32_t>());
    else fixed_base_g2(c, tmp.as<uint32_t>(), (uint32_t)n, b->buf.as<uint32_t>());
    GS_HIP(hipStreamSynchronize(c.stream));
  }
  *out = c.put(std::move(b));
  return GS_OK;
}

// one MSM over resident scalars
template <class T>
int msm_resident(Ctx& c, Kind kind, gs_handle hb, size_t off, const uint32_t* scalars_dev, size_t n,
                 uint64_t* out_affine, int* is_inf) {
  Bases* b = c.get<Bases>(hb, kind);
  if (!b) return fail(GS_ERR_ARG, "bad base handle");
  if (!out_affine || !is_inf) return fail(GS_ERR_ARG, "null output");
  if (n > (size_t)kIndexMask) return fail(GS_ERR_ARG, "at most 2^26 - 1 terms per MSM call (shard larger sums)");
  if (off > b->n || n > b->n - off) return fail(GS_ERR_ARG, "term range [%zu, %zu) exceeds the %zu resident points", off, off + n, b->n);
  // window table of this base array for the width the plan will use (built on first use, kept resident)
  if (!b->table) b->table = std::make_shared<BaseTable>();
  BaseTable* tab = static_cast<BaseTable*>(b->table.get());
  int cbits = 0;
  double credit = build_credit(T::kWords == 16 ? 2.76 : 1.0, n);      // policy auto: this call's instalment of the array's table
  const bool tabled = prepare_tables(c, {TableRef{tab, b->buf.as<uint32_t>(), b->n, T::kWords == 16}}, (uint32_t)n, &cbits, &credit);
  PhaseTimer total(c.stream);
  MsmPlan plan;
  {
    PhaseTimer tp(c.stream);
    build_plan(c, 2 * c.blocking_slot(), scalars_dev, (uint32_t)n, plan, {{1, T::kWords == 16}}, cbits, !tabled);
    tp.stop();
    c.timing.plan_ms += tp.ms();
  }
  std::vector<MsmBase> bases{MsmBase{tab, off, b->buf.as<uint32_t>(), b->n}};
  bool inf;
  if constexpr (T::kWords == 8) {
    std::vector<G1Xyzz> r;
    msm_run_g1(c, plan, bases, r);
    inf = g1_to_affine_std(r[0], out_affine);
  } else {
    std::vector<G2Xyzz> r;
    msm_run_g2(c, plan, bases, r);
    inf = g2_to_affine_std(r[0], out_affine);
  }
  GS_HIP(hipStreamSynchronize(c.stream));
  *is_inf = inf ? 1 : 0;
  total.stop();
  c.timing.total_ms += total.ms();
  return GS_OK;
}

// ---- pipelined MSM: begin enqueues (plan on aux stream 1, accumulation on the main stream, tail on aux stream 2), end collects --
struct MsmInFlight : InFlightBase {
  bool g2 = false;
  MsmPending pend;
  std::shared_ptr<PhaseTimer> tplan;
  hipEvent_t planned = nullptr, done = nullptr;
  MsmInFlight() {
    GS_HIP(hipEventCreateWithFlags(&planned, hipEventDisableTiming));
    GS_HIP(hipEventCreateWithFlags(&done, hipEventDisableTiming));
  }
  void wait_device() const override { if (done) GS_HIP(hipEventSynchronize(done)); }
  ~MsmInFlight() override { for (hipEvent_t e : {planned, done}) if (e) (void)hipEventDestroy(e); }
};

template <class T>
int msm_begin(Ctx& c, Kind kind, gs_handle hb, size_t off, gs_handle hs, size_t soff, size_t n, uint64_t* ticket) {
  Bases* b = c.get<Bases>(hb, kind);
  Scalars* sc = c.get<Scalars>(hs, Kind::Scalars);
  if (!b || !sc || !ticket) return fail(GS_ERR_ARG, "gs_msm_begin: bad handle");
  if (n == 0 || n > (size_t)kIndexMask) return fail(GS_ERR_ARG, "gs_msm_begin: 1 .. 2^26 - 1 terms per call");
  if (off > b->n || n > b->n - off || soff > sc->n || n > sc->n - soff) return fail(GS_ERR_ARG, "gs_msm_begin: range exceeds the resident arrays");
  const int parity = c.free_parity();
  if (parity < 0) return fail(GS_ERR_BUSY, "gs_msm_begin: three operations are already outstanding");
  if (!b->table) b->table = std::make_shared<BaseTable>();
  BaseTable* tab = static_cast<BaseTable*>(b->table.get());
  int cbits = 0;
  double credit = build_credit(T::kWords == 16 ? 2.76 : 1.0, n);      // policy auto: this call's instalment of the array's table
  const bool tabled = prepare_tables(c, {TableRef{tab, b->buf.as<uint32_t>(), b->n, T::kWords == 16}}, (uint32_t)n, &cbits, &credit);
  auto st = std::make_unique<MsmInFlight>();
  st->g2 = T::kWords == 16;
  st->keep = {c.share<Object>(hb, kind), c.share<Object>(hs, Kind::Scalars)};
  MsmPlan plan;
  // GS_MSM_TICKET_STREAMS (scheduling only, same results): 0 = every ticket's plan on aux 1, tails alternating between aux 0 / aux 2;
  // 1 = ticket slot p keeps its plan AND its tail on aux p, so the plans of consecutive tickets run beside each other instead of
  // back to back (profiles/r04_timeline_msm_g1_steady.txt: the plan stream is the busy one of the MSM pipeline); 2 (default) = 1 above
  // 3 * 2^20 terms.  Measured (profiles/r04_ab_msm_ticket_streams.txt): 2^22 terms 5.69-5.88 -> 5.29-5.39 ms per MSM, 2^21 and below
  // unchanged within 2 % -- there the plans and tails cost the accumulation as much beside it as they would behind it.
  static const long ticket_mode = run_knob("GS_MSM_TICKET_STREAMS", 2, 0, 2);
  const bool ticket_streams = ticket_mode == 1 || (ticket_mode == 2 && n >= ((size_t)3 << 20));
  hipStream_t plan_stream = ticket_streams ? c.aux_stream[parity % Ctx::kAuxStreams] : c.aux_stream[1];
  {
    StreamScope ss(c, plan_stream);
    st->tplan = std::make_shared<PhaseTimer>(c.stream);
    build_plan(c, 2 * parity, sc->buf.as<uint32_t>() + soff * 8, (uint32_t)n, plan, {{1, st->g2}}, cbits, !tabled);
    st->tplan->stop();
    GS_HIP(hipEventRecord(st->planned, c.stream));
  }
  hipStream_t tail = nullptr;
  {
    StreamScope ss(c, c.main_stream);
    GS_HIP(hipStreamWaitEvent(c.stream, st->planned, 0));
    std::vector<MsmBase> bases{MsmBase{tab, off, b->buf.as<uint32_t>(), b->n}};
    if (!ticket_streams) c.next_tails((uint32_t)n);          // consecutive small MSMs reduce on alternating tail streams
    tail = ticket_streams ? plan_stream : c.tail_stream(0);
    if constexpr (T::kWords == 8) msm_enqueue_g1(c, plan, bases, 8 * parity, 3 * parity, st->pend, tail);
    else msm_enqueue_g2(c, plan, bases, 8 * parity + 4, 3 * parity, st->pend, tail);
  }
  GS_HIP(hipEventRecord(st->done, tail));
  sc->mark_read(plan_stream);                          // gs_scalars_update on this vector waits for the plan's digit pass
  st->ticket = c.new_ticket();
  *ticket = st->ticket;
  c.inflight[parity] = std::move(st);
  return GS_OK;
}

int msm_end(Ctx& c, uint64_t ticket, uint64_t* out_affine, int* is_inf) {
  if (!out_affine || !is_inf) return fail(GS_ERR_ARG, "null output");
  int parity = -1;
  for (int p = 0; p < Ctx::kMaxInFlight; ++p) if (c.inflight[p] && c.inflight[p]->ticket == ticket) parity = p;
  if (parity < 0) return fail(GS_ERR_ARG, "gs_msm_end: unknown ticket %llu", (unsigned long long)ticket);
  MsmInFlight* st = dynamic_cast<MsmInFlight*>(c.inflight[parity].get());
  if (!st) return fail(GS_ERR_ARG, "gs_msm_end: ticket %llu belongs to a proof (use gs_groth16_prove_end)", (unsigned long long)ticket);
  std::shared_ptr<InFlightBase> base = std::move(c.inflight[parity]);
  GS_HIP(hipEventSynchronize(st->done));
  reset_timing(c);
  c.timing.plan_ms = st->tplan->ms();
  msm_book_timing(c, st->pend);
  bool inf;
  if (!st->g2) { std::vector<G1Xyzz> r; msm_finish_g1(c, st->pend, r); inf = g1_to_affine_std(r[0], out_affine); }
  else { std::vector<G2Xyzz> r; msm_finish_g2(c, st->pend, r); inf = g2_to_affine_std(r[0], out_affine); }
  *is_inf = inf ? 1 : 0;
  c.timing.total_ms = c.timing.plan_ms + c.timing.accumulate_ms + c.timing.reduce_ms;
  return GS_OK;
}

template <class T>
int msm_host_scalars(Ctx& c, Kind kind, gs_handle hb, const uint64_t* scalars, size_t off, size_t n,
                     uint64_t* out_affine, int* is_inf) {
  if (n && !scalars) return fail(GS_ERR_ARG, "null scalars");
  if (n >= (1ull << 31)) return fail(GS_ERR_ARG, "too many terms");
  reset_timing(c);
  c.ws_misc.ensure(std::max<size_t>(n, 1) * 32);
  if (n) {
    PhaseTimer th(c.stream);
    staged_h2d(c, c.ws_misc.p, scalars, n * 32, c.stream);
    th.stop();
    c.timing.h2d_ms += th.ms();
  }
  return msm_resident<T>(c, kind, hb, off, c.ws_misc.as<uint32_t>(), n, out_affine, is_inf);
}

template <class T>
int sum_affine(const uint64_t* pts, const int* inf, size_t n, uint64_t* out, int* is_inf) {
  if ((n && (!pts || !inf)) || !out || !is_inf) return fail(GS_ERR_ARG, "null argument");
  constexpr int cw = T::kWords;     // u32 words per coordinate = u64 words per point half... (2*cw u32 per point)
  Xyzz<T> acc = xyzz_inf<T>();
  for (size_t i = 0; i < n; ++i) {
    if (inf[i]) continue;
    const uint32_t* p = reinterpret_cast<const uint32_t*>(pts) + i * 2 * cw;
    Affine<T> a;
    a.x = canon(PointIO<T>::load_std(p));
    a.y = canon(PointIO<T>::load_std(p + cw));
    xyzz_madd(acc, a, false);
  }
  bool r;
  if constexpr (T::kWords == 8) r = g1_to_affine_std(acc, out); else r = g2_to_affine_std(acc, out);
  *is_inf = r ? 1 : 0;
  return GS_OK;
}

}  // namespace

extern "C" {

#ifndef GS_BUILD_FLAGS
#define GS_BUILD_FLAGS "unknown"
#endif
const char* gs_version(void) { return "gosnark-hip 0.5 gfx950 (9x29-bit Montgomery, XYZZ Pippenger on window tables or table-free, evaluation-basis keys, host-buffer tickets); built with " GS_BUILD_FLAGS; }
const char* gs_last_error(void) { return last_error_ref().c_str(); }

// One context per entry of `devices` (a "logical device": its own streams, workspaces, handle table and lock).  The same
// HIP ordinal may be listed several times -- N logical devices time-slicing one GPU -- which is how the multi-device
// entry points (multi.hip) are exercised on a single-GPU box.
static void ctx_create(Ctx& c, int logical, int device) {
  c.logical = logical;
  c.device = device;
  GS_HIP(hipSetDevice(device));
  GS_HIP(hipStreamCreateWithFlags(&c.main_stream, hipStreamNonBlocking));
  // The aux streams carry the latency-/bandwidth-bound shadow work of a proof (NTT passes, plan kernels, bucket
  // combine / reduction tails) next to the ALU-bound accumulations on the main stream.  They get the HIGHEST queue
  // priority: their kernels are short but hard to place (k_hist wants 128 KiB of LDS and 16 wave slots of one CU),
  // and a starved plan(h) would stall the last accumulation (seen in an earlier timeline).
  int least = 0, greatest = 0;
  GS_HIP(hipDeviceGetStreamPriorityRange(&least, &greatest));
  constexpr long kTailPriorityDefault = 1;
  const bool no_overlap = run_flag("GS_NO_OVERLAP");                // debugging aid: everything on the main stream (same results)
  const bool no_prio = run_flag("GS_NO_PRIORITY");
  // GS_TAIL_PRIORITY (same results, scheduling only): queue priority of the two reduction-tail streams -- 0 = highest, like the
  // plan / H(x) stream (rounds 1-3), 1 = the default priority of the accumulation stream, 2 = lowest.  Round 4's tail kernels fit
  // beside an accumulation wave (<= 256 VGPRs), so they no longer need help to be placed; at small sizes they got in the way of
  // the plans and NTT passes that set the pace (profiles/r04_ab_tail_priority.txt).
  const long tail_prio = run_knob("GS_TAIL_PRIORITY", kTailPriorityDefault, 0, 2);
  for (int i = 0; i < Ctx::kAuxStreams; ++i) {
    hipStream_t& a = c.aux_stream[i];
    if (no_overlap) { a = c.main_stream; continue; }
    int prio = greatest;
    if (i != 1 && tail_prio == 1) prio = (least + greatest) / 2;
    if (i != 1 && tail_prio == 2) prio = least;
    GS_HIP(hipStreamCreateWithPriority(&a, hipStreamNonBlocking, no_prio ? least : prio));
  }
  for (auto& p : c.pinned) GS_HIP(hipHostMalloc(&p, Ctx::kPinnedBytes, hipHostMallocDefault));
  c.bad_dev.alloc(Ctx::kSlots * 4);
  GS_HIP(hipHostMalloc(reinterpret_cast<void**>(&c.bad_host), Ctx::kSlots * 4, hipHostMallocDefault));
  memset(c.bad_host, 0, Ctx::kSlots * 4);
  c.stream = c.main_stream;
  c.table_policy = (int)run_knob("GS_TABLE_POLICY", 0, 0, 2);      // initial value of gs_set_table_policy (same results either way)
  c.ready = true;
}

static void ctx_destroy(Ctx& c) {
  std::lock_guard<FairMutex> lk(c.mu);
  if (!c.ready) return;
  (void)hipSetDevice(c.device);
  (void)hipDeviceSynchronize();
  for (auto& f : c.inflight) f.reset();
  c.objs.clear();
  c.msm_state.reset(); c.poly_state.reset(); c.prove_state.reset();
  for (auto& a : c.aux_stream) {
    if (a && a != c.main_stream) (void)hipStreamDestroy(a);
    a = nullptr;
  }
  if (c.main_stream) (void)hipStreamDestroy(c.main_stream);
  c.main_stream = nullptr;
  c.stream = nullptr;
  for (auto& pp : c.pinned) {
    if (pp) (void)hipHostFree(pp);
    pp = nullptr;
  }
  if (c.bad_host) (void)hipHostFree(c.bad_host);
  c.bad_host = nullptr;
  c.bad_dev.release();
  if (c.copy_stream) (void)hipStreamDestroy(c.copy_stream);
  c.copy_stream = nullptr;
  if (c.table_stream) (void)hipStreamDestroy(c.table_stream);
  c.table_stream = nullptr;
  for (int b = 0; b < Ctx::kStageBuffers; ++b) {
    if (c.stage[b]) (void)hipHostFree(c.stage[b]);
    if (c.stage_ev[b]) (void)hipEventDestroy(c.stage_ev[b]);
    c.stage[b] = nullptr; c.stage_ev[b] = nullptr;
  }
  for (auto& b : c.ws_buckets) b.release();
  for (auto& b : c.ws_chunks) b.release();
  for (auto& b : c.ws_partials) b.release();
  for (auto& b : c.ws_out) b.release();
  c.ws_misc.release(); c.g1_pow2.release(); c.g2_pow2.release();
  c.ready = false;
}

int gs_init(const int* devices, int ndev) {
  Registry& r = registry();
  std::lock_guard<std::mutex> lk(r.mu);
  try {
    if (!devices || ndev < 1 || ndev > kMaxLogicalDevices)
      return fail(GS_ERR_ARG, "gs_init: need 1 .. %d devices (got ndev=%d)", kMaxLogicalDevices, ndev);
    int count = 0;
    hipError_t e = hipGetDeviceCount(&count);
    if (e != hipSuccess || count == 0)
      return fail(GS_ERR_NO_DEVICE, "no HIP device visible (%s); libgosnark_hip has no CPU path", e == hipSuccess ? "count=0" : hipGetErrorString(e));
    for (int i = 0; i < ndev; ++i) {
      if (devices[i] < 0 || devices[i] >= count) return fail(GS_ERR_ARG, "device %d out of range (0..%d)", devices[i], count - 1);
      hipDeviceProp_t prop;
      GS_HIP(hipGetDeviceProperties(&prop, devices[i]));
      if (strncmp(prop.gcnArchName, "gfx950", 6) != 0)
        return fail(GS_ERR_NO_DEVICE, "device %d is %s; this library is built for gfx950 (MI355X) only", devices[i], prop.gcnArchName);
    }
    if (!r.ctxs.empty()) {                          // a second gs_init must name the same list (idempotent), anything else needs gs_shutdown
      bool same = r.ctxs.size() == (size_t)ndev;
      for (int i = 0; same && i < ndev; ++i) same = r.ctxs[i]->device == devices[i];
      if (same) return GS_OK;
      return fail(GS_ERR_ARG, "already initialised on %zu device(s) starting with device %d: call gs_shutdown first", r.ctxs.size(), r.ctxs[0]->device);
    }
    static bool registered = false;
    if (!registered) { atexit([] { process_exiting() = true; }); registered = true; }
    std::vector<std::shared_ptr<Ctx>> fresh;
    for (int i = 0; i < ndev; ++i) {
      fresh.push_back(std::make_shared<Ctx>());
      ctx_create(*fresh.back(), i, devices[i]);
    }
    // distinct physical devices copy key slices / scalar vectors to each other directly over xGMI
    for (int i = 0; i < ndev; ++i)
      for (int j = 0; j < ndev; ++j)
        if (devices[i] != devices[j]) {
          int can = 0;
          if (hipDeviceCanAccessPeer(&can, devices[i], devices[j]) == hipSuccess && can) {
            (void)hipSetDevice(devices[i]);
            (void)hipDeviceEnablePeerAccess(devices[j], 0);      // "already enabled" is fine
            (void)hipGetLastError();
          }
        }
    r.ctxs = std::move(fresh);
    return GS_OK;
  } catch (const HipError& e) {
    return fail(GS_ERR_HIP, "HIP error %d (%s) at %s line %d", (int)e.e, hipGetErrorString(e.e), e.what, e.line);
  } catch (const std::exception& e) {
    return fail(GS_ERR_ARG, "%s", e.what());
  }
}

void gs_shutdown(void) {
  multi_shutdown();                                 // communicators first (multi.hip)
  Registry& r = registry();
  std::lock_guard<std::mutex> lk(r.mu);
  for (auto& c : r.ctxs) ctx_destroy(*c);
  r.ctxs.clear();                                   // nothing survives: a later gs_init may name other devices
}

int gs_device_count(void) { return (int)logical_device_count(); }

int gs_set_device(int logical) {
  const size_t n = logical_device_count();
  if (logical < 0 || (size_t)logical >= n) return fail(GS_ERR_ARG, "gs_set_device: no logical device %d (gs_init listed %zu)", logical, n);
  current_logical() = logical;
  return GS_OK;
}

int gs_get_device(void) { return current_logical(); }

int gs_handle_device(gs_handle h) { return handle_device(h); }

int gs_free(gs_handle h) {
  return guarded([&](Ctx& c) -> int {
    if (!c.objs.erase(h)) return fail(GS_ERR_ARG, "gs_free: unknown handle %llu", (unsigned long long)h);
    return GS_OK;
  }, true, true, h);
}

int gs_len(gs_handle h, size_t* out) {
  return guarded([&](Ctx& c) -> int {
    auto it = c.objs.find(h);
    if (it == c.objs.end() || !out) return fail(GS_ERR_ARG, "gs_len: bad handle");
    switch (it->second->kind) {
      case Kind::G1Bases: case Kind::G2Bases: *out = static_cast<Bases*>(it->second.get())->n; return GS_OK;
      case Kind::Scalars: *out = static_cast<Scalars*>(it->second.get())->n; return GS_OK;
      case Kind::GrothPk: *out = static_cast<GrothPkObj*>(it->second.get())->n_w; return GS_OK;       // entries of At held (a slice holds fewer than NVars)
      default: return fail(GS_ERR_ARG, "gs_len: handle has no length");
    }
  }, true, true, h);
}

int gs_g1_upload(const uint64_t* jac, size_t n, gs_handle* out) {
  return guarded([&](Ctx& c) { return upload_bases<FqTag>(c, Kind::G1Bases, jac, n, out); }, true, true);
}
int gs_g2_upload(const uint64_t* jac, size_t n, gs_handle* out) {
  return guarded([&](Ctx& c) { return upload_bases<Fq2Tag>(c, Kind::G2Bases, jac, n, out); }, true, true);
}
int gs_g1_download(gs_handle h, uint64_t* jac, size_t n) {
  return guarded([&](Ctx& c) { return download_bases<FqTag>(c, Kind::G1Bases, h, jac, n); }, true, true, h);
}
int gs_g2_download(gs_handle h, uint64_t* jac, size_t n) {
  return guarded([&](Ctx& c) { return download_bases<Fq2Tag>(c, Kind::G2Bases, h, jac, n); }, true, true, h);
}
int gs_g1_fixed_base(const uint64_t* s, size_t n, gs_handle* out) {
  return guarded([&](Ctx& c) { return fixed_base_api<FqTag>(c, Kind::G1Bases, s, n, out); });
}
int gs_g2_fixed_base(const uint64_t* s, size_t n, gs_handle* out) {
  return guarded([&](Ctx& c) { return fixed_base_api<Fq2Tag>(c, Kind::G2Bases, s, n, out); });
}

int gs_scalars_upload(const uint64_t* s, size_t n, gs_handle* out) {
  return guarded([&](Ctx& c) -> int {
    if (!out || (n && !s)) return fail(GS_ERR_ARG, "null argument");
    if (n >= (1ull << 31)) return fail(GS_ERR_ARG, "too many scalars");
    auto o = std::make_unique<Scalars>();
    o->n = n;
    o->buf.alloc(std::max<size_t>(n, 1) * 32);
    if (n) {                       // own stream: an upload must not queue behind the accumulations of outstanding tickets
      if (!c.copy_stream) GS_HIP(hipStreamCreateWithFlags(&c.copy_stream, hipStreamNonBlocking));
      staged_h2d(c, o->buf.p, s, n * 32, c.copy_stream);
      GS_HIP(hipStreamSynchronize(c.copy_stream));
    }
    *out = c.put(std::move(o));
    return GS_OK;
  }, true, true);
}
// Overwrite a resident vector IN PLACE with n = its length new scalars from caller memory: no hipMalloc, no hipFree (which would
// synchronise the whole device under outstanding tickets).  The copy runs on the copy stream behind every device read of the vector
// that pipelined operations enqueued before this call (Scalars::reads) and has landed when the call returns, so whatever is
// enqueued afterwards sees the new values.  With >= 4 vectors rotating under three tickets the wait is nil: a ticket's last read of
// its witness is its plan's digit pass, long before it is collected.
int gs_scalars_update(gs_handle h, const uint64_t* s, size_t n) {
  return guarded([&](Ctx& c) -> int {
    Scalars* o = c.get<Scalars>(h, Kind::Scalars);
    if (!o) return fail(GS_ERR_ARG, "gs_scalars_update: bad handle");
    if (n != o->n || (n && !s)) return fail(GS_ERR_ARG, "gs_scalars_update: the vector holds %zu scalars, the call brings %zu", o->n, n);
    if (!n) return GS_OK;
    if (!c.copy_stream) GS_HIP(hipStreamCreateWithFlags(&c.copy_stream, hipStreamNonBlocking));
    for (auto& r : o->reads) GS_HIP(hipStreamWaitEvent(c.copy_stream, r.ev, 0));
    staged_h2d(c, o->buf.p, s, n * 32, c.copy_stream);
    GS_HIP(hipStreamSynchronize(c.copy_stream));
    return GS_OK;
  }, true, true, h);
}
// how often the library called hipMalloc / hipFree so far (process-wide): a proof stream in steady state moves neither
int gs_alloc_counters(uint64_t* allocs, uint64_t* frees) {
  if (allocs) *allocs = devbuf_allocs().load();
  if (frees) *frees = devbuf_frees().load();
  return GS_OK;
}
int gs_scalars_download(gs_handle h, uint64_t* out, size_t n) {
  return guarded([&](Ctx& c) -> int {
    Scalars* s = c.get<Scalars>(h, Kind::Scalars);
    if (!s || n != s->n || (n && !out)) return fail(GS_ERR_ARG, "gs_scalars_download: bad handle or size");
    GS_HIP(hipStreamSynchronize(c.stream));
    if (n) GS_HIP(hipMemcpy(out, s->buf.p, n * 32, hipMemcpyDeviceToHost));
    return GS_OK;
  }, true, true, h);
}

// Copies onto another logical device (the source may live on any device: the handle says where).
int gs_scalars_clone(gs_handle h, size_t off, size_t n, int target_device, gs_handle* out) {
  return guarded_pair(h, target_device, [&](Ctx& src, Ctx& dst) -> int {
    Scalars* s = src.get<Scalars>(h, Kind::Scalars);
    if (!s || !out) return fail(GS_ERR_ARG, "gs_scalars_clone: bad handle or null output");
    if (off > s->n || n > s->n - off) return fail(GS_ERR_ARG, "gs_scalars_clone: range [%zu, %zu) exceeds the %zu resident scalars", off, off + n, s->n);
    auto o = std::make_unique<Scalars>();
    o->n = n;
    o->buf.alloc(std::max<size_t>(n, 1) * 32);
    copy_between(dst, o->buf.p, src, s->buf.as<uint32_t>() + off * 8, n * 32);
    GS_HIP(hipStreamSynchronize(dst.stream));
    *out = dst.put(std::move(o));
    return GS_OK;
  });
}
static int bases_clone(Kind kind, size_t words, gs_handle h, size_t off, size_t n, int target_device, gs_handle* out) {
  return guarded_pair(h, target_device, [&](Ctx& src, Ctx& dst) -> int {
    Bases* b = src.get<Bases>(h, kind);
    if (!b || !out) return fail(GS_ERR_ARG, "gs_bases_clone: bad handle or null output");
    if (off > b->n || n > b->n - off) return fail(GS_ERR_ARG, "gs_bases_clone: range [%zu, %zu) exceeds the %zu resident points", off, off + n, b->n);
    auto o = std::make_unique<Bases>(kind);
    o->n = n;
    o->buf.alloc(std::max<size_t>(n, 1) * words * 4);
    copy_between(dst, o->buf.p, src, b->buf.as<uint32_t>() + off * words, n * words * 4);
    GS_HIP(hipStreamSynchronize(dst.stream));
    *out = dst.put(std::move(o));
    return GS_OK;
  });
}
int gs_g1_clone(gs_handle h, size_t off, size_t n, int target_device, gs_handle* out) { return bases_clone(Kind::G1Bases, 16, h, off, n, target_device, out); }
int gs_g2_clone(gs_handle h, size_t off, size_t n, int target_device, gs_handle* out) { return bases_clone(Kind::G2Bases, 32, h, off, n, target_device, out); }

int gs_msm_g1(gs_handle bases, const uint64_t* scalars, size_t off, size_t n, uint64_t out_affine[8], int* is_inf) {
  return guarded([&](Ctx& c) { return msm_host_scalars<FqTag>(c, Kind::G1Bases, bases, scalars, off, n, out_affine, is_inf); }, true, false, bases);
}
int gs_msm_g2(gs_handle bases, const uint64_t* scalars, size_t off, size_t n, uint64_t out_affine[16], int* is_inf) {
  return guarded([&](Ctx& c) { return msm_host_scalars<Fq2Tag>(c, Kind::G2Bases, bases, scalars, off, n, out_affine, is_inf); }, true, false, bases);
}
int gs_msm_g1_resident(gs_handle bases, size_t off, gs_handle scalars, size_t soff, size_t n, uint64_t out_affine[8], int* is_inf) {
  return guarded([&](Ctx& c) -> int {
    Scalars* s = c.get<Scalars>(scalars, Kind::Scalars);
    if (!s || soff > s->n || n > s->n - soff) return fail(GS_ERR_ARG, "bad scalar handle or range");
    reset_timing(c);
    return msm_resident<FqTag>(c, Kind::G1Bases, bases, off, s->buf.as<uint32_t>() + soff * 8, n, out_affine, is_inf);
  }, true, false, bases);
}
int gs_msm_g2_resident(gs_handle bases, size_t off, gs_handle scalars, size_t soff, size_t n, uint64_t out_affine[16], int* is_inf) {
  return guarded([&](Ctx& c) -> int {
    Scalars* s = c.get<Scalars>(scalars, Kind::Scalars);
    if (!s || soff > s->n || n > s->n - soff) return fail(GS_ERR_ARG, "bad scalar handle or range");
    reset_timing(c);
    return msm_resident<Fq2Tag>(c, Kind::G2Bases, bases, off, s->buf.as<uint32_t>() + soff * 8, n, out_affine, is_inf);
  }, true, false, bases);
}

int gs_msm_g1_begin(gs_handle bases, size_t off, gs_handle scalars, size_t soff, size_t n, uint64_t* ticket) {
  return guarded([&](Ctx& c) { return msm_begin<FqTag>(c, Kind::G1Bases, bases, off, scalars, soff, n, ticket); }, true, true, bases);
}
int gs_msm_g2_begin(gs_handle bases, size_t off, gs_handle scalars, size_t soff, size_t n, uint64_t* ticket) {
  return guarded([&](Ctx& c) { return msm_begin<Fq2Tag>(c, Kind::G2Bases, bases, off, scalars, soff, n, ticket); }, true, true, bases);
}
int gs_msm_end(uint64_t ticket, uint64_t* out_affine, int* is_inf) {
  wait_ticket_unlocked(ticket);          // the device wait, outside the context lock (runtime.h)
  return guarded([&](Ctx& c) { return msm_end(c, ticket, out_affine, is_inf); }, true, true, ticket);
}

int gs_g1_sum_affine(const uint64_t* pts, const int* inf, size_t n, uint64_t out[8], int* is_inf) {
  return guarded([&](Ctx&) { return sum_affine<FqTag>(pts, inf, n, out, is_inf); }, false);   // host arithmetic: no device needed
}
int gs_g2_sum_affine(const uint64_t* pts, const int* inf, size_t n, uint64_t out[16], int* is_inf) {
  return guarded([&](Ctx&) { return sum_affine<Fq2Tag>(pts, inf, n, out, is_inf); }, false);
}

int gs_last_timing(gs_timing* out) {
  return guarded([&](Ctx& c) -> int {
    if (!out) return fail(GS_ERR_ARG, "null");
    *out = c.timing;
    return GS_OK;
  }, true, true);
}

int gs_device_timing(int logical_device, gs_timing* out) {
  if (!out) return fail(GS_ERR_ARG, "null");
  std::shared_ptr<Ctx> pc = ctx_ref(logical_device);
  if (!pc) return fail(GS_ERR_ARG, "gs_device_timing: no logical device %d", logical_device);
  Ctx& c = *pc;
  std::lock_guard<FairMutex> lk(c.mu);
  if (!c.ready) return fail(GS_ERR_NOT_INIT, "the library was shut down");
  *out = c.timing;
  return GS_OK;
}

// Give up a pipelined operation: waits for the device work it enqueued (its workspaces must be quiet before another operation
// takes the slot), then releases the slot and the references to the key and vectors it read.  Nothing is returned.
int gs_ticket_cancel(uint64_t ticket) {
  return guarded([&](Ctx& c) -> int {
    for (int p = 0; p < Ctx::kMaxInFlight; ++p)
      if (c.inflight[p] && c.inflight[p]->ticket == ticket) {
        c.drain();
        c.inflight[p].reset();
        return GS_OK;
      }
    return fail(GS_ERR_ARG, "gs_ticket_cancel: unknown ticket %llu", (unsigned long long)ticket);
  }, true, true, ticket);
}

// applies to every logical device (a process-wide tunable, like the environment switches)
int gs_set_window_bits(int cbits) {
  if (cbits != 0 && (cbits < 8 || cbits > kMaxWindowBits)) return fail(GS_ERR_ARG, "window bits must be 0 (auto) or 8..%d", kMaxWindowBits);
  Registry& r = registry();
  std::lock_guard<std::mutex> rl(r.mu);
  if (r.ctxs.empty()) return fail(GS_ERR_NOT_INIT, "gs_init has not been called (or failed)");
  for (auto& pc : r.ctxs) {
    std::lock_guard<FairMutex> lk(pc->mu);
    pc->window_bits = cbits;
  }
  return GS_OK;
}

// Witness route of keys that carry an evaluation-basis array: 1 (default) = h-MSM over H's values, 0 = the coefficient route
// (interpolation + Taylor shift) every other key takes.  Same proofs; a switch for measurements and tests.  Every logical device.
int gs_set_eval_basis(int enabled) {
  Registry& r = registry();
  std::lock_guard<std::mutex> rl(r.mu);
  if (r.ctxs.empty()) return fail(GS_ERR_NOT_INIT, "gs_init has not been called (or failed)");
  for (auto& pc : r.ctxs) {
    std::lock_guard<FairMutex> lk(pc->mu);
    pc->eval_basis = enabled != 0;
  }
  return GS_OK;
}

}  // extern "C"
