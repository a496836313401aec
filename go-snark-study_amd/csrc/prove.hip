// Groth16 / Pinocchio provers on top of the MSM and polynomial engines, plus the extern "C"
// polynomial entry points.  Replaces groth16.GenerateProofs (groth16/groth16.go:225-278) and
// snark.GenerateProofs (snark.go:254-289).
#include "prove.h"

#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <chrono>
#include <future>
#include <vector>

#include "hostcopy.h"
#include "point_io.h"

using namespace gs;

namespace {

constexpr size_t kG1Aff = 16, kG2Aff = 32;      // u32 words per packed affine point

// copy n packed points [off, off+n) of a base handle into an owned buffer
void copy_points(Ctx& c, const Bases* b, size_t words, DevBuf& dst) {
  dst.alloc(std::max<size_t>(b->n, 1) * words * 4);
  if (b->n) GS_HIP(hipMemcpyAsync(dst.p, b->buf.p, b->n * words * 4, hipMemcpyDeviceToDevice, c.stream));
}

struct DevScalars {            // a scalar vector used by a prove call (not owned)
  const uint32_t* p;           // resident copy (device)
  size_t n;
  const uint64_t* host = nullptr;   // if set: `p` is a staging buffer that still has to be filled from here, on the stream
                                    // that consumes it (so the PCIe copy of px overlaps the accumulations over w)
  hipStream_t host_stream = nullptr;   // ... or (host-buffer tickets) on this stream, with the consumer waiting for `host_done`,
  hipEvent_t host_done = nullptr;      //     so that the DMA does not queue behind the previous proof's polynomial stage
  std::function<void(Ctx&)> produce;   // if set: `p` is an output buffer this call still has to compute, on the stream that
                                       // consumes it (px from the resident R1CS, behind the accumulations over w)
  std::function<bool(Ctx&, uint32_t*)> produce_hx;   // if set: try to compute hx = px / Z directly (poly.h: hx_direct_dev) into the given
                                                     // buffer; false = not applicable, fall back to produce + quotient
  // Evaluation-basis route (keys with ptd_eval / g1t_eval): if set, the h-MSM runs over H's VALUES at the nodes n+1..2n, which this
  // writes (n_eval of them) into the first buffer; the second is a device word that receives the number of violated constraints.
  // No host wait: the word is read when the proof is collected, and a non-zero count sends the caller down the exact route.
  std::function<void(Ctx&, uint32_t*, uint32_t*)> produce_hv;
  // ... or the values of this key slice's range [e_lo, e_lo + n_e) are already there (computed by another rank and scattered:
  // gs_groth16_prove_partials_values); n = n_e then
  const uint32_t* hv_slice = nullptr;
};

// Per-context staging of the prover entry points (device memory belongs to one device).
struct ProveState {
  DevBuf hx[Ctx::kSlots];                       // hx = floor(px / Z) -- or H's values on the evaluation-basis route --, one per slot (standard form)
  DevBuf up_w, up_px, up_a, up_b, up_o;         // uploads of host operands / results (blocking entry points only)
  DevBuf exact_px[Ctx::kSlots];                 // px of the exact witness route, one per slot (allocated only if that route is ever taken)
  // Host-buffer tickets (gs_*_host_begin): every in-flight slot owns the device copies of ITS w / px.  Grow-only, so a stream of
  // proofs from host memory allocates nothing after its first lap over the slots (gs_alloc_counters); a slot is re-used only after
  // its ticket was collected, i.e. after every device read of these buffers.
  DevBuf slot_w[Ctx::kSlots], slot_px[Ctx::kSlots];
};
ProveState& prove_state(Ctx& c) { return c.state<ProveState>(c.prove_state); }

size_t quotient_len(size_t npx, size_t nz) { return npx >= nz ? npx - nz + 1 : 0; }
// internal status of *_collect: the witness violates a constraint, so the optimistic evaluation-basis result is void --
// run the exact route (px, floor quotient) instead.  Never crosses the C ABI.
constexpr int kRetryExact = 1;

// Stream layout of one proof (all device work is enqueued before the host waits once):
//   main  : plan(w) -> accumulate G2 over w -> accumulate G1 x3 over w -> [wait plan(h)] accumulate G1 over h -> its tail
//   aux 0 / aux 2 : the tails (bucket combine, reduction, download) of the G2 / G1 groups over w, each after its accumulation
//   aux 1 : H(x) = P(x)/Z(x) -> plan(h)
// The ALU-bound accumulation kernels run back to back on one stream (they would only fight for the
// instruction cache if overlapped), and everything latency- or bandwidth-bound runs in their shadow.

// The five raw MSM results of a proof, before the O(1) tail.
struct GrothSums {
  G1Xyzz at, bacgamma1, bacdelta, h;
  G2Xyzz bacgamma2;
};

// Shard of the term ranges this call sums over (multi-GPU: rank k of N takes [k/N, (k+1)/N) of both ranges).
struct Shard { size_t index = 0, count = 1; };
static void shard_range(size_t n, const Shard& sh, size_t& lo, size_t& hi) {
  const size_t q = n / sh.count, rem = n % sh.count;
  lo = sh.index * q + std::min(sh.index, rem);
  hi = lo + q + (sh.index < rem ? 1 : 0);
}

void groth16_tail(GrothPkObj* pk, const GrothSums& sums, const uint64_t r[4], const uint64_t s[4], uint64_t out_proof[32], int inf[3]);

// One proof in flight.  Up to Ctx::kMaxInFlight may be outstanding (slots 0 / 1 / 2 own disjoint plan buffers, bucket workspaces and pinned result
// slots), so the plan and accumulations of proof k+1 are enqueued behind proof k's last accumulation and run while proof
// k's combine/reduce tails, its result download and its host-side tail are still in progress.
struct GrothTailPre {             // the tail products that need no MSM result (computed while the device runs)
  std::future<G2Xyzz> sdelta2;
  std::future<G1Xyzz> sdelta;
  G1Xyzz rdelta, rsdelta;
};
// The two result-dependent products of the tail, s piA and r piB1 (groth16.go:272-273), only need the sums over w -- which a
// proof has milliseconds before its sum over h: groth16_collect starts them (two host cores) as soon as the G1 group over w is
// folded, so that what is left when the device goes idle is a handful of additions and three normalisations.
struct GrothTailEarly {
  GrothPkObj* pk = nullptr;                // armed when set (with r, s, pre and the future that fills pre)
  const uint64_t* r = nullptr; const uint64_t* s = nullptr;
  GrothTailPre* pre = nullptr;
  std::future<void>* fpre = nullptr;
  bool done = false;
  G1Xyzz piA, piB1, sA, rB;
};
// GS_HOST_STAGE (scheduling only, same results): how a host-buffer ticket's arrays reach its slot.
//   0  copy stream + event: the kernels that read the vector wait for the copy's event on their own streams
//   1  copy stream, and the host waits for the copy to land before it enqueues the readers (what gs_scalars_update + _begin does)
//   2  on the stream of the first reader itself (w: the plan(w) stream, px: the polynomial stream); other readers wait for its event
// Measured at 2^20, three in flight, eight rotating witnesses (profiles/r05_ab_host_stage.txt; resident: 10.03-10.07 / px 8.84-9.01 ms):
//   witness from the host   0: 11.2-11.3   1: 10.12-10.17   2: 10.1-11.0        w + px from the host   0: 10.9   1: 9.45-9.51   2: 11.4-11.5
// The cross-stream event of mode 0 costs more than the host wait of mode 1 (the host has nothing else to do: the device is busy with the
// two tickets before this one), hence 1.
static long host_stage_mode() {
  static const long mode = run_knob("GS_HOST_STAGE", 1, 0, 2);
  return mode;
}
// w: staged at once (everything of a proof starts with plan(w))
// The streams ONE proof's device work is enqueued on, decided once per proof (round 4 toggled a context flag between two
// evaluations of an accessor instead).  plan(w) of a pipelined proof gets its own stream (aux 2, every tail then shares aux 0) where
// the polynomial stage is long: on aux 1 the NEXT proof's plan(w) queues behind this proof's H stage and plan(h), and the
// accumulation stream then waits ~0.3 ms twice per proof.  Measured (profiles/r04_ab_plan_w_stream.txt): witness route at 2^20
// 10.2-10.3 -> 9.8 ms, at 2^19 5.55 -> 5.17, px route at 2^22 35.9 -> 35.4 ms, but the px route at 2^20 (H(x) is only 0.7 ms there)
// 8.7-8.9 -> 9.0-9.1: hence the rule.  GS_PLANW_STREAM: 0 never, 1 this rule, 2 always (identical results).
struct ProofStreams {
  hipStream_t planw = nullptr, poly = nullptr, tail_g2 = nullptr, tail_g1 = nullptr;
  bool poly_first = false;      // enqueue H(x) and plan(h) right behind plan(w), in front of the accumulations' tails (slot layout below)
};
struct DevScalars;
// GS_SLOT_STREAMS (scheduling only, same results; round 6): 1 = a pipelined proof keeps ALL its side work -- plan(w), H(x), plan(h), then the
// three reduction tails -- on the aux stream of its ticket slot, so the side chains of the three proofs in flight run beside each
// other instead of queueing on one plan / polynomial stream.  For witnesses full of zeros and small values the accumulations of a 2^20
// proof are 2.5 ms but plan(w) + H(x) + plan(h) of consecutive proofs on ONE stream take 3.6 ms per proof beside them
// (profiles/r05_timeline_realistic_px_steady.txt: the accumulation stream idles 0.9 + 0.2 ms per proof waiting for a plan).  0 = the layouts above.
static ProofStreams proof_streams(Ctx& c, bool pipelined, size_t nterms, bool from_witness, int parity) {
  static const long mode = run_knob("GS_PLANW_STREAM", 1, 0, 2);
  static const long slot_mode = run_knob("GS_SLOT_STREAMS", 0, 0, 1);
  if (pipelined) c.next_tails((uint32_t)nterms);      // consecutive pipelined operations swap the two tail streams
  ProofStreams ps;
  if (pipelined && slot_mode == 1 && parity >= 0 && parity < Ctx::kAuxStreams && c.aux_stream[parity] != c.main_stream) {
    ps.planw = ps.poly = ps.tail_g2 = ps.tail_g1 = c.aux_stream[parity];
    ps.poly_first = true;
    return ps;
  }
  const bool own = pipelined && c.aux_stream[2] != c.aux_stream[1] &&
                   (mode == 2 || (mode == 1 && nterms >= ((size_t)1 << 19) && (from_witness || nterms >= ((size_t)1 << 21))));
  ps.poly = c.aux_stream[1];
  ps.planw = own ? c.aux_stream[2] : c.aux_stream[1];
  ps.tail_g2 = own ? c.aux_stream[0] : c.tail_stream(0);
  ps.tail_g1 = own ? c.aux_stream[0] : c.tail_stream(1);
  return ps;
}

// Inputs that arrive in host memory with a ticket (gs_*_host_begin): staged on the copy stream into the slot's own buffers.
struct HostInputs {
  hipEvent_t w_done = nullptr, px_done = nullptr;       // recorded on the copy stream behind the last DMA of w / px
  std::shared_ptr<PhaseTimer> th2d;                     // read when the ticket is collected (a read here would wait for the copy)
  void create() {
    if (!w_done) GS_HIP(hipEventCreateWithFlags(&w_done, hipEventDisableTiming));
    if (!px_done) GS_HIP(hipEventCreateWithFlags(&px_done, hipEventDisableTiming));
  }
  ~HostInputs() { for (hipEvent_t e : {w_done, px_done}) if (e) (void)hipEventDestroy(e); }
};

// the part of a term list that falls into the held range [base, base + nterms): pointer into the device list and count
static const uint32_t* term_list_range(const DevBuf& index_dev, const std::vector<uint32_t>& index_host, size_t base, size_t nterms, uint32_t& count) {
  const auto lo = std::lower_bound(index_host.begin(), index_host.end(), (uint32_t)base);
  const auto hi = std::lower_bound(lo, index_host.end(), (uint32_t)(base + nterms));
  count = (uint32_t)(hi - lo);
  return index_dev.as<uint32_t>() + (lo - index_host.begin());
}

struct GrothInFlight : InFlightBase {
  GrothPkObj* pk = nullptr;
  ProofStreams streams;
  HostInputs in;
  uint64_t r[4] = {0, 0, 0, 0}, s[4] = {0, 0, 0, 0};
  bool with_tail = false;
  GrothTailEarly early;
  // done_g2 / done_g1w / done_h: one per MSM group, recorded behind that group's reduction tail, so the host can add up a
  // group's partial sums while the later groups are still on the device
  hipEvent_t planw = nullptr, planh = nullptr, done_main = nullptr, done_g2 = nullptr, done_g1w = nullptr, done_h = nullptr;
  // keys with sparse B arrays (prove.h, b_index): B1 and B2 run over a plan over the listed terms of their own; pend_g1w then carries At and BACDelta
  // only and pend_g1b the sum over G1.BACGamma
  bool split_b = false;
  hipEvent_t planb = nullptr, done_g1b = nullptr;
  MsmPending pend_g1b;
  std::shared_ptr<PhaseTimer> tplanb;
  std::unique_ptr<PhaseTimer> total;
  MsmPending pend_g1w, pend_g2w, pend_h;
  std::shared_ptr<PhaseTimer> tpoly, tplanw, tplanh;
  GrothTailPre pre;
  std::future<void> fpre;
  const uint32_t* bad_host = nullptr;      // evaluation-basis route: the violated-constraint count (valid once done_h has fired)
  std::function<int(Ctx&, GrothSums&)> exact_route;   // ... and what to run instead when it is not zero (witness tickets)
  GrothInFlight() {
    GS_HIP(hipEventCreateWithFlags(&planw, hipEventDisableTiming));
    GS_HIP(hipEventCreateWithFlags(&planh, hipEventDisableTiming));
    for (hipEvent_t* e : {&done_main, &done_g2, &done_g1w, &done_h, &planb, &done_g1b}) GS_HIP(hipEventCreateWithFlags(e, hipEventDisableTiming));
  }
  void wait_device() const override {
    for (hipEvent_t e : {done_g2, done_g1w, done_h, done_main}) if (e) GS_HIP(hipEventSynchronize(e));
    if (split_b && done_g1b) GS_HIP(hipEventSynchronize(done_g1b));
  }
  ~GrothInFlight() override {
    if (fpre.valid()) fpre.wait();
    for (hipEvent_t e : {planw, planh, done_main, done_g2, done_g1w, done_h, planb, done_g1b}) if (e) (void)hipEventDestroy(e);
  }
};

static void groth16_tail_pre(GrothPkObj* pk, const uint64_t r[4], const uint64_t s[4], GrothTailPre& pre);

// Enqueue every device operation of one proof (no host wait).  `wait_inputs`: w / px were uploaded on the main stream
// in this call, so the aux streams must order themselves behind that point.
int groth16_enqueue(Ctx& c, GrothPkObj* pk, DevScalars w, DevScalars px, const Shard& shard, int parity, bool wait_inputs, bool pipelined,
                    GrothInFlight& st) {
  DevBuf& hxbuf = prove_state(c).hx[parity];
  if (w.n != pk->nvars) return fail(GS_ERR_SHAPE, "len(w) = %zu but the key has %zu variables", w.n, pk->nvars);
  const bool eval = (bool)px.produce_hv || px.hv_slice;   // h-MSM over H's values against the evaluation-basis table (the caller checked the key has one)
  const size_t nh = eval ? pk->n_eval : quotient_len(px.n, pk->nz);
  if (!eval && nh > pk->nptd)
    return fail(GS_ERR_SHAPE, "len(hx) = len(px) - len(Z) + 1 = %zu exceeds len(PowersTauDelta) = %zu (groth16.go:269-271)", nh, pk->nptd);
  const bool sliced = pk->shard_count > 1;
  if (sliced && (shard.index != pk->shard_index || shard.count != pk->shard_count))
    return fail(GS_ERR_ARG, "this key holds shard %zu of %zu of the term ranges only: call gs_groth16_prove_partials with that shard "
                "(asked for %zu of %zu)", pk->shard_index, pk->shard_count, shard.index, shard.count);
  size_t wlo, whi, hlo, hhi;
  shard_range(w.n, shard, wlo, whi);
  if (sliced && eval) { hlo = pk->e_lo; hhi = pk->e_lo + pk->n_e; }
  else if (sliced) {                // a slice's PowersTauDelta range was fixed at key creation (split of len(PTD), clipped to len(hx))
    hlo = std::min(pk->h_lo, nh);
    hhi = std::min(pk->h_lo + pk->n_h, nh);
  } else shard_range(nh, shard, hlo, hhi);
  const size_t held_lo = eval ? pk->e_lo : pk->h_lo;
  const size_t wbase = wlo - pk->w_lo, hbase = hlo - std::min(held_lo, hlo);      // offsets into the arrays this key holds
  // window tables or table-free, per plan (msm.h, prepare_tables: the four arrays over w share one plan, so they go one way together)
  // (before any table instalment goes onto the main stream: the plans on the aux streams wait for the inputs, not for the slabs)
  if (wait_inputs) {
    hipEvent_t start;
    GS_HIP(hipEventCreateWithFlags(&start, hipEventDisableTiming));
    GS_HIP(hipEventRecord(start, c.main_stream));
    for (auto a : c.aux_stream) GS_HIP(hipStreamWaitEvent(a, start, 0));
    GS_HIP(hipEventDestroy(start));
  }
  int cw = 0, ch = 0;
  // every table of the call is stamped before one is built (an allocation for the first group must not evict the second group's), and
  // under policy `auto` the call grants itself a build credit for its ~6.8 job-units over w and h (msm.h, prepare_tables)
  stamp_tables(c, {&pk->t_at, &pk->t_bacgamma1, &pk->t_bacdelta, &pk->t_bacgamma2, eval ? &pk->t_ptd_eval : &pk->t_ptd});
  double credit = build_credit(3.0 + 2.76 + 1.0, whi - wlo);
  const bool tab_w = prepare_tables(c, {TableRef{&pk->t_at, pk->at.as<uint32_t>(), pk->n_w, false}, TableRef{&pk->t_bacgamma1, pk->bacgamma1.as<uint32_t>(), pk->n_w, false},
                                        TableRef{&pk->t_bacdelta, pk->bacdelta.as<uint32_t>(), pk->n_w, false},
                                        TableRef{&pk->t_bacgamma2, pk->bacgamma2.as<uint32_t>(), pk->n_w, true}}, (uint32_t)(whi - wlo), &cw, &credit);
  const bool tab_h = prepare_tables(c, {eval ? TableRef{&pk->t_ptd_eval, pk->ptd_eval.as<uint32_t>(), pk->n_e, false}
                                             : TableRef{&pk->t_ptd, pk->ptd.as<uint32_t>(), pk->n_h, false}}, (uint32_t)(hhi - hlo), &ch, &credit);
  hxbuf.ensure(std::max<size_t>(nh, 1) * 32);
  auto base_w = [&](BaseTable& t, const DevBuf& pts) { return MsmBase{&t, wbase, pts.as<uint32_t>(), pk->n_w}; };
  st.pk = pk;
  st.total = std::make_unique<PhaseTimer>(c.main_stream);
  const int ws = 8 * parity, pin = 3 * parity;
  MsmPlan plan_w, plan_h;
  // The main stream carries NOTHING but the ALU-bound bucket accumulations (G2, then the three G1 arrays over w, then h),
  // so with two proofs in flight it never idles: both plans, H(x) and every combine/reduce tail run on the aux streams.
  const ProofStreams ps = proof_streams(c, pipelined, whi - wlo, (bool)px.produce_hv || (bool)px.produce_hx || (bool)px.produce, parity);   // (not hv_slice: no polynomial work here)
  st.streams = ps;
  if (w.host && w.host_done) {                                   // host-buffer ticket, GS_HOST_STAGE=2: w is copied on the plan(w) stream itself
    staged_h2d(c, const_cast<uint32_t*>(w.p), w.host, w.n * 32, ps.planw);
    GS_HIP(hipEventRecord(w.host_done, ps.planw));
    if (ps.poly != ps.planw) GS_HIP(hipStreamWaitEvent(ps.poly, w.host_done, 0));
  } else if (w.host_done) {                                      // w was staged on the copy stream: its readers wait for the DMA
    GS_HIP(hipStreamWaitEvent(ps.planw, w.host_done, 0));
    if (ps.poly != ps.planw) GS_HIP(hipStreamWaitEvent(ps.poly, w.host_done, 0));
  }
  // Sparse B (prove.h, b_index): when fewer than 55 % of the key's variables have a B entry, B1 and B2 are summed over a second plan of
  // w without the others -- one more sort (~0.35 ms at 2^20) against that share of 3.8 of the proof's 6.8 job-units.  Measured at 2^20
  // (profiles/r05_ab_sparse_b_split.txt): a circuit of the reference compiler's shape (33 % of the variables in B, full-width witness)
  // 8.3-8.4 -> 6.1-6.2 ms per proof in flight, 8.8 -> 6.9 blocking, witness route 9.15 -> 7.2; the realistic-witness instance (60 % in
  // B, and the absent variables are the small ones that carry one or two digits anyway) LOSES 3 % -- hence 55, not 90.  (Window
  // widths of 18 and more -- only by gs_set_window_bits -- keep the single plan: their partition-first sort takes no term list.  ADVICE r5:
  // the gate used to say `cw < 19` while build_plan partitions from c = 18 on, so a forced width of 18 on a sparse-B key failed every proof.)
  static const long split_pct = run_knob("GS_SPLIT_B_PERCENT", 55, 0, 100);      // split below this share of finite B points; 0 = never (same results)
  const size_t nterms_w = whi - wlo;
  st.split_b = pk->b_index.p != nullptr && nterms_w >= 4096 && !plan_partitions_first(cw) && pk->b_finite * 100 < (size_t)split_pct * pk->n_w;
  MsmPlan plan_b;
  {                                                              // aux 1 (or aux 2, proof_streams): plan(w) [, the plan over the listed terms for B first: G2 starts the proof]
    StreamScope sc(c, ps.planw);
    if (st.split_b) {
      st.tplanb = std::make_shared<PhaseTimer>(c.stream);
      uint32_t nb = 0;                                             // the listed terms of this call's range only
      const uint32_t* list = term_list_range(pk->b_index, pk->b_index_host, wbase, nterms_w, nb);
      build_plan(c, 2 * Ctx::kSlots + parity, w.p + wlo * 8, nb, plan_b, {{1, true}, {1, false}}, cw, !tab_w, list, (uint32_t)wbase);
      st.tplanb->stop();
      GS_HIP(hipEventRecord(st.planb, c.stream));
    }
    st.tplanw = std::make_shared<PhaseTimer>(c.stream);
    build_plan(c, 0 + 2 * parity, w.p + wlo * 8, (uint32_t)nterms_w, plan_w, st.split_b ? std::vector<LaunchShape>{{2, false}} : std::vector<LaunchShape>{{1, true}, {3, false}},
               cw, !tab_w);
    st.tplanw->stop();
    GS_HIP(hipEventRecord(st.planw, c.stream));
  }
  // Host order: the accumulations over w are enqueued BEFORE the block that may copy px from pageable host memory -- that copy
  // stages through the runtime inside the call (~5 ms for 64 MiB), and the device must already have its 7 ms of work by then.
  // (Slot layout, ps.poly_first: everything of the proof's side chain first -- its tails follow on the same stream.)
  auto enqueue_acc_w = [&] {                                     // main: the accumulations over w, back to back
    StreamScope sc(c, c.main_stream);
    GS_HIP(hipStreamWaitEvent(c.stream, st.split_b ? st.planb : st.planw, 0));
    // BACDelta's first npublic+1 entries are forced to infinity at key creation, so the reference's
    // loop from NPublic+1 (:248-250) equals a full-range MSM sharing w's plan.
    // G2 first: its accumulation (2 waves/SIMD, 256 VGPRs) is the one a foreign wave hurts most, and its long
    // combine/reduce tail then hides behind the G1 accumulations.
    msm_enqueue_g2(c, st.split_b ? plan_b : plan_w, {base_w(pk->t_bacgamma2, pk->bacgamma2)}, ws + 4, pin + 1, st.pend_g2w, ps.tail_g2);
    GS_HIP(hipEventRecord(st.done_g2, ps.tail_g2));
    if (st.split_b) {
      msm_enqueue_g1(c, plan_b, {base_w(pk->t_bacgamma1, pk->bacgamma1)}, ws + 2, 3 * Ctx::kSlots + parity, st.pend_g1b, ps.tail_g1);
      GS_HIP(hipEventRecord(st.done_g1b, ps.tail_g1));
      GS_HIP(hipStreamWaitEvent(c.stream, st.planw, 0));
      msm_enqueue_g1(c, plan_w, {base_w(pk->t_at, pk->at), base_w(pk->t_bacdelta, pk->bacdelta)}, ws + 0, pin + 0, st.pend_g1w, ps.tail_g1);
    } else {
      msm_enqueue_g1(c, plan_w, {base_w(pk->t_at, pk->at), base_w(pk->t_bacgamma1, pk->bacgamma1), base_w(pk->t_bacdelta, pk->bacdelta)}, ws + 0, pin + 0,
                     st.pend_g1w, ps.tail_g1);
    }
    GS_HIP(hipEventRecord(st.done_g1w, ps.tail_g1));
  };
  // (Starting H(x) and plan(h) of a lone proof beside plan(w) on another stream instead of behind it was tried: the blocking proof
  // got SLOWER, 11.5-11.7 vs 11.0-11.25 ms -- the NTT passes then overlap the G2 accumulation's first milliseconds more densely.)
  auto enqueue_poly = [&] {                                      // aux 1 again: (late upload of px,) H(x), plan(h)
    StreamScope sc(c, ps.poly);
    if (px.host && px.n && px.host_stream) {   // host-buffer ticket: on the copy stream, beside whatever aux 1 still carries of the previous proof
      staged_h2d(c, const_cast<uint32_t*>(px.p), px.host, px.n * 32, px.host_stream);
      if (host_stage_mode() == 1) GS_HIP(hipStreamSynchronize(px.host_stream));
      else {
        GS_HIP(hipEventRecord(px.host_done, px.host_stream));
        GS_HIP(hipStreamWaitEvent(c.stream, px.host_done, 0));
      }
    } else if (px.host && px.n && pipelined) {      // host-buffer ticket, GS_HOST_STAGE=2: in stream order on the polynomial stream, no host wait
      staged_h2d(c, const_cast<uint32_t*>(px.p), px.host, px.n * 32, c.stream);
    } else if (px.host && px.n) {     // the device is already busy with ~8 ms of accumulations: this copy is off the critical path
      PhaseTimer th(c.stream);
      staged_h2d(c, const_cast<uint32_t*>(px.p), px.host, px.n * 32, c.stream);
      th.stop();
      c.timing.h2d_ms += th.ms();
    }
    st.tpoly = std::make_shared<PhaseTimer>(c.stream);
    bool have_hx = false;
    if (px.hv_slice) have_hx = true;                                           // H's values came from another rank
    else if (eval) {                                                           // H's values, for the evaluation-basis table
      px.produce_hv(c, hxbuf.as<uint32_t>(), c.bad_dev.as<uint32_t>() + parity);
      GS_HIP(hipMemcpyAsync(c.bad_host + parity, c.bad_dev.as<uint32_t>() + parity, 4, hipMemcpyDeviceToHost, c.stream));
      st.bad_host = c.bad_host + parity;
      have_hx = true;
    }
    if (!have_hx && px.produce_hx && nh) have_hx = px.produce_hx(c, hxbuf.as<uint32_t>());  // H from the constraint values (satisfying witness)
    if (!have_hx) {
      if (px.produce) px.produce(c);                                           // r1csqap.go:161-210 on the sparse system
      if (nh) poly_quotient_dev(c, pk->z, px.p, px.n, hxbuf.as<uint32_t>());    // groth16.go:266
    }
    st.tpoly->stop();
    st.tplanh = std::make_shared<PhaseTimer>(c.stream);
    build_plan(c, 1 + 2 * parity, px.hv_slice ? px.hv_slice : hxbuf.as<uint32_t>() + hlo * 8, (uint32_t)(hhi - hlo), plan_h, {{1, false}}, ch, !tab_h);
    st.tplanh->stop();
    GS_HIP(hipEventRecord(st.planh, c.stream));
  };
  if (ps.poly_first) { enqueue_poly(); enqueue_acc_w(); } else { enqueue_acc_w(); enqueue_poly(); }
  {                                                              // main again: the accumulation over h
    StreamScope sc(c, c.main_stream);
    GS_HIP(hipStreamWaitEvent(c.stream, st.planh, 0));
    // a lone proof finishes soonest with the last tail right behind its accumulation; in a pipeline that tail must not sit
    // in front of the next proof's accumulations
    msm_enqueue_g1(c, plan_h, {eval ? MsmBase{&pk->t_ptd_eval, hbase, pk->ptd_eval.as<uint32_t>(), pk->n_e} : MsmBase{&pk->t_ptd, hbase, pk->ptd.as<uint32_t>(), pk->n_h}},
                   ws + 3, pin + 2, st.pend_h, pipelined ? ps.tail_g1 : nullptr);   // :269-271
    GS_HIP(hipEventRecord(st.done_h, pipelined ? ps.tail_g1 : c.main_stream));
  }
  st.total->stop();
  GS_HIP(hipEventRecord(st.done_main, c.main_stream));
  return GS_OK;
}

// Wait for that proof's device work (only its own events: later proofs keep running) and add up the results.
static double host_now_ms() {
  return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count();
}
static bool host_trace() { static const bool on = run_flag("GS_HOST_TRACE"); return on; }

int groth16_collect(Ctx& c, GrothInFlight& st, GrothSums& sums) {
  const double t0 = host_trace() ? host_now_ms() : 0;
  double t1 = 0;
  std::vector<G1Xyzz> g1w, g1h;
  std::vector<G2Xyzz> g2w;
  {   // The host-side pair sums of the three groups, each as soon as its own tail is through: the groups over w finish milliseconds
      // before the one over h, so only h's handful of additions is left when the device goes idle.
    // (HIP calls stay on the calling thread: a fresh thread pays ~0.3 ms for its first hipSetDevice / event wait.)
    GS_HIP(hipEventSynchronize(st.done_g2));
    auto f2 = std::async(std::launch::async, [&] { msm_finish_g2(c, st.pend_g2w, g2w); });
    struct Join { std::future<void>& f; ~Join() { if (f.valid()) f.wait(); } } j2{f2};     // a throwing wait below must not outrun the threads
    GS_HIP(hipEventSynchronize(st.done_g1w));       // (the B1 group of a split proof was enqueued before this one on the same tail stream)
    auto f1 = std::async(std::launch::async, [&] {
      msm_finish_g1(c, st.pend_g1w, g1w);
      if (st.split_b) {                              // g1w = [At, BACDelta]: bring it into the order [At, BACGamma, BACDelta] of the single plan
        std::vector<G1Xyzz> g1b;
        msm_finish_g1(c, st.pend_g1b, g1b);
        g1w.insert(g1w.begin() + 1, g1b[0]);
      }
      GrothTailEarly& e = st.early;
      if (!e.pk) return;
      e.fpre->wait();                                            // r delta, s delta (the owner calls get() after this thread is joined)
      e.piA = g1w[0];
      xyzz_madd(e.piA, e.pk->alpha);                             // + alpha          groth16.go:253
      xyzz_add(e.piA, e.pre->rdelta);                            // + r delta        :254-255
      e.piB1 = g1w[1];
      xyzz_madd(e.piB1, e.pk->beta);                             // + beta           :259
      xyzz_add(e.piB1, e.pre->sdelta.get());                     // + s delta        :261-262
      const G1Xyzz b1 = e.piB1;
      const uint64_t* r = e.r;
      auto f_rB = std::async(std::launch::async, [b1, r] { return g1_mul_scalar(b1, r); });
      e.sA = g1_mul_scalar(e.piA, e.s);                          // s piA            :272
      e.rB = f_rB.get();                                         // r piB1           :273
      e.done = true;
    });
    Join j1{f1};
    GS_HIP(hipEventSynchronize(st.done_h));
    t1 = host_trace() ? host_now_ms() : 0;
    msm_finish_g1(c, st.pend_h, g1h);
    f2.get(); f1.get();
  }
  GS_HIP(hipEventSynchronize(st.done_main));
  if (host_trace()) fprintf(stderr, "[gs host] collect: wait %.3f ms, fold %.3f ms\n", t1 - t0, host_now_ms() - t1);
  msm_book_timing(c, st.pend_g1w); msm_book_timing(c, st.pend_g2w); msm_book_timing(c, st.pend_h);
  if (st.split_b) { msm_book_timing(c, st.pend_g1b); c.timing.plan_ms += st.tplanb->ms(); }
  c.timing.poly_ms += st.tpoly->ms();
  c.timing.plan_ms += st.tplanw->ms() + st.tplanh->ms();
  c.timing.total_ms += st.total->ms();
  if (st.in.th2d) c.timing.h2d_ms += st.in.th2d->ms();
  sums.at = g1w[0]; sums.bacgamma1 = g1w[1]; sums.bacdelta = g1w[2]; sums.h = g1h[0]; sums.bacgamma2 = g2w[0];
  // evaluation-basis route: H's values only determine H when A B - C vanishes at every root of Z
  if (st.bad_host && *st.bad_host != 0) return kRetryExact;
  return GS_OK;
}

int groth16_sums_impl(Ctx& c, GrothPkObj* pk, DevScalars w, DevScalars px, const Shard& shard, GrothSums& sums) {
  for (;;) {
    GrothInFlight st;
    int rc = groth16_enqueue(c, pk, w, px, shard, c.blocking_slot(), true, false, st);
    if (rc != GS_OK) return rc;
    rc = groth16_collect(c, st, sums);
    if (rc != kRetryExact) return rc;
    px.produce_hv = nullptr;                   // violated constraint: once more through px and the floor quotient
    c.timing.fallbacks += 1;
  }
}

// --- the O(1) tail on host cores (groth16.go:253-275) ----------------------------------------------------
static void groth16_tail_pre(GrothPkObj* pk, const uint64_t r[4], const uint64_t s[4], GrothTailPre& pre) {
  // r delta, s delta, s delta2, (r s) delta: multiples of points that belong to the KEY -- fixed-base window tables (msm.h), built by
  // whichever proof of the key comes first; ~64 additions each instead of 256 doublings + 64 additions
  std::call_once(pk->fixed_once, [pk] { pk->delta_fixed.build(pk->delta); pk->delta2_fixed.build(pk->delta2); });
  uint64_t s_copy[4] = {s[0], s[1], s[2], s[3]};
  pre.sdelta2 = std::async(std::launch::async, [pk, s_copy] { return pk->delta2_fixed.mul(s_copy); });
  pre.sdelta = std::async(std::launch::deferred, [pk, s_copy] { return pk->delta_fixed.mul(s_copy); });   // ~60 us: no thread of its own
  pre.rdelta = pk->delta_fixed.mul(r);
  uint64_t rs[4];
  fr_mul_words(r, s, rs);
  pre.rsdelta = pk->delta_fixed.mul(rs);                       // (r s) delta = s (r delta)
}
static void groth16_tail_post(GrothPkObj* pk, const GrothSums& sums, GrothTailPre& pre, const uint64_t r[4], const uint64_t s[4],
                              uint64_t out_proof[32], int inf[3], const GrothTailEarly* early = nullptr) {
  const bool have = early && early->done;                      // piA, piB1, s piA, r piB1 came from groth16_collect (same sums)
  G1Xyzz piA, piB1, sA, rB;
  G2Xyzz piB = sums.bacgamma2;
  xyzz_madd(piB, pk->beta2);                                   // + beta2          :260
  xyzz_add(piB, pre.sdelta2.get());                            // + s delta2       :263-264
  G1Xyzz piC = sums.bacdelta;
  xyzz_add(piC, sums.h);                                       // + sum h_i PTD_i  :269-271
  if (have) { piA = early->piA; piB1 = early->piB1; sA = early->sA; rB = early->rB; }
  else {
    piA = sums.at;
    xyzz_madd(piA, pk->alpha);                                 // + alpha          :253
    xyzz_add(piA, pre.rdelta);                                 // + r delta        :254-255
    piB1 = sums.bacgamma1;
    xyzz_madd(piB1, pk->beta);                                 // + beta           :259
    xyzz_add(piB1, pre.sdelta.get());                          // + s delta        :261-262
    auto f_rB = std::async(std::launch::async, [piB1, r] { return g1_mul_scalar(piB1, r); });   // two host cores for the two
    sA = g1_mul_scalar(piA, s);                                                                 // result-dependent products
    rB = f_rB.get();
  }
  xyzz_add(piC, sA);                                           // + s piA          :272
  xyzz_add(piC, rB);                                           // + r piB1         :273
  xyzz_add(piC, xyzz_neg(pre.rsdelta));                        // - (r s) delta    :274-275
  inf[0] = g1_to_affine_std(piA, out_proof) ? 1 : 0;
  inf[1] = g2_to_affine_std(piB, out_proof + 8) ? 1 : 0;
  inf[2] = g1_to_affine_std(piC, out_proof + 24) ? 1 : 0;
}
void groth16_tail(GrothPkObj* pk, const GrothSums& sums, const uint64_t r[4], const uint64_t s[4], uint64_t out_proof[32], int inf[3]) {
  GrothTailPre pre;
  groth16_tail_pre(pk, r, s, pre);
  groth16_tail_post(pk, sums, pre, r, s, out_proof, inf);
}

int groth16_prove_impl(Ctx& c, GrothPkObj* pk, DevScalars w, DevScalars px, const uint64_t r[4], const uint64_t s[4],
                       uint64_t out_proof[32], int inf[3]) {
  // the result-independent tail products run on other host cores while this thread enqueues and waits for the device
  const double t0 = host_trace() ? host_now_ms() : 0;
  GrothTailPre pre;
  std::future<void> fpre = std::async(std::launch::async, [&] { groth16_tail_pre(pk, r, s, pre); });
  GrothSums sums;
  int rc;
  double t1 = 0;
  GrothTailEarly early;
  {
    GrothInFlight st;
    st.early.pk = pk; st.early.r = r; st.early.s = s; st.early.pre = &pre; st.early.fpre = &fpre;
    rc = groth16_enqueue(c, pk, w, px, Shard{}, c.blocking_slot(), true, false, st);
    t1 = host_trace() ? host_now_ms() : 0;
    if (rc == GS_OK) rc = groth16_collect(c, st, sums);
    early = st.early;
  }
  if (rc == kRetryExact) {                     // violated constraint on the evaluation-basis route: the exact route, blocking
    px.produce_hv = nullptr;
    c.timing.fallbacks += 1;
    // (the sums over w do not depend on the route: the early products stay valid)
    rc = groth16_sums_impl(c, pk, w, px, Shard{}, sums);
  }
  const double t2 = host_trace() ? host_now_ms() : 0;
  fpre.get();
  if (rc != GS_OK) return rc;
  groth16_tail_post(pk, sums, pre, r, s, out_proof, inf, &early);
  if (host_trace()) fprintf(stderr, "[gs host] prove: enqueue %.3f ms, collect %.3f ms, tail %.3f ms (entered at %.3f)\n", t1 - t0, t2 - t1, host_now_ms() - t2, t0);
  return GS_OK;
}

// One Pinocchio proof in flight: same stream layout as a Groth16 proof (main = accumulations only).
struct PinInFlight : InFlightBase {
  ProofStreams streams;
  HostInputs in;
  hipEvent_t planw = nullptr, planh = nullptr, done_main = nullptr, done_g2 = nullptr, done_g1w = nullptr, done_h = nullptr;   // as GrothInFlight
  bool split_b = false;                                     // as GrothInFlight: B (G2) and B' over the plan over the listed terms, the five other G1 sums over w's
  hipEvent_t planb = nullptr, done_g1b = nullptr;
  MsmPending pend_g1b;
  std::shared_ptr<PhaseTimer> tplanb;
  std::unique_ptr<PhaseTimer> total;
  MsmPending pend_g1w, pend_g2w, pend_h;
  std::shared_ptr<PhaseTimer> tpoly, tplanw, tplanh;
  const uint32_t* bad_host = nullptr;                       // as GrothInFlight
  std::function<int(Ctx&, uint64_t*, int*)> exact_route;
  PinInFlight() {
    for (hipEvent_t* e : {&planw, &planh, &done_main, &done_g2, &done_g1w, &done_h, &planb, &done_g1b}) GS_HIP(hipEventCreateWithFlags(e, hipEventDisableTiming));
  }
  void wait_device() const override {
    for (hipEvent_t e : {done_g2, done_g1w, done_h, done_main}) if (e) GS_HIP(hipEventSynchronize(e));
    if (split_b && done_g1b) GS_HIP(hipEventSynchronize(done_g1b));
  }
  ~PinInFlight() override {
    for (hipEvent_t e : {planw, planh, done_main, done_g2, done_g1w, done_h, planb, done_g1b}) if (e) (void)hipEventDestroy(e);
  }
};

// snark.GenerateProofs (snark.go:254-289): six G1 sums over w sharing one plan, one G2 sum over w, H(x) = px / Z, one G1 sum
// over h.  Ticket `parity` owns plan slots 2p / 2p + 1, workspace sets 8p .. 8p + 7 and pinned slots 3p .. 3p + 2.
// `shard`: the term ranges this call sums over (several GPUs: rank k of N; the eight sums of the ranks add up to the proof).
int pinocchio_enqueue(Ctx& c, PinocchioPkObj* pk, DevScalars w, DevScalars px, const Shard& shard, int parity, bool wait_inputs, bool pipelined,
                      PinInFlight& st) {
  DevBuf& hxbuf = prove_state(c).hx[parity];
  if (w.n != pk->nvars) return fail(GS_ERR_SHAPE, "len(w) = %zu but the key has %zu variables", w.n, pk->nvars);
  const bool eval = (bool)px.produce_hv || px.hv_slice;   // h-MSM over H's values against g1t_eval (the caller checked the key has one)
  const size_t nh = eval ? pk->n_eval : quotient_len(px.n, pk->nz);
  if (!eval && nh > pk->ng1t) return fail(GS_ERR_SHAPE, "len(hx) = %zu exceeds len(G1T) = %zu (snark.go:284-286)", nh, pk->ng1t);
  const bool sliced = pk->shard_count > 1;
  if (sliced && (shard.index != pk->shard_index || shard.count != pk->shard_count))
    return fail(GS_ERR_ARG, "this key holds shard %zu of %zu of the term ranges only: call gs_pinocchio_prove_partials with that shard "
                "(asked for %zu of %zu)", pk->shard_index, pk->shard_count, shard.index, shard.count);
  size_t wlo, whi, hlo, hhi;                     // exactly groth16_enqueue's ranges
  shard_range(w.n, shard, wlo, whi);
  if (sliced && eval) { hlo = pk->e_lo; hhi = pk->e_lo + pk->n_e; }
  else if (sliced) { hlo = std::min(pk->h_lo, nh); hhi = std::min(pk->h_lo + pk->n_h, nh); }
  else shard_range(nh, shard, hlo, hhi);
  const size_t held_lo = eval ? pk->e_lo : pk->h_lo;
  const size_t wbase = wlo - pk->w_lo, hbase = hlo - std::min(held_lo, hlo);
  if (wait_inputs) {
    hipEvent_t start;
    GS_HIP(hipEventCreateWithFlags(&start, hipEventDisableTiming));
    GS_HIP(hipEventRecord(start, c.main_stream));
    for (auto a : c.aux_stream) GS_HIP(hipStreamWaitEvent(a, start, 0));
    GS_HIP(hipEventDestroy(start));
  }
  int cw = 0, ch = 0;                          // window tables or table-free, per plan (as groth16_enqueue)
  auto ref1 = [&](BaseTable& t, const DevBuf& pts) { return TableRef{&t, pts.as<uint32_t>(), pk->n_w, false}; };
  stamp_tables(c, {&pk->t_a, &pk->t_ap, &pk->t_bp, &pk->t_c, &pk->t_cp, &pk->t_kp, &pk->t_b2, eval ? &pk->t_g1t_eval : &pk->t_g1t});      // as groth16_enqueue
  double credit = build_credit(6.0 + 2.76 + 1.0, whi - wlo);
  const bool tab_w = prepare_tables(c, {ref1(pk->t_a, pk->a), ref1(pk->t_ap, pk->ap), ref1(pk->t_bp, pk->bp), ref1(pk->t_c, pk->c), ref1(pk->t_cp, pk->cp),
                                        ref1(pk->t_kp, pk->kp), TableRef{&pk->t_b2, pk->b2.as<uint32_t>(), pk->n_w, true}}, (uint32_t)(whi - wlo), &cw, &credit);
  const bool tab_h = prepare_tables(c, {eval ? TableRef{&pk->t_g1t_eval, pk->g1t_eval.as<uint32_t>(), pk->n_e, false}
                                             : TableRef{&pk->t_g1t, pk->g1t.as<uint32_t>(), pk->n_h, false}}, (uint32_t)(hhi - hlo), &ch, &credit);
  hxbuf.ensure(std::max<size_t>(nh, 1) * 32);
  auto base_w = [&](BaseTable& t, const DevBuf& pts) { return MsmBase{&t, wbase, pts.as<uint32_t>(), pk->n_w}; };
  st.total = std::make_unique<PhaseTimer>(c.main_stream);
  const int ws = 8 * parity, pin = 3 * parity;
  MsmPlan plan_w, plan_h;
  const ProofStreams ps = proof_streams(c, pipelined, whi - wlo, (bool)px.produce_hv || (bool)px.produce_hx || (bool)px.produce, parity);
  st.streams = ps;
  if (w.host && w.host_done) {                                   // host-buffer ticket, GS_HOST_STAGE=2: w is copied on the plan(w) stream itself
    staged_h2d(c, const_cast<uint32_t*>(w.p), w.host, w.n * 32, ps.planw);
    GS_HIP(hipEventRecord(w.host_done, ps.planw));
    if (ps.poly != ps.planw) GS_HIP(hipStreamWaitEvent(ps.poly, w.host_done, 0));
  } else if (w.host_done) {                                      // w was staged on the copy stream: its readers wait for the DMA
    GS_HIP(hipStreamWaitEvent(ps.planw, w.host_done, 0));
    if (ps.poly != ps.planw) GS_HIP(hipStreamWaitEvent(ps.poly, w.host_done, 0));
  }
  static const long split_pct = run_knob("GS_SPLIT_B_PERCENT", 55, 0, 100);      // as groth16_enqueue
  const size_t nterms_w = whi - wlo;
  st.split_b = pk->b_index.p != nullptr && nterms_w >= 4096 && !plan_partitions_first(cw) && pk->b_finite * 100 < (size_t)split_pct * pk->n_w;
  MsmPlan plan_b;
  {                                                              // aux 1 (or its own stream, proof_streams): [the plan over the listed terms for B, B',] plan(w)
    StreamScope sc(c, ps.planw);
    if (st.split_b) {
      st.tplanb = std::make_shared<PhaseTimer>(c.stream);
      uint32_t nb = 0;                                             // the listed terms of this call's range only
      const uint32_t* list = term_list_range(pk->b_index, pk->b_index_host, wbase, nterms_w, nb);
      build_plan(c, 2 * Ctx::kSlots + parity, w.p + wlo * 8, nb, plan_b, {{1, true}, {1, false}}, cw, !tab_w, list, (uint32_t)wbase);
      st.tplanb->stop();
      GS_HIP(hipEventRecord(st.planb, c.stream));
    }
    st.tplanw = std::make_shared<PhaseTimer>(c.stream);
    build_plan(c, 2 * parity, w.p + wlo * 8, (uint32_t)nterms_w, plan_w, st.split_b ? std::vector<LaunchShape>{{5, false}} : std::vector<LaunchShape>{{1, true}, {6, false}},
               cw, !tab_w);
    st.tplanw->stop();
    GS_HIP(hipEventRecord(st.planw, c.stream));
  }
  auto enqueue_acc_w = [&] {                                     // main: the accumulations over w, back to back
    StreamScope sc(c, c.main_stream);
    GS_HIP(hipStreamWaitEvent(c.stream, st.split_b ? st.planb : st.planw, 0));
    // A and Ap run over i > NPublic only (snark.go:265-268): their first npublic+1 points were forced
    // to infinity at key creation; Bp, C, Cp, Kp and B run over all variables (:270-278).
    msm_enqueue_g2(c, st.split_b ? plan_b : plan_w, {base_w(pk->t_b2, pk->b2)}, ws + 6, pin + 1, st.pend_g2w, ps.tail_g2);
    GS_HIP(hipEventRecord(st.done_g2, ps.tail_g2));
    if (st.split_b) {
      msm_enqueue_g1(c, plan_b, {base_w(pk->t_bp, pk->bp)}, ws + 5, 3 * Ctx::kSlots + parity, st.pend_g1b, ps.tail_g1);
      GS_HIP(hipEventRecord(st.done_g1b, ps.tail_g1));
      GS_HIP(hipStreamWaitEvent(c.stream, st.planw, 0));
      msm_enqueue_g1(c, plan_w, {base_w(pk->t_a, pk->a), base_w(pk->t_ap, pk->ap), base_w(pk->t_c, pk->c), base_w(pk->t_cp, pk->cp), base_w(pk->t_kp, pk->kp)},
                     ws + 0, pin + 0, st.pend_g1w, ps.tail_g1);
    } else {
      msm_enqueue_g1(c, plan_w, {base_w(pk->t_a, pk->a), base_w(pk->t_ap, pk->ap), base_w(pk->t_bp, pk->bp), base_w(pk->t_c, pk->c),
                                 base_w(pk->t_cp, pk->cp), base_w(pk->t_kp, pk->kp)}, ws + 0, pin + 0, st.pend_g1w, ps.tail_g1);
    }
    GS_HIP(hipEventRecord(st.done_g1w, ps.tail_g1));
  };
  auto enqueue_poly = [&] {                                      // aux 1 again: H(x), plan(h)
    StreamScope sc(c, ps.poly);
    if (px.host && px.n && px.host_stream) {   // host-buffer ticket: on the copy stream (groth16_enqueue)
      staged_h2d(c, const_cast<uint32_t*>(px.p), px.host, px.n * 32, px.host_stream);
      if (host_stage_mode() == 1) GS_HIP(hipStreamSynchronize(px.host_stream));
      else {
        GS_HIP(hipEventRecord(px.host_done, px.host_stream));
        GS_HIP(hipStreamWaitEvent(c.stream, px.host_done, 0));
      }
    } else if (px.host && px.n && pipelined) {      // host-buffer ticket, GS_HOST_STAGE=2 (groth16_enqueue)
      staged_h2d(c, const_cast<uint32_t*>(px.p), px.host, px.n * 32, c.stream);
    } else if (px.host && px.n) {     // the accumulations over w are already enqueued: this copy is off the critical path
      PhaseTimer th(c.stream);
      staged_h2d(c, const_cast<uint32_t*>(px.p), px.host, px.n * 32, c.stream);
      th.stop();
      c.timing.h2d_ms += th.ms();
    }
    st.tpoly = std::make_shared<PhaseTimer>(c.stream);
    bool have_hx = false;
    if (px.hv_slice) have_hx = true;                                            // H's values came from another rank
    else if (eval) {                                                            // H's values, for the evaluation-basis table
      px.produce_hv(c, hxbuf.as<uint32_t>(), c.bad_dev.as<uint32_t>() + parity);
      GS_HIP(hipMemcpyAsync(c.bad_host + parity, c.bad_dev.as<uint32_t>() + parity, 4, hipMemcpyDeviceToHost, c.stream));
      st.bad_host = c.bad_host + parity;
      have_hx = true;
    }
    if (!have_hx && px.produce_hx && nh) have_hx = px.produce_hx(c, hxbuf.as<uint32_t>());   // H from the constraint values (satisfying witness)
    if (!have_hx) {
      if (px.produce) px.produce(c);                                            // r1csqap.go:191-210 on the sparse system
      if (nh) poly_quotient_dev(c, pk->z, px.p, px.n, hxbuf.as<uint32_t>());     // snark.go:280
    }
    st.tpoly->stop();
    st.tplanh = std::make_shared<PhaseTimer>(c.stream);
    build_plan(c, 2 * parity + 1, px.hv_slice ? px.hv_slice : hxbuf.as<uint32_t>() + hlo * 8, (uint32_t)(hhi - hlo), plan_h, {{1, false}}, ch, !tab_h);
    st.tplanh->stop();
    GS_HIP(hipEventRecord(st.planh, c.stream));
  };
  if (ps.poly_first) { enqueue_poly(); enqueue_acc_w(); } else { enqueue_acc_w(); enqueue_poly(); }
  {                                                              // main again: the accumulation over h
    StreamScope sc(c, c.main_stream);
    GS_HIP(hipStreamWaitEvent(c.stream, st.planh, 0));
    msm_enqueue_g1(c, plan_h, {eval ? MsmBase{&pk->t_g1t_eval, hbase, pk->g1t_eval.as<uint32_t>(), pk->n_e} : MsmBase{&pk->t_g1t, hbase, pk->g1t.as<uint32_t>(), pk->n_h}},
                   ws + 7, pin + 2, st.pend_h, pipelined ? ps.tail_g1 : nullptr);   // :284-286
    GS_HIP(hipEventRecord(st.done_h, pipelined ? ps.tail_g1 : c.main_stream));
  }
  st.total->stop();
  GS_HIP(hipEventRecord(st.done_main, c.main_stream));
  return GS_OK;
}

int pinocchio_collect(Ctx& c, PinInFlight& st, uint64_t out[72], int inf[8]) {
  std::vector<G1Xyzz> g1w, g1h;
  std::vector<G2Xyzz> g2w;
  {                                          // the host-side pair sums, each group as soon as its own tail is through (groth16_collect)
    GS_HIP(hipEventSynchronize(st.done_g2));
    auto f2 = std::async(std::launch::async, [&] { msm_finish_g2(c, st.pend_g2w, g2w); });
    struct Join { std::future<void>& f; ~Join() { if (f.valid()) f.wait(); } } j2{f2};
    GS_HIP(hipEventSynchronize(st.done_g1w));
    auto f1 = std::async(std::launch::async, [&] {
      msm_finish_g1(c, st.pend_g1w, g1w);
      if (st.split_b) {                              // g1w = [A, A', C, C', K']: B' goes back to position 2
        std::vector<G1Xyzz> g1b;
        msm_finish_g1(c, st.pend_g1b, g1b);
        g1w.insert(g1w.begin() + 2, g1b[0]);
      }
    });
    Join j1{f1};
    GS_HIP(hipEventSynchronize(st.done_h));
    msm_finish_g1(c, st.pend_h, g1h);
    f2.get(); f1.get();
  }
  GS_HIP(hipEventSynchronize(st.done_main));
  msm_book_timing(c, st.pend_g1w); msm_book_timing(c, st.pend_g2w); msm_book_timing(c, st.pend_h);
  if (st.split_b) { msm_book_timing(c, st.pend_g1b); c.timing.plan_ms += st.tplanb->ms(); }
  c.timing.poly_ms += st.tpoly->ms();
  c.timing.plan_ms += st.tplanw->ms() + st.tplanh->ms();
  c.timing.total_ms += st.total->ms();
  if (st.in.th2d) c.timing.h2d_ms += st.in.th2d->ms();
  if (st.bad_host && *st.bad_host != 0) return kRetryExact;   // evaluation-basis route on a witness that violates a constraint
  // output order: PiA | PiAp | PiB | PiBp | PiC | PiCp | PiH | PiKp
  inf[0] = g1_to_affine_std(g1w[0], out) ? 1 : 0;
  inf[1] = g1_to_affine_std(g1w[1], out + 8) ? 1 : 0;
  inf[2] = g2_to_affine_std(g2w[0], out + 16) ? 1 : 0;
  inf[3] = g1_to_affine_std(g1w[2], out + 32) ? 1 : 0;
  inf[4] = g1_to_affine_std(g1w[3], out + 40) ? 1 : 0;
  inf[5] = g1_to_affine_std(g1w[4], out + 48) ? 1 : 0;
  inf[6] = g1_to_affine_std(g1h[0], out + 56) ? 1 : 0;
  inf[7] = g1_to_affine_std(g1w[5], out + 64) ? 1 : 0;
  return GS_OK;
}

int pinocchio_prove_impl(Ctx& c, PinocchioPkObj* pk, DevScalars w, DevScalars px, uint64_t out[72], int inf[8], const Shard& shard = Shard{}) {
  for (;;) {
    PinInFlight st;
    int rc = pinocchio_enqueue(c, pk, w, px, shard, c.blocking_slot(), true, false, st);
    if (rc != GS_OK) return rc;
    rc = pinocchio_collect(c, st, out, inf);
    if (rc != kRetryExact) return rc;
    px.produce_hv = nullptr;                   // violated constraint: once more through px and the floor quotient
    c.timing.fallbacks += 1;
  }
}

// zero the first `count` packed points (-> infinity)
void force_infinity(Ctx& c, DevBuf& pts, size_t count, size_t words) {
  if (count) GS_HIP(hipMemsetAsync(pts.p, 0, count * words * 4, c.stream));
}

// upload host scalars into a scratch buffer
const uint32_t* upload_tmp(Ctx& c, DevBuf& buf, const uint64_t* host, size_t n) {
  buf.ensure(std::max<size_t>(n, 1) * 32);
  if (n) {
    PhaseTimer th(c.stream);
    staged_h2d(c, buf.p, host, n * 32, c.stream);
    th.stop();
    c.timing.h2d_ms += th.ms();
  }
  return buf.as<uint32_t>();
}
void download(Ctx& c, uint64_t* host, const void* dev, size_t n) {
  if (n) GS_HIP(hipMemcpyAsync(host, dev, n * 32, hipMemcpyDeviceToHost, c.stream));
  GS_HIP(hipStreamSynchronize(c.stream));
}

// ---- host-buffer tickets -----------------------------------------------------------------------------------------------------
// The reference hands GenerateProofs a FRESH w (and px) in host memory on every call (groth16/groth16.go:225, cli/main.go:480-501).
// gs_scalars_upload + gs_*_begin + gs_free costs a hipMalloc, a blocking copy and a hipFree (a device-wide synchronisation) per proof
// and breaks a pipeline of three; a host-buffer ticket instead stages the caller's arrays into buffers its SLOT owns (grow-only:
// nothing is allocated or freed once the three slots have been used), on the copy stream -- the DMA of proof k + 3 runs beside the
// accumulations of proofs k + 1 and k + 2 -- and enqueues the proof when the copy has landed (GS_HOST_STAGE above: measured against
// a cross-stream event and against copying on the readers' own streams).  The caller's arrays are consumed when _begin returns (cgo
// pointer rule).
static void ensure_copy_stream(Ctx& c) {
  if (!c.copy_stream) GS_HIP(hipStreamCreateWithFlags(&c.copy_stream, hipStreamNonBlocking));
}
// px_follows: slot_px_from_host stages px on the same stream right after and waits for both (one DMA tail instead of two)
DevScalars stage_slot_w(Ctx& c, int parity, const uint64_t* host, size_t n, HostInputs& in, bool px_follows = false) {
  ensure_copy_stream(c);
  in.create();
  DevBuf& b = prove_state(c).slot_w[parity];
  b.ensure(std::max<size_t>(n, 1) * 32);
  DevScalars d{b.as<uint32_t>(), n};
  if (host_stage_mode() == 2) {          // copied by the enqueue function on the plan(w) stream
    d.host = host;
    d.host_done = in.w_done;
    return d;
  }
  in.th2d = std::make_shared<PhaseTimer>(c.copy_stream);
  if (n) staged_h2d(c, b.p, host, n * 32, c.copy_stream);
  if (host_stage_mode() == 1) { if (!px_follows) GS_HIP(hipStreamSynchronize(c.copy_stream)); }
  else {
    GS_HIP(hipEventRecord(in.w_done, c.copy_stream));
    d.host_done = in.w_done;
  }
  return d;
}
// px: staged by the enqueue function AFTER the accumulations over w were queued (they do not need it), on the copy stream
DevScalars slot_px_from_host(Ctx& c, int parity, const uint64_t* host, size_t n, HostInputs& in) {
  ensure_copy_stream(c);
  in.create();
  DevBuf& b = prove_state(c).slot_px[parity];
  b.ensure(std::max<size_t>(n, 1) * 32);
  if (host_stage_mode() == 1) {          // landed before anything of the proof is enqueued (the device is busy with the tickets before this one)
    if (n) staged_h2d(c, b.p, host, n * 32, c.copy_stream);
    GS_HIP(hipStreamSynchronize(c.copy_stream));
    return DevScalars{b.as<uint32_t>(), n};
  }
  DevScalars d{b.as<uint32_t>(), n, host};
  d.host_stream = host_stage_mode() == 2 ? nullptr : c.copy_stream;      // (nullptr: on the polynomial stream itself, as the blocking entry points do)
  d.host_done = in.px_done;
  return d;
}
// a resident vector that a ticket reads: gs_scalars_update must wait for these points (runtime.h, Scalars::reads)
void mark_ticket_reads(const ProofStreams& ps, Scalars* w, Scalars* px_or_hv) {
  if (w) { w->mark_read(ps.planw); if (ps.poly != ps.planw) w->mark_read(ps.poly); }
  if (px_or_hv) px_or_hv->mark_read(ps.poly);
}

}  // namespace

// the list of the variables with a finite point in either array: the device scans (one bit per variable), the host turns the bits
// into the ascending index list and uploads it
static size_t scan_finite_terms(Ctx& c, const uint32_t* g1_pts, const uint32_t* g2_pts, size_t n, DevBuf& index_dev, std::vector<uint32_t>& index_host) {
  index_host.clear();
  index_dev.release();
  if (n == 0) return 0;
  const size_t words = (n + 31) / 32;
  DevBuf mask(words * 4);
  const size_t finite = finite_mask_dev(c, g1_pts, g2_pts, (uint32_t)n, mask.as<uint32_t>());
  if (finite == n) return finite;                        // nothing missing: no list, no second plan
  std::vector<uint32_t> bits(words);
  GS_HIP(hipMemcpyAsync(bits.data(), mask.p, words * 4, hipMemcpyDeviceToHost, c.stream));
  GS_HIP(hipStreamSynchronize(c.stream));
  index_host.reserve(finite);
  for (size_t wd = 0; wd < words; ++wd)
    for (uint32_t m = bits[wd]; m; m &= m - 1) index_host.push_back((uint32_t)(wd * 32 + (size_t)__builtin_ctz(m)));
  index_dev.alloc(std::max<size_t>(index_host.size(), 1) * 4);
  if (!index_host.empty()) {
    GS_HIP(hipMemcpyAsync(index_dev.p, index_host.data(), index_host.size() * 4, hipMemcpyHostToDevice, c.stream));
    GS_HIP(hipStreamSynchronize(c.stream));
  }
  return finite;
}
void gs::groth_pk_scan_sparsity(Ctx& c, GrothPkObj& pk) {
  pk.b_finite = scan_finite_terms(c, pk.bacgamma1.as<uint32_t>(), pk.bacgamma2.as<uint32_t>(), pk.n_w, pk.b_index, pk.b_index_host);
}

void gs::pinocchio_pk_scan_sparsity(Ctx& c, PinocchioPkObj& pk) {
  pk.b_finite = scan_finite_terms(c, pk.bp.as<uint32_t>(), pk.b2.as<uint32_t>(), pk.n_w, pk.b_index, pk.b_index_host);
}

extern "C" {

// ---- polynomial field ------------------------------------------------------------------------------------
int gs_poly_mul(const uint64_t* a, size_t na, const uint64_t* b, size_t nb, uint64_t* out) {
  return guarded([&](Ctx& c) -> int {
    if (!a || !b || !out || na == 0 || nb == 0) return fail(GS_ERR_ARG, "gs_poly_mul: empty or null operand");
    if (na + nb > (1ull << 27)) return fail(GS_ERR_ARG, "gs_poly_mul: product too large");
    const uint32_t* da = upload_tmp(c, prove_state(c).up_a, a, na);
    const uint32_t* db = upload_tmp(c, prove_state(c).up_b, b, nb);
    const size_t nr = na + nb - 1;
    prove_state(c).up_o.ensure(nr * 32);
    poly_mul_dev(c, da, na, Form::Std, db, nb, Form::Std, prove_state(c).up_o.as<uint32_t>());
    poly_canon_dev(c, prove_state(c).up_o.as<uint32_t>(), nr, 0);
    download(c, out, prove_state(c).up_o.p, nr);
    return GS_OK;
  });
}

int gs_poly_div(const uint64_t* a, size_t na, const uint64_t* b, size_t nb, uint64_t* quo, uint64_t* rem) {
  return guarded([&](Ctx& c) -> int {
    if (!a || !b || !quo || nb == 0 || na < nb) return fail(GS_ERR_ARG, "gs_poly_div: need len(a) >= len(b) >= 1");
    bool lead_zero = true;
    for (int i = 0; i < 4; ++i) lead_zero = lead_zero && b[4 * (nb - 1) + i] == 0;
    if (lead_zero) return fail(GS_ERR_ARG, "gs_poly_div: leading coefficient of the divisor is zero");
    const uint32_t* da = upload_tmp(c, prove_state(c).up_a, a, na);
    const uint32_t* db = upload_tmp(c, prove_state(c).up_b, b, nb);
    Divisor d;
    divisor_init(c, d, db, nb);
    const size_t nq = na - nb + 1;
    prove_state(c).up_o.ensure((nq + na + nb) * 32);
    uint32_t* q = prove_state(c).up_o.as<uint32_t>();
    poly_quotient_dev(c, d, da, na, q);
    if (rem && nb > 1) {
      // rem = (a - q b) mod x^(nb-1)        (r1csqap.go:70-84 returns the final `rem`)
      uint32_t* qb = q + nq * 8;
      poly_mul_dev(c, q, nq, Form::Std, db, nb, Form::Std, qb);
      uint32_t* rr = qb + (nq + nb - 1) * 8;              // up_o holds nq + (nq + nb - 1) + (nb - 1) <= nq + na + nb elements: no allocation here
      poly_addsub_dev(c, da, nb - 1, qb, nb - 1, true, rr);
      poly_canon_dev(c, rr, nb - 1, 0);
      GS_HIP(hipMemcpyAsync(rem, rr, (nb - 1) * 32, hipMemcpyDeviceToHost, c.stream));
      GS_HIP(hipStreamSynchronize(c.stream));
    }
    poly_canon_dev(c, q, nq, 0);
    download(c, quo, q, nq);
    return GS_OK;
  });
}

static int addsub_api(const uint64_t* a, size_t na, const uint64_t* b, size_t nb, uint64_t* out, bool sub) {
  return guarded([&](Ctx& c) -> int {
    if ((na && !a) || (nb && !b) || !out) return fail(GS_ERR_ARG, "null operand");
    const size_t n = std::max(na, nb);
    if (n == 0) return GS_OK;
    const uint32_t* da = upload_tmp(c, prove_state(c).up_a, a, na);
    const uint32_t* db = upload_tmp(c, prove_state(c).up_b, b, nb);
    prove_state(c).up_o.ensure(n * 32);
    poly_addsub_dev(c, da, na, db, nb, sub, prove_state(c).up_o.as<uint32_t>());
    poly_canon_dev(c, prove_state(c).up_o.as<uint32_t>(), n, 0);
    download(c, out, prove_state(c).up_o.p, n);
    return GS_OK;
  });
}
int gs_poly_add(const uint64_t* a, size_t na, const uint64_t* b, size_t nb, uint64_t* out) { return addsub_api(a, na, b, nb, out, false); }
int gs_poly_sub(const uint64_t* a, size_t na, const uint64_t* b, size_t nb, uint64_t* out) { return addsub_api(a, na, b, nb, out, true); }

int gs_poly_eval(const uint64_t* v, size_t n, const uint64_t x[4], uint64_t out[4]) {
  return guarded([&](Ctx& c) -> int {
    if ((n && !v) || !x || !out) return fail(GS_ERR_ARG, "null operand");
    const uint32_t* dv = upload_tmp(c, prove_state(c).up_a, v, n);
    prove_state(c).up_o.ensure(32);
    poly_eval_dev(c, dv, n, x, prove_state(c).up_o.as<uint32_t>());
    download(c, out, prove_state(c).up_o.p, 1);
    return GS_OK;
  });
}

// ---- Groth16 ----------------------------------------------------------------------------------------------
// Shared builder of full keys and key slices: copies [lo, lo + n) of each source array (device, packed affine).
struct PkSrc { const DevBuf* buf; size_t lo; };
// `from`: the context the source arrays live on (another GPU for gs_groth16_pk_shard_to)
static void copy_slice(Ctx& c, Ctx& from, const PkSrc& s, size_t n, size_t words, DevBuf& dst) {
  dst.alloc(std::max<size_t>(n, 1) * words * 4);
  copy_between(c, dst.p, from, static_cast<const char*>(s.buf->p) + s.lo * words * 4, n * words * 4);
}
static void groth_pk_fill(Ctx& c, Ctx& from, GrothPkObj& pk, PkSrc at, PkSrc b1, PkSrc b2, PkSrc cd, PkSrc pt) {
  copy_slice(c, from, at, pk.n_w, kG1Aff, pk.at);
  copy_slice(c, from, b1, pk.n_w, kG1Aff, pk.bacgamma1);
  copy_slice(c, from, cd, pk.n_w, kG1Aff, pk.bacdelta);
  copy_slice(c, from, pt, pk.n_h, kG1Aff, pk.ptd);
  copy_slice(c, from, b2, pk.n_w, kG2Aff, pk.bacgamma2);
  // groth16.go:177-180 / :248: the C sum runs over i > NPublic; global entries [0, NPublic] of BACDelta become infinity
  const size_t zero_hi = std::min(pk.npublic + 1, pk.w_lo + pk.n_w);
  if (zero_hi > pk.w_lo) force_infinity(c, pk.bacdelta, zero_hi - pk.w_lo, kG1Aff);
}
static void set_shard(GrothPkObj& pk, size_t index, size_t count) {
  Shard sh; sh.index = index; sh.count = count;
  size_t lo, hi;
  pk.shard_index = index; pk.shard_count = count;
  shard_range(pk.nvars, sh, lo, hi);
  pk.w_lo = lo; pk.n_w = hi - lo;
  shard_range(pk.nptd, sh, lo, hi);
  pk.h_lo = lo; pk.n_h = hi - lo;
}

static int groth_pk_create_impl(Ctx& c, gs_handle g1_at, gs_handle g1_bacgamma, gs_handle g2_bacgamma, gs_handle bacdelta, gs_handle ptd,
                                const uint64_t g1_alpha[12], const uint64_t g1_beta[12], const uint64_t g1_delta[12],
                                const uint64_t g2_beta[24], const uint64_t g2_delta[24], const uint64_t* z, size_t nz,
                                size_t nvars, size_t npublic, size_t nptd_total, size_t shard_index, size_t shard_count, gs_handle* out) {
  Bases* at = c.get<Bases>(g1_at, Kind::G1Bases);
  Bases* b1 = c.get<Bases>(g1_bacgamma, Kind::G1Bases);
  Bases* b2 = c.get<Bases>(g2_bacgamma, Kind::G2Bases);
  Bases* cd = c.get<Bases>(bacdelta, Kind::G1Bases);
  Bases* pt = c.get<Bases>(ptd, Kind::G1Bases);
  if (!at || !b1 || !b2 || !cd || !pt) return fail(GS_ERR_ARG, "gs_groth16_pk_create: bad base handle");
  if (!g1_alpha || !g1_beta || !g1_delta || !g2_beta || !g2_delta || !z || !out || nz == 0) return fail(GS_ERR_ARG, "null argument");
  if (shard_count == 0 || shard_index >= shard_count) return fail(GS_ERR_ARG, "gs_groth16_pk_create_shard: bad shard %zu of %zu", shard_index, shard_count);
  if (npublic + 1 > nvars) return fail(GS_ERR_SHAPE, "NPublic + 1 > NVars");
  bool lead_zero = true;
  for (int i = 0; i < 4; ++i) lead_zero = lead_zero && z[4 * (nz - 1) + i] == 0;
  if (lead_zero) return fail(GS_ERR_ARG, "leading coefficient of Z is zero");
  auto pk = std::make_unique<GrothPkObj>();
  pk->nvars = nvars; pk->npublic = npublic; pk->nz = nz; pk->nptd = shard_count == 1 ? pt->n : nptd_total;
  set_shard(*pk, shard_index, shard_count);
  if (at->n != pk->n_w || b1->n != pk->n_w || b2->n != pk->n_w || cd->n != pk->n_w)
    return fail(GS_ERR_SHAPE, "At/BACGamma/BACDelta must have %zu points (NVars = %zu, shard %zu of %zu), got %zu/%zu/%zu/%zu", pk->n_w, nvars,
                shard_index, shard_count, at->n, b1->n, b2->n, cd->n);
  if (pt->n != pk->n_h)
    return fail(GS_ERR_SHAPE, "PowersTauDelta must have %zu points (total %zu, shard %zu of %zu), got %zu", pk->n_h, pk->nptd, shard_index, shard_count, pt->n);
  groth_pk_fill(c, c, *pk, PkSrc{&at->buf, 0}, PkSrc{&b1->buf, 0}, PkSrc{&b2->buf, 0}, PkSrc{&cd->buf, 0}, PkSrc{&pt->buf, 0});
  pk->alpha = g1_affine_from_jacobian_std(g1_alpha);
  pk->beta = g1_affine_from_jacobian_std(g1_beta);
  pk->delta = g1_affine_from_jacobian_std(g1_delta);
  pk->beta2 = g2_affine_from_jacobian_std(g2_beta);
  pk->delta2 = g2_affine_from_jacobian_std(g2_delta);
  const uint32_t* dz = upload_tmp(c, prove_state(c).up_a, z, nz);
  divisor_init(c, pk->z, dz, nz);
  GS_HIP(hipStreamSynchronize(c.stream));
  groth_pk_scan_sparsity(c, *pk);
  *out = c.put(std::move(pk));
  return GS_OK;
}

int gs_groth16_pk_create(gs_handle g1_at, gs_handle g1_bacgamma, gs_handle g2_bacgamma, gs_handle bacdelta, gs_handle ptd,
                         const uint64_t g1_alpha[12], const uint64_t g1_beta[12], const uint64_t g1_delta[12],
                         const uint64_t g2_beta[24], const uint64_t g2_delta[24], const uint64_t* z, size_t nz,
                         size_t nvars, size_t npublic, gs_handle* out) {
  return guarded([&](Ctx& c) -> int {
    return groth_pk_create_impl(c, g1_at, g1_bacgamma, g2_bacgamma, bacdelta, ptd, g1_alpha, g1_beta, g1_delta, g2_beta, g2_delta, z, nz,
                                nvars, npublic, 0, 0, 1, out);
  }, true, false, g1_at);
}

int gs_groth16_pk_create_shard(gs_handle g1_at, gs_handle g1_bacgamma, gs_handle g2_bacgamma, gs_handle bacdelta, gs_handle ptd,
                               const uint64_t g1_alpha[12], const uint64_t g1_beta[12], const uint64_t g1_delta[12],
                               const uint64_t g2_beta[24], const uint64_t g2_delta[24], const uint64_t* z, size_t nz,
                               size_t nvars, size_t npublic, size_t nptd_total, size_t shard_index, size_t shard_count, gs_handle* out) {
  return guarded([&](Ctx& c) -> int {
    return groth_pk_create_impl(c, g1_at, g1_bacgamma, g2_bacgamma, bacdelta, ptd, g1_alpha, g1_beta, g1_delta, g2_beta, g2_delta, z, nz,
                                nvars, npublic, nptd_total, shard_index, shard_count, out);
  }, true, false, g1_at);
}

// A slice of a resident full key (device-to-device copies): what each rank keeps when the full key was built or loaded
// locally; the caller then frees the full key.  `c` is the context the slice is created on -- the key's own, or another
// logical device's (gs_groth16_pk_shard_to: the copies then cross xGMI, or stay on the GPU when both share one).
static int groth_pk_shard_impl(Ctx& c, Ctx& from, GrothPkObj* full, size_t shard_index, size_t shard_count, gs_handle* out) {
  if (!full || !out) return fail(GS_ERR_ARG, "gs_groth16_pk_shard: bad proving-key handle or null output");
  if (full->shard_count != 1) return fail(GS_ERR_ARG, "gs_groth16_pk_shard: the source key is itself a slice");
  if (shard_count == 0 || shard_index >= shard_count) return fail(GS_ERR_ARG, "gs_groth16_pk_shard: bad shard %zu of %zu", shard_index, shard_count);
  auto pk = std::make_unique<GrothPkObj>();
  pk->nvars = full->nvars; pk->npublic = full->npublic; pk->nz = full->nz; pk->nptd = full->nptd;
  set_shard(*pk, shard_index, shard_count);
  groth_pk_fill(c, from, *pk, PkSrc{&full->at, pk->w_lo}, PkSrc{&full->bacgamma1, pk->w_lo}, PkSrc{&full->bacgamma2, pk->w_lo},
                PkSrc{&full->bacdelta, pk->w_lo}, PkSrc{&full->ptd, pk->h_lo});
  if (full->n_eval) {                                   // the evaluation-basis array is cut like the other term ranges (its own split of [0, n))
    Shard sh; sh.index = shard_index; sh.count = shard_count;
    size_t lo, hi;
    shard_range(full->n_eval, sh, lo, hi);
    pk->n_eval = full->n_eval; pk->e_lo = lo; pk->n_e = hi - lo;
    copy_slice(c, from, PkSrc{&full->ptd_eval, lo}, pk->n_e, kG1Aff, pk->ptd_eval);
  }
  pk->alpha = full->alpha; pk->beta = full->beta; pk->delta = full->delta; pk->beta2 = full->beta2; pk->delta2 = full->delta2;
  {                                                     // Z travels with every slice
    DevBuf zc(std::max<size_t>(full->nz, 1) * 32);
    copy_between(c, zc.p, from, full->z.b_std.p, full->nz * 32);
    divisor_init(c, pk->z, zc.as<uint32_t>(), full->nz);
    GS_HIP(hipStreamSynchronize(c.stream));             // `zc` is released here
  }
  GS_HIP(hipStreamSynchronize(c.stream));
  groth_pk_scan_sparsity(c, *pk);
  *out = c.put(std::move(pk));
  return GS_OK;
}

int gs_groth16_pk_shard(gs_handle hfull, size_t shard_index, size_t shard_count, gs_handle* out) {
  return guarded([&](Ctx& c) -> int {
    return groth_pk_shard_impl(c, c, c.get<GrothPkObj>(hfull, Kind::GrothPk), shard_index, shard_count, out);
  }, true, false, hfull);
}

int gs_groth16_pk_shard_to(gs_handle hfull, size_t shard_index, size_t shard_count, int target_device, gs_handle* out) {
  return guarded_pair(hfull, target_device, [&](Ctx& src, Ctx& dst) -> int {
    return groth_pk_shard_impl(dst, src, src.get<GrothPkObj>(hfull, Kind::GrothPk), shard_index, shard_count, out);
  });
}

int gs_groth16_prove(gs_handle hpk, const uint64_t* w, size_t nw, const uint64_t* px, size_t npx,
                     const uint64_t r[4], const uint64_t s[4], uint64_t out_proof[32], int inf[3]) {
  return guarded([&](Ctx& c) -> int {
    GrothPkObj* pk = c.get<GrothPkObj>(hpk, Kind::GrothPk);
    if (!pk) return fail(GS_ERR_ARG, "gs_groth16_prove: bad proving-key handle");
    if (!w || !px || !r || !s || !out_proof || !inf) return fail(GS_ERR_ARG, "null argument");
    reset_timing(c);
    DevScalars dw{upload_tmp(c, prove_state(c).up_w, w, nw), nw};
    prove_state(c).up_px.ensure(std::max<size_t>(npx, 1) * 32);
    DevScalars dp{prove_state(c).up_px.as<uint32_t>(), npx, px};               // copied inside, behind the work that only needs w
    return groth16_prove_impl(c, pk, dw, dp, r, s, out_proof, inf);
  }, true, false, hpk);
}

int gs_groth16_prove_resident(gs_handle hpk, gs_handle hw, gs_handle hpx, const uint64_t r[4], const uint64_t s[4],
                              uint64_t out_proof[32], int inf[3]) {
  return guarded([&](Ctx& c) -> int {
    GrothPkObj* pk = c.get<GrothPkObj>(hpk, Kind::GrothPk);
    Scalars* w = c.get<Scalars>(hw, Kind::Scalars);
    Scalars* px = c.get<Scalars>(hpx, Kind::Scalars);
    if (!pk || !w || !px) return fail(GS_ERR_ARG, "gs_groth16_prove_resident: bad handle");
    if (!r || !s || !out_proof || !inf) return fail(GS_ERR_ARG, "null argument");
    reset_timing(c);
    return groth16_prove_impl(c, pk, DevScalars{w->buf.as<uint32_t>(), w->n}, DevScalars{px->buf.as<uint32_t>(), px->n}, r, s, out_proof, inf);
  }, true, false, hpk);
}

// Pipelined proving: begin enqueues a whole proof and returns; end waits for THAT proof only and runs its tail.  With two
// proofs outstanding the device never idles between proofs (the next plan/accumulations are already queued) and the host
// tail of proof k overlaps the device work of proof k+1.
// `w_host` / `px_host` non-null: the host-buffer form (gs_groth16_prove_host_begin), hw / hpx are ignored then.
static int groth16_begin_impl(Ctx& c, const char* fn, gs_handle hpk, gs_handle hw, gs_handle hpx, const uint64_t* w_host, size_t nw,
                              const uint64_t* px_host, size_t npx, const uint64_t r[4], const uint64_t s[4], uint64_t* ticket) {
  const bool host = w_host != nullptr || px_host != nullptr;
  GrothPkObj* pk = c.get<GrothPkObj>(hpk, Kind::GrothPk);
  Scalars* w = host ? nullptr : c.get<Scalars>(hw, Kind::Scalars);
  Scalars* px = host ? nullptr : c.get<Scalars>(hpx, Kind::Scalars);
  if (!pk || (!host && (!w || !px))) return fail(GS_ERR_ARG, "%s: bad handle", fn);
  if (!r || !s || !ticket || (host && ((nw && !w_host) || (npx && !px_host)))) return fail(GS_ERR_ARG, "null argument");
  if (host && (nw >= (1ull << 31) || npx >= (1ull << 31))) return fail(GS_ERR_ARG, "%s: too many scalars", fn);
  if (host && nw != pk->nvars) return fail(GS_ERR_SHAPE, "len(w) = %zu but the key has %zu variables", nw, pk->nvars);   // before anything is staged
  const int parity = c.free_parity();
  if (parity < 0) return fail(GS_ERR_BUSY, "%s: three operations are already outstanding; call gs_groth16_prove_end first", fn);
  auto st = std::make_unique<GrothInFlight>();
  memcpy(st->r, r, 32); memcpy(st->s, s, 32);
  st->with_tail = true;
  GrothInFlight* raw = st.get();
  raw->pk = pk;
  raw->keep = {c.share<Object>(hpk, Kind::GrothPk)};
  if (!host) { raw->keep.push_back(c.share<Object>(hw, Kind::Scalars)); raw->keep.push_back(c.share<Object>(hpx, Kind::Scalars)); }
  raw->fpre = std::async(std::launch::async, [raw] { groth16_tail_pre(raw->pk, raw->r, raw->s, raw->pre); });
  raw->early.pk = pk; raw->early.r = raw->r; raw->early.s = raw->s; raw->early.pre = &raw->pre; raw->early.fpre = &raw->fpre;
  const double h0 = host_trace() ? host_now_ms() : 0;
  const DevScalars dw = host ? stage_slot_w(c, parity, w_host, nw, raw->in, true) : DevScalars{w->buf.as<uint32_t>(), w->n};
  const double h1 = host_trace() ? host_now_ms() : 0;
  const DevScalars dp = host ? slot_px_from_host(c, parity, px_host, npx, raw->in) : DevScalars{px->buf.as<uint32_t>(), px->n};
  const double h2 = host_trace() ? host_now_ms() : 0;
  const int rc = groth16_enqueue(c, pk, dw, dp, Shard{}, parity, false, true, *raw);
  if (rc != GS_OK) return rc;
  if (host_trace()) fprintf(stderr, "[gs host] begin: stage w %.3f ms, stage px %.3f ms, enqueue %.3f ms\n", h1 - h0, h2 - h1, host_now_ms() - h2);
  if (raw->in.th2d) raw->in.th2d->stop();
  mark_ticket_reads(raw->streams, w, px);
  st->ticket = c.new_ticket();
  *ticket = st->ticket;
  c.inflight[parity] = std::move(st);
  return GS_OK;
}

int gs_groth16_prove_begin(gs_handle hpk, gs_handle hw, gs_handle hpx, const uint64_t r[4], const uint64_t s[4], uint64_t* ticket) {
  return guarded([&](Ctx& c) -> int {
    return groth16_begin_impl(c, "gs_groth16_prove_begin", hpk, hw, hpx, nullptr, 0, nullptr, 0, r, s, ticket);
  }, true, true, hpk);
}

// groth16.GenerateProofs' own call shape, pipelined: w and px in caller memory, new ones with every call (host-buffer ticket, above).
int gs_groth16_prove_host_begin(gs_handle hpk, const uint64_t* w, size_t nw, const uint64_t* px, size_t npx, const uint64_t r[4], const uint64_t s[4],
                                uint64_t* ticket) {
  return guarded([&](Ctx& c) -> int {
    if (!w || !px) return fail(GS_ERR_ARG, "gs_groth16_prove_host_begin: null w or px");
    return groth16_begin_impl(c, "gs_groth16_prove_host_begin", hpk, 0, 0, w, nw, px, npx, r, s, ticket);
  }, true, true, hpk);
}

int gs_groth16_prove_end(uint64_t ticket, uint64_t out_proof[32], int inf[3]) {
  wait_ticket_unlocked(ticket);          // the device wait, outside the context lock (runtime.h)
  return guarded([&](Ctx& c) -> int {
    if (!out_proof || !inf) return fail(GS_ERR_ARG, "null argument");
    int parity = -1;
    for (int p = 0; p < Ctx::kMaxInFlight; ++p) if (c.inflight[p] && c.inflight[p]->ticket == ticket) parity = p;
    if (parity < 0) return fail(GS_ERR_ARG, "gs_groth16_prove_end: unknown ticket %llu", (unsigned long long)ticket);
    if (!dynamic_cast<GrothInFlight*>(c.inflight[parity].get()))
      return fail(GS_ERR_ARG, "gs_groth16_prove_end: ticket %llu belongs to an MSM (use gs_msm_end)", (unsigned long long)ticket);
    if (!static_cast<GrothInFlight*>(c.inflight[parity].get())->with_tail)
      return fail(GS_ERR_ARG, "gs_groth16_prove_end: ticket %llu is a partial-sums operation (use gs_groth16_partials_end)", (unsigned long long)ticket);
    std::shared_ptr<InFlightBase> base = std::move(c.inflight[parity]);
    GrothInFlight& st = static_cast<GrothInFlight&>(*base);
    reset_timing(c);
    GrothSums sums;
    int rc = groth16_collect(c, st, sums);
    if (rc == kRetryExact && st.exact_route) { c.timing.fallbacks += 1; rc = st.exact_route(c, sums); }
    if (rc != GS_OK) return rc;
    st.fpre.get();
    groth16_tail_post(st.pk, sums, st.pre, st.r, st.s, out_proof, inf, &st.early);
    return GS_OK;
  }, true, true, ticket);
}

// Sharded proving: the five raw sums over this rank's term ranges, as affine points.
int gs_groth16_prove_partials(gs_handle hpk, gs_handle hw, gs_handle hpx, size_t shard_index, size_t shard_count,
                              uint64_t out_sums[48], int inf[5]) {
  return guarded([&](Ctx& c) -> int {
    GrothPkObj* pk = c.get<GrothPkObj>(hpk, Kind::GrothPk);
    Scalars* w = c.get<Scalars>(hw, Kind::Scalars);
    Scalars* px = c.get<Scalars>(hpx, Kind::Scalars);
    if (!pk || !w || !px) return fail(GS_ERR_ARG, "gs_groth16_prove_partials: bad handle");
    if (!out_sums || !inf || shard_count == 0 || shard_index >= shard_count) return fail(GS_ERR_ARG, "gs_groth16_prove_partials: bad shard or null output");
    reset_timing(c);
    GrothSums sums;
    Shard sh; sh.index = shard_index; sh.count = shard_count;
    const int rc = groth16_sums_impl(c, pk, DevScalars{w->buf.as<uint32_t>(), w->n}, DevScalars{px->buf.as<uint32_t>(), px->n}, sh, sums);
    if (rc != GS_OK) return rc;
    inf[0] = g1_to_affine_std(sums.at, out_sums) ? 1 : 0;
    inf[1] = g1_to_affine_std(sums.bacgamma1, out_sums + 8) ? 1 : 0;
    inf[2] = g2_to_affine_std(sums.bacgamma2, out_sums + 16) ? 1 : 0;
    inf[3] = g1_to_affine_std(sums.bacdelta, out_sums + 32) ? 1 : 0;
    inf[4] = g1_to_affine_std(sums.h, out_sums + 40) ? 1 : 0;
    return GS_OK;
  }, true, false, hpk);
}

static void r1cs_values_dev(Ctx& c, R1csObj& o, const uint32_t* w_dev);
static bool hx_shape(size_t n, size_t nz);
// Strong scaling without a replicated polynomial stage ("run on GPU 0 and broadcast hx shards"), in two
// entry points.  gs_groth16_witness_values: ONE rank (the proof's owner; with a stream of proofs the ranks take turns) turns the
// resident witness into the n values H(n+1..2n) -- the whole polynomial stage of the evaluation-basis route -- as a resident scalar
// vector (*hv_inout as gs_r1cs_px's px_inout); *violated = number of roots of Z at which the witness breaks a constraint (the values are
// then meaningless: take gs_r1cs_px + gs_groth16_prove_partials).  The owner scatters slice k of the vector to rank k
// (gs_scalars_clone between the logical devices of one process, gs_scalars_scatter over RCCL between processes), and ...
static int witness_values_impl(Ctx& c, const char* fn, size_t nz, size_t n_eval, gs_handle hr1cs, gs_handle hw, gs_handle* hv_inout, uint32_t* violated) {
  R1csObj* o = c.get<R1csObj>(hr1cs, Kind::R1cs);
  Scalars* w = c.get<Scalars>(hw, Kind::Scalars);
  if (!o || !w || !hv_inout || !violated) return fail(GS_ERR_ARG, "%s: bad handle or null output", fn);
  if (w->n != o->m) return fail(GS_ERR_SHAPE, "len(w) = %zu but the system has %zu variables", w->n, o->m);
  if (!hx_shape(o->n, nz) || n_eval != o->n)
    return fail(GS_ERR_SHAPE, "%s: the key has no evaluation-basis array for a system of %zu constraints", fn, o->n);
  Scalars* hv = nullptr;
  if (*hv_inout) {
    hv = c.get<Scalars>(*hv_inout, Kind::Scalars);
    if (!hv || hv->n != o->n) return fail(GS_ERR_ARG, "%s: the output handle does not hold n = %zu values", fn, o->n);
  } else {
    auto fresh = std::make_unique<Scalars>();
    fresh->n = o->n;
    fresh->buf.alloc(o->n * 32);
    hv = fresh.get();
    *hv_inout = c.put(std::move(fresh));
  }
  StreamScope sc(c, c.aux_stream[1]);             // the stream that carries every proof's polynomial stage (gs_r1cs_px)
  PhaseTimer t(c.stream);
  uint32_t* bad = c.bad_dev.as<uint32_t>() + c.blocking_slot();
  r1cs_values_dev(c, *o, w->buf.as<uint32_t>());
  r1cs_check_dev(c, o->vals.as<uint32_t>(), o->n, nz - 1, bad);
  hx_values_dev(c, o->vals.as<uint32_t>(), o->n, nz - 1, hv->buf.as<uint32_t>());
  GS_HIP(hipMemcpyAsync(c.bad_host + c.blocking_slot(), bad, 4, hipMemcpyDeviceToHost, c.stream));
  t.stop();
  GS_HIP(hipStreamSynchronize(c.stream));
  *violated = c.bad_host[c.blocking_slot()];
  if (!c.any_inflight()) reset_timing(c);
  c.timing.poly_ms = t.ms();
  c.timing.total_ms = c.timing.poly_ms;
  return GS_OK;
}

int gs_groth16_witness_values(gs_handle hpk, gs_handle hr1cs, gs_handle hw, gs_handle* hv_inout, uint32_t* violated) {
  return guarded([&](Ctx& c) -> int {
    GrothPkObj* pk = c.get<GrothPkObj>(hpk, Kind::GrothPk);
    if (!pk) return fail(GS_ERR_ARG, "gs_groth16_witness_values: bad proving-key handle");
    return witness_values_impl(c, "gs_groth16_witness_values", pk->nz, pk->n_eval, hr1cs, hw, hv_inout, violated);
  }, true, true, hpk);
}

// ... every rank sums its term ranges: the four sums over its slice of w as gs_groth16_prove_partials does, the fifth over ITS slice
// of H's values (`hv_slice`: the n_e values of the key slice's evaluation-basis range) -- no polynomial work at all on this rank.
int gs_groth16_prove_partials_values(gs_handle hpk, gs_handle hw, gs_handle hv_slice, size_t shard_index, size_t shard_count,
                                     uint64_t out_sums[48], int inf[5]) {
  return guarded([&](Ctx& c) -> int {
    GrothPkObj* pk = c.get<GrothPkObj>(hpk, Kind::GrothPk);
    Scalars* w = c.get<Scalars>(hw, Kind::Scalars);
    Scalars* hv = c.get<Scalars>(hv_slice, Kind::Scalars);
    if (!pk || !w || !hv) return fail(GS_ERR_ARG, "gs_groth16_prove_partials_values: bad handle");
    if (!out_sums || !inf || shard_count == 0 || shard_index >= shard_count) return fail(GS_ERR_ARG, "gs_groth16_prove_partials_values: bad shard or null output");
    if (pk->n_eval == 0) return fail(GS_ERR_SHAPE, "gs_groth16_prove_partials_values: the key has no evaluation-basis array");
    Shard sh; sh.index = shard_index; sh.count = shard_count;
    size_t lo, hi;
    if (pk->shard_count > 1) { lo = pk->e_lo; hi = pk->e_lo + pk->n_e; } else shard_range(pk->n_eval, sh, lo, hi);
    if (hv->n != hi - lo) return fail(GS_ERR_SHAPE, "gs_groth16_prove_partials_values: shard %zu of %zu covers %zu of the %zu values, the vector holds %zu",
                                      shard_index, shard_count, hi - lo, pk->n_eval, hv->n);
    reset_timing(c);
    GrothSums sums;
    DevScalars dh{nullptr, 0};
    dh.hv_slice = hv->buf.as<uint32_t>();
    const int rc = groth16_sums_impl(c, pk, DevScalars{w->buf.as<uint32_t>(), w->n}, dh, sh, sums);
    if (rc != GS_OK) return rc;
    inf[0] = g1_to_affine_std(sums.at, out_sums) ? 1 : 0;
    inf[1] = g1_to_affine_std(sums.bacgamma1, out_sums + 8) ? 1 : 0;
    inf[2] = g2_to_affine_std(sums.bacgamma2, out_sums + 16) ? 1 : 0;
    inf[3] = g1_to_affine_std(sums.bacdelta, out_sums + 32) ? 1 : 0;
    inf[4] = g1_to_affine_std(sums.h, out_sums + 40) ? 1 : 0;
    return GS_OK;
  }, true, false, hpk);
}

// The same, pipelined (a rank streams its shards of consecutive proofs: the plan and accumulations of shard work k + 1 queue behind
// those of k): gs_groth16_partials_values_begin enqueues and returns a ticket, gs_groth16_partials_end collects the five sums.
int gs_groth16_partials_values_begin(gs_handle hpk, gs_handle hw, gs_handle hv_slice, size_t shard_index, size_t shard_count, uint64_t* ticket) {
  return guarded([&](Ctx& c) -> int {
    GrothPkObj* pk = c.get<GrothPkObj>(hpk, Kind::GrothPk);
    Scalars* w = c.get<Scalars>(hw, Kind::Scalars);
    Scalars* hv = c.get<Scalars>(hv_slice, Kind::Scalars);
    if (!pk || !w || !hv || !ticket) return fail(GS_ERR_ARG, "gs_groth16_partials_values_begin: bad handle or null ticket");
    if (shard_count == 0 || shard_index >= shard_count) return fail(GS_ERR_ARG, "gs_groth16_partials_values_begin: bad shard");
    if (pk->n_eval == 0) return fail(GS_ERR_SHAPE, "gs_groth16_partials_values_begin: the key has no evaluation-basis array");
    Shard sh; sh.index = shard_index; sh.count = shard_count;
    size_t lo, hi;
    if (pk->shard_count > 1) { lo = pk->e_lo; hi = pk->e_lo + pk->n_e; } else shard_range(pk->n_eval, sh, lo, hi);
    if (hv->n != hi - lo) return fail(GS_ERR_SHAPE, "gs_groth16_partials_values_begin: the shard covers %zu values, the vector holds %zu", hi - lo, hv->n);
    const int parity = c.free_parity();
    if (parity < 0) return fail(GS_ERR_BUSY, "gs_groth16_partials_values_begin: three operations are already outstanding; collect one first");
    auto st = std::make_unique<GrothInFlight>();
    st->pk = pk;
    st->keep = {c.share<Object>(hpk, Kind::GrothPk), c.share<Object>(hw, Kind::Scalars), c.share<Object>(hv_slice, Kind::Scalars)};
    DevScalars dh{nullptr, 0};
    dh.hv_slice = hv->buf.as<uint32_t>();
    const int rc = groth16_enqueue(c, pk, DevScalars{w->buf.as<uint32_t>(), w->n}, dh, sh, parity, false, true, *st);
    if (rc != GS_OK) return rc;
    mark_ticket_reads(st->streams, w, hv);
    st->ticket = c.new_ticket();
    *ticket = st->ticket;
    c.inflight[parity] = std::move(st);
    return GS_OK;
  }, true, true, hpk);
}

int gs_groth16_partials_end(uint64_t ticket, uint64_t out_sums[48], int inf[5]) {
  wait_ticket_unlocked(ticket);          // the device wait, outside the context lock (runtime.h)
  return guarded([&](Ctx& c) -> int {
    if (!out_sums || !inf) return fail(GS_ERR_ARG, "null argument");
    int parity = -1;
    for (int p = 0; p < Ctx::kMaxInFlight; ++p) if (c.inflight[p] && c.inflight[p]->ticket == ticket) parity = p;
    if (parity < 0) return fail(GS_ERR_ARG, "gs_groth16_partials_end: unknown ticket %llu", (unsigned long long)ticket);
    GrothInFlight* g = dynamic_cast<GrothInFlight*>(c.inflight[parity].get());
    if (!g || g->with_tail) return fail(GS_ERR_ARG, "gs_groth16_partials_end: ticket %llu is not a partial-sums operation", (unsigned long long)ticket);
    std::shared_ptr<InFlightBase> base = std::move(c.inflight[parity]);
    reset_timing(c);
    GrothSums sums;
    const int rc = groth16_collect(c, *g, sums);
    if (rc != GS_OK) return rc;
    inf[0] = g1_to_affine_std(sums.at, out_sums) ? 1 : 0;
    inf[1] = g1_to_affine_std(sums.bacgamma1, out_sums + 8) ? 1 : 0;
    inf[2] = g2_to_affine_std(sums.bacgamma2, out_sums + 16) ? 1 : 0;
    inf[3] = g1_to_affine_std(sums.bacdelta, out_sums + 32) ? 1 : 0;
    inf[4] = g1_to_affine_std(sums.h, out_sums + 40) ? 1 : 0;
    return GS_OK;
  }, true, true, ticket);
}

// ... and the O(1) tail of groth16.go:253-275 on the combined sums (same layout as gs_groth16_prove_partials emits).
int gs_groth16_finish(gs_handle hpk, const uint64_t sums_in[48], const int inf_in[5], const uint64_t r[4], const uint64_t s[4],
                      uint64_t out_proof[32], int inf[3]) {
  return guarded([&](Ctx& c) -> int {
    GrothPkObj* pk = c.get<GrothPkObj>(hpk, Kind::GrothPk);
    if (!pk) return fail(GS_ERR_ARG, "gs_groth16_finish: bad proving-key handle");
    if (!sums_in || !inf_in || !r || !s || !out_proof || !inf) return fail(GS_ERR_ARG, "null argument");
    auto g1 = [&](const uint64_t* p, int isinf) -> G1Xyzz {
      if (isinf) return xyzz_inf<FqTag>();
      const uint32_t* w32 = reinterpret_cast<const uint32_t*>(p);
      G1Affine a;
      a.x = canon(PointIO<FqTag>::load_std(w32)); a.y = canon(PointIO<FqTag>::load_std(w32 + 8));
      return xyzz_from_affine(a);
    };
    GrothSums sums;
    sums.at = g1(sums_in, inf_in[0]);
    sums.bacgamma1 = g1(sums_in + 8, inf_in[1]);
    if (inf_in[2]) sums.bacgamma2 = xyzz_inf<Fq2Tag>();
    else {
      const uint32_t* w32 = reinterpret_cast<const uint32_t*>(sums_in + 16);
      G2Affine a;
      a.x = canon(PointIO<Fq2Tag>::load_std(w32)); a.y = canon(PointIO<Fq2Tag>::load_std(w32 + 16));
      sums.bacgamma2 = xyzz_from_affine(a);
    }
    sums.bacdelta = g1(sums_in + 32, inf_in[3]);
    sums.h = g1(sums_in + 40, inf_in[4]);
    groth16_tail(pk, sums, r, s, out_proof, inf);
    return GS_OK;
  }, true, false, hpk);
}

// ---- Pinocchio ----------------------------------------------------------------------------------------------
int gs_pinocchio_pk_create(gs_handle a, gs_handle ap, gs_handle b_g2, gs_handle bp, gs_handle cc, gs_handle cp, gs_handle kp,
                           gs_handle g1t, const uint64_t* z, size_t nz, size_t nvars, size_t npublic, gs_handle* out) {
  return guarded([&](Ctx& c) -> int {
    Bases* A = c.get<Bases>(a, Kind::G1Bases);
    Bases* Ap = c.get<Bases>(ap, Kind::G1Bases);
    Bases* B = c.get<Bases>(b_g2, Kind::G2Bases);
    Bases* Bp = c.get<Bases>(bp, Kind::G1Bases);
    Bases* C = c.get<Bases>(cc, Kind::G1Bases);
    Bases* Cp = c.get<Bases>(cp, Kind::G1Bases);
    Bases* Kp = c.get<Bases>(kp, Kind::G1Bases);
    Bases* T = c.get<Bases>(g1t, Kind::G1Bases);
    if (!A || !Ap || !B || !Bp || !C || !Cp || !Kp || !T) return fail(GS_ERR_ARG, "gs_pinocchio_pk_create: bad base handle");
    if (!z || !out || nz == 0) return fail(GS_ERR_ARG, "null argument");
    for (Bases* x : {A, Ap, B, Bp, C, Cp, Kp})
      if (x->n != nvars) return fail(GS_ERR_SHAPE, "every per-variable key array must have NVars = %zu points (got %zu)", nvars, x->n);
    if (npublic + 1 > nvars) return fail(GS_ERR_SHAPE, "NPublic + 1 > NVars");
    auto pk = std::make_unique<PinocchioPkObj>();
    pk->nvars = nvars; pk->npublic = npublic; pk->nz = nz; pk->ng1t = T->n;
    pk->n_w = nvars; pk->n_h = T->n;                                // a full key
    copy_points(c, A, kG1Aff, pk->a);
    copy_points(c, Ap, kG1Aff, pk->ap);
    copy_points(c, Bp, kG1Aff, pk->bp);
    copy_points(c, C, kG1Aff, pk->c);
    copy_points(c, Cp, kG1Aff, pk->cp);
    copy_points(c, Kp, kG1Aff, pk->kp);
    copy_points(c, T, kG1Aff, pk->g1t);
    copy_points(c, B, kG2Aff, pk->b2);
    force_infinity(c, pk->a, npublic + 1, kG1Aff);                 // snark.go:265
    force_infinity(c, pk->ap, npublic + 1, kG1Aff);
    const uint32_t* dz = upload_tmp(c, prove_state(c).up_a, z, nz);
    divisor_init(c, pk->z, dz, nz);
    GS_HIP(hipStreamSynchronize(c.stream));
    pinocchio_pk_scan_sparsity(c, *pk);
    *out = c.put(std::move(pk));
    return GS_OK;
  }, true, false, a);
}

int gs_pinocchio_prove(gs_handle hpk, const uint64_t* w, size_t nw, const uint64_t* px, size_t npx, uint64_t out_proof[72], int inf[8]) {
  return guarded([&](Ctx& c) -> int {
    PinocchioPkObj* pk = c.get<PinocchioPkObj>(hpk, Kind::PinocchioPk);
    if (!pk) return fail(GS_ERR_ARG, "gs_pinocchio_prove: bad proving-key handle");
    if (!w || !px || !out_proof || !inf) return fail(GS_ERR_ARG, "null argument");
    reset_timing(c);
    DevScalars dw{upload_tmp(c, prove_state(c).up_w, w, nw), nw};
    prove_state(c).up_px.ensure(std::max<size_t>(npx, 1) * 32);
    DevScalars dp{prove_state(c).up_px.as<uint32_t>(), npx, px};               // copied inside, behind the work that only needs w
    return pinocchio_prove_impl(c, pk, dw, dp, out_proof, inf);
  }, true, false, hpk);
}

// Pipelined Pinocchio proving: same ticket discipline as gs_groth16_prove_begin / _end (they share the three slots).
static int pinocchio_begin_impl(Ctx& c, const char* fn, gs_handle hpk, gs_handle hw, gs_handle hpx, const uint64_t* w_host, size_t nw,
                                const uint64_t* px_host, size_t npx, uint64_t* ticket) {
  const bool host = w_host != nullptr || px_host != nullptr;
  PinocchioPkObj* pk = c.get<PinocchioPkObj>(hpk, Kind::PinocchioPk);
  Scalars* w = host ? nullptr : c.get<Scalars>(hw, Kind::Scalars);
  Scalars* px = host ? nullptr : c.get<Scalars>(hpx, Kind::Scalars);
  if (!pk || (!host && (!w || !px)) || !ticket) return fail(GS_ERR_ARG, "%s: bad handle or null ticket", fn);
  if (host && ((nw && !w_host) || (npx && !px_host))) return fail(GS_ERR_ARG, "null argument");
  if (host && (nw >= (1ull << 31) || npx >= (1ull << 31))) return fail(GS_ERR_ARG, "%s: too many scalars", fn);
  if (host && nw != pk->nvars) return fail(GS_ERR_SHAPE, "len(w) = %zu but the key has %zu variables", nw, pk->nvars);
  const int parity = c.free_parity();
  if (parity < 0) return fail(GS_ERR_BUSY, "%s: three operations are already outstanding; call gs_pinocchio_prove_end first", fn);
  auto st = std::make_unique<PinInFlight>();
  st->keep = {c.share<Object>(hpk, Kind::PinocchioPk)};
  if (!host) { st->keep.push_back(c.share<Object>(hw, Kind::Scalars)); st->keep.push_back(c.share<Object>(hpx, Kind::Scalars)); }
  const DevScalars dw = host ? stage_slot_w(c, parity, w_host, nw, st->in, true) : DevScalars{w->buf.as<uint32_t>(), w->n};
  const DevScalars dp = host ? slot_px_from_host(c, parity, px_host, npx, st->in) : DevScalars{px->buf.as<uint32_t>(), px->n};
  const int rc = pinocchio_enqueue(c, pk, dw, dp, Shard{}, parity, false, true, *st);
  if (rc != GS_OK) return rc;
  if (st->in.th2d) st->in.th2d->stop();
  mark_ticket_reads(st->streams, w, px);
  st->ticket = c.new_ticket();
  *ticket = st->ticket;
  c.inflight[parity] = std::move(st);
  return GS_OK;
}

int gs_pinocchio_prove_begin(gs_handle hpk, gs_handle hw, gs_handle hpx, uint64_t* ticket) {
  return guarded([&](Ctx& c) -> int {
    return pinocchio_begin_impl(c, "gs_pinocchio_prove_begin", hpk, hw, hpx, nullptr, 0, nullptr, 0, ticket);
  }, true, true, hpk);
}

int gs_pinocchio_prove_host_begin(gs_handle hpk, const uint64_t* w, size_t nw, const uint64_t* px, size_t npx, uint64_t* ticket) {
  return guarded([&](Ctx& c) -> int {
    if (!w || !px) return fail(GS_ERR_ARG, "gs_pinocchio_prove_host_begin: null w or px");
    return pinocchio_begin_impl(c, "gs_pinocchio_prove_host_begin", hpk, 0, 0, w, nw, px, npx, ticket);
  }, true, true, hpk);
}

int gs_pinocchio_prove_end(uint64_t ticket, uint64_t out_proof[72], int inf[8]) {
  wait_ticket_unlocked(ticket);          // the device wait, outside the context lock (runtime.h)
  return guarded([&](Ctx& c) -> int {
    if (!out_proof || !inf) return fail(GS_ERR_ARG, "null argument");
    int parity = -1;
    for (int p = 0; p < Ctx::kMaxInFlight; ++p) if (c.inflight[p] && c.inflight[p]->ticket == ticket) parity = p;
    if (parity < 0) return fail(GS_ERR_ARG, "gs_pinocchio_prove_end: unknown ticket %llu", (unsigned long long)ticket);
    if (!dynamic_cast<PinInFlight*>(c.inflight[parity].get()))
      return fail(GS_ERR_ARG, "gs_pinocchio_prove_end: ticket %llu is not a Pinocchio proof", (unsigned long long)ticket);
    std::shared_ptr<InFlightBase> base = std::move(c.inflight[parity]);
    PinInFlight& st = static_cast<PinInFlight&>(*base);
    reset_timing(c);
    int rc = pinocchio_collect(c, st, out_proof, inf);
    if (rc == kRetryExact && st.exact_route) { c.timing.fallbacks += 1; rc = st.exact_route(c, out_proof, inf); }
    return rc;
  }, true, true, ticket);
}

int gs_pinocchio_prove_resident(gs_handle hpk, gs_handle hw, gs_handle hpx, uint64_t out_proof[72], int inf[8]) {
  return guarded([&](Ctx& c) -> int {
    PinocchioPkObj* pk = c.get<PinocchioPkObj>(hpk, Kind::PinocchioPk);
    Scalars* w = c.get<Scalars>(hw, Kind::Scalars);
    Scalars* px = c.get<Scalars>(hpx, Kind::Scalars);
    if (!pk || !w || !px) return fail(GS_ERR_ARG, "gs_pinocchio_prove_resident: bad handle");
    if (!out_proof || !inf) return fail(GS_ERR_ARG, "null argument");
    reset_timing(c);
    return pinocchio_prove_impl(c, pk, DevScalars{w->buf.as<uint32_t>(), w->n}, DevScalars{px->buf.as<uint32_t>(), px->n}, out_proof, inf);
  }, true, false, hpk);
}

// ---- Pinocchio over several GPUs (SURVEY 8e applied to snark.go:254-289) ------------------------------------------------------
// A Pinocchio proof IS its eight MSM sums (there is no tail): rank k of N sums its term ranges, the eight partial points of the
// ranks add up to the proof.  Key slices, partial sums (from px, or from the owner's slice of H's values), and the addition.
static void pin_set_shard(PinocchioPkObj& pk, size_t index, size_t count) {
  Shard sh; sh.index = index; sh.count = count;
  size_t lo, hi;
  pk.shard_index = index; pk.shard_count = count;
  shard_range(pk.nvars, sh, lo, hi);
  pk.w_lo = lo; pk.n_w = hi - lo;
  shard_range(pk.ng1t, sh, lo, hi);
  pk.h_lo = lo; pk.n_h = hi - lo;
}

static int pinocchio_pk_shard_impl(Ctx& c, Ctx& from, PinocchioPkObj* full, size_t shard_index, size_t shard_count, gs_handle* out) {
  if (!full || !out) return fail(GS_ERR_ARG, "gs_pinocchio_pk_shard: bad proving-key handle or null output");
  if (full->shard_count != 1) return fail(GS_ERR_ARG, "gs_pinocchio_pk_shard: the source key is itself a slice");
  if (shard_count == 0 || shard_index >= shard_count) return fail(GS_ERR_ARG, "gs_pinocchio_pk_shard: bad shard %zu of %zu", shard_index, shard_count);
  auto pk = std::make_unique<PinocchioPkObj>();
  pk->nvars = full->nvars; pk->npublic = full->npublic; pk->nz = full->nz; pk->ng1t = full->ng1t;
  pin_set_shard(*pk, shard_index, shard_count);
  // (A / Ap of the full key already hold infinity for i <= NPublic, snark.go:265: the slices inherit it)
  const DevBuf* src[6] = {&full->a, &full->ap, &full->bp, &full->c, &full->cp, &full->kp};
  DevBuf* dst[6] = {&pk->a, &pk->ap, &pk->bp, &pk->c, &pk->cp, &pk->kp};
  for (int i = 0; i < 6; ++i) copy_slice(c, from, PkSrc{src[i], pk->w_lo}, pk->n_w, kG1Aff, *dst[i]);
  copy_slice(c, from, PkSrc{&full->b2, pk->w_lo}, pk->n_w, kG2Aff, pk->b2);
  copy_slice(c, from, PkSrc{&full->g1t, pk->h_lo}, pk->n_h, kG1Aff, pk->g1t);
  if (full->n_eval) {
    Shard sh; sh.index = shard_index; sh.count = shard_count;
    size_t lo, hi;
    shard_range(full->n_eval, sh, lo, hi);
    pk->n_eval = full->n_eval; pk->e_lo = lo; pk->n_e = hi - lo;
    copy_slice(c, from, PkSrc{&full->g1t_eval, lo}, pk->n_e, kG1Aff, pk->g1t_eval);
  }
  {                                                     // Z travels with every slice
    DevBuf zc(std::max<size_t>(full->nz, 1) * 32);
    copy_between(c, zc.p, from, full->z.b_std.p, full->nz * 32);
    divisor_init(c, pk->z, zc.as<uint32_t>(), full->nz);
    GS_HIP(hipStreamSynchronize(c.stream));             // `zc` is released here
  }
  GS_HIP(hipStreamSynchronize(c.stream));
  pinocchio_pk_scan_sparsity(c, *pk);
  *out = c.put(std::move(pk));
  return GS_OK;
}

int gs_pinocchio_pk_shard(gs_handle hfull, size_t shard_index, size_t shard_count, gs_handle* out) {
  return guarded([&](Ctx& c) -> int {
    return pinocchio_pk_shard_impl(c, c, c.get<PinocchioPkObj>(hfull, Kind::PinocchioPk), shard_index, shard_count, out);
  }, true, false, hfull);
}

int gs_pinocchio_pk_shard_to(gs_handle hfull, size_t shard_index, size_t shard_count, int target_device, gs_handle* out) {
  return guarded_pair(hfull, target_device, [&](Ctx& src, Ctx& dst) -> int {
    return pinocchio_pk_shard_impl(dst, src, src.get<PinocchioPkObj>(hfull, Kind::PinocchioPk), shard_index, shard_count, out);
  });
}

// the eight sums over this rank's term ranges, in the layout of a proof (PiA | PiAp | PiB | PiBp | PiC | PiCp | PiH | PiKp)
int gs_pinocchio_prove_partials(gs_handle hpk, gs_handle hw, gs_handle hpx, size_t shard_index, size_t shard_count, uint64_t out_sums[72], int inf[8]) {
  return guarded([&](Ctx& c) -> int {
    PinocchioPkObj* pk = c.get<PinocchioPkObj>(hpk, Kind::PinocchioPk);
    Scalars* w = c.get<Scalars>(hw, Kind::Scalars);
    Scalars* px = c.get<Scalars>(hpx, Kind::Scalars);
    if (!pk || !w || !px) return fail(GS_ERR_ARG, "gs_pinocchio_prove_partials: bad handle");
    if (!out_sums || !inf || shard_count == 0 || shard_index >= shard_count) return fail(GS_ERR_ARG, "gs_pinocchio_prove_partials: bad shard or null output");
    reset_timing(c);
    Shard sh; sh.index = shard_index; sh.count = shard_count;
    return pinocchio_prove_impl(c, pk, DevScalars{w->buf.as<uint32_t>(), w->n}, DevScalars{px->buf.as<uint32_t>(), px->n}, out_sums, inf, sh);
  }, true, false, hpk);
}

// the owner's polynomial stage (as gs_groth16_witness_values: the same H, the key only says which Z and how many constraints)
int gs_pinocchio_witness_values(gs_handle hpk, gs_handle hr1cs, gs_handle hw, gs_handle* hv_inout, uint32_t* violated) {
  return guarded([&](Ctx& c) -> int {
    PinocchioPkObj* pk = c.get<PinocchioPkObj>(hpk, Kind::PinocchioPk);
    if (!pk) return fail(GS_ERR_ARG, "gs_pinocchio_witness_values: bad proving-key handle");
    return witness_values_impl(c, "gs_pinocchio_witness_values", pk->nz, pk->n_eval, hr1cs, hw, hv_inout, violated);
  }, true, true, hpk);
}

// ... and the ranks' sums with PiH over THEIR slice of those values (no polynomial work on this rank)
int gs_pinocchio_prove_partials_values(gs_handle hpk, gs_handle hw, gs_handle hv_slice, size_t shard_index, size_t shard_count,
                                       uint64_t out_sums[72], int inf[8]) {
  return guarded([&](Ctx& c) -> int {
    PinocchioPkObj* pk = c.get<PinocchioPkObj>(hpk, Kind::PinocchioPk);
    Scalars* w = c.get<Scalars>(hw, Kind::Scalars);
    Scalars* hv = c.get<Scalars>(hv_slice, Kind::Scalars);
    if (!pk || !w || !hv) return fail(GS_ERR_ARG, "gs_pinocchio_prove_partials_values: bad handle");
    if (!out_sums || !inf || shard_count == 0 || shard_index >= shard_count) return fail(GS_ERR_ARG, "gs_pinocchio_prove_partials_values: bad shard or null output");
    if (pk->n_eval == 0) return fail(GS_ERR_SHAPE, "gs_pinocchio_prove_partials_values: the key has no evaluation-basis array");
    Shard sh; sh.index = shard_index; sh.count = shard_count;
    size_t lo, hi;
    if (pk->shard_count > 1) { lo = pk->e_lo; hi = pk->e_lo + pk->n_e; } else shard_range(pk->n_eval, sh, lo, hi);
    if (hv->n != hi - lo) return fail(GS_ERR_SHAPE, "gs_pinocchio_prove_partials_values: shard %zu of %zu covers %zu of the %zu values, the vector holds %zu",
                                      shard_index, shard_count, hi - lo, pk->n_eval, hv->n);
    reset_timing(c);
    DevScalars dh{nullptr, 0};
    dh.hv_slice = hv->buf.as<uint32_t>();
    return pinocchio_prove_impl(c, pk, DevScalars{w->buf.as<uint32_t>(), w->n}, dh, out_sums, inf, sh);
  }, true, false, hpk);
}

// the addition: n records of eight sums (the layout above, n x 72 words and n x 8 flags) -> the proof.  Host arithmetic on 8 n points.
int gs_pinocchio_combine(const uint64_t* sums, const int* inf_in, size_t n, uint64_t out_proof[72], int inf[8]) {
  if (!sums || !inf_in || !out_proof || !inf || n == 0) return fail(GS_ERR_ARG, "gs_pinocchio_combine: null argument or no records");
  static const int off[8] = {0, 8, 16, 32, 40, 48, 56, 64};
  std::vector<uint64_t> pts(n * 16);
  std::vector<int> fl(n);
  for (int k = 0; k < 8; ++k) {
    const bool g2 = k == 2;
    const size_t words = g2 ? 16 : 8;
    for (size_t i = 0; i < n; ++i) { memcpy(&pts[i * words], sums + i * 72 + off[k], words * 8); fl[i] = inf_in[i * 8 + k]; }
    const int rc = g2 ? gs_g2_sum_affine(pts.data(), fl.data(), n, out_proof + off[k], &inf[k]) : gs_g1_sum_affine(pts.data(), fl.data(), n, out_proof + off[k], &inf[k]);
    if (rc != GS_OK) return rc;
  }
  return GS_OK;
}

// PolynomialField.LagrangeInterpolation (r1csqap.go:150-158): n values at the nodes 1..n -> n coefficients.
int gs_lagrange_interpolation(const uint64_t* values, size_t n, uint64_t* coeffs) {
  return guarded([&](Ctx& c) -> int {
    if (n == 0) return GS_OK;
    if (!values || !coeffs) return fail(GS_ERR_ARG, "gs_lagrange_interpolation: null argument");
    if (n >= (1ull << 26)) return fail(GS_ERR_ARG, "gs_lagrange_interpolation: too many nodes");
    const uint32_t* dv = upload_tmp(c, prove_state(c).up_a, values, n);
    prove_state(c).up_o.ensure(n * 32);
    interpolate_dev(c, dv, n, 1, prove_state(c).up_o.as<uint32_t>());
    poly_canon_dev(c, prove_state(c).up_o.as<uint32_t>(), n, 0);
    download(c, coeffs, prove_state(c).up_o.p, n);
    return GS_OK;
  });
}

int gs_zpoly(size_t deg, uint64_t* out) {
  return guarded([&](Ctx& c) -> int {
    if (!out) return fail(GS_ERR_ARG, "gs_zpoly: null output");
    if (deg >= (1ull << 26)) return fail(GS_ERR_ARG, "gs_zpoly: degree too large");
    prove_state(c).up_o.ensure((deg + 1) * 32);
    zpoly_dev(c, deg, prove_state(c).up_o.as<uint32_t>());
    download(c, out, prove_state(c).up_o.p, deg + 1);
    return GS_OK;
  });
}

// Sparse R1CS + witness -> ax, bx, cx, px = ax * bx - cx: the scalable replacement of the dense
// R1CSToQAP + CombinePolynomials pair (r1csqap.go:161-210).  CombinePolynomials' ax = sum_i w_i alpha_i(x) is the
// interpolant of the values (A w)_j at the nodes j = 1..n, so the m x n coefficient matrices are never formed.
static const char* validate_csr(size_t n, size_t m, const uint32_t* rowptr, const uint32_t* col, const uint64_t* val) {
  if (!rowptr) return "null row_ptr";
  if (rowptr[0] != 0) return "row_ptr[0] must be 0";
  for (size_t r = 0; r < n; ++r) if (rowptr[r + 1] < rowptr[r]) return "row_ptr not monotone";
  const size_t nnz = rowptr[n];
  if (nnz && (!col || !val)) return "null column/value array";
  for (size_t e = 0; e < nnz; ++e) if (col[e] >= m) return "column index out of range";
  return nullptr;
}

static int r1cs_upload_impl(Ctx& c, size_t n, size_t m, const uint32_t* const rp[3], const uint32_t* const cl[3], const uint64_t* const vl[3],
                            R1csObj& o) {
  if (n == 0 || m == 0) return fail(GS_ERR_ARG, "empty R1CS");
  if (n >= (1ull << 26) || m >= (1ull << 31)) return fail(GS_ERR_ARG, "R1CS too large");
  for (int k = 0; k < 3; ++k)
    if (const char* e = validate_csr(n, m, rp[k], cl[k], vl[k])) return fail(GS_ERR_ARG, "R1CS matrix %c: %s", "ABC"[k], e);
  o.n = n; o.m = m;
  for (int k = 0; k < 3; ++k) {
    const size_t nnz = rp[k][n];
    o.nnz[k] = nnz;
    o.rowptr[k].alloc((n + 1) * 4);
    o.col[k].alloc(std::max<size_t>(nnz, 1) * 4);
    o.val[k].alloc(std::max<size_t>(nnz, 1) * 32);
    GS_HIP(hipMemcpyAsync(o.rowptr[k].p, rp[k], (n + 1) * 4, hipMemcpyHostToDevice, c.stream));
    if (nnz) {
      GS_HIP(hipMemcpyAsync(o.col[k].p, cl[k], nnz * 4, hipMemcpyHostToDevice, c.stream));
      GS_HIP(hipMemcpyAsync(o.val[k].p, vl[k], nnz * 32, hipMemcpyHostToDevice, c.stream));
    }
  }
  GS_HIP(hipStreamSynchronize(c.stream));
  return GS_OK;
}

// w (standard form, m elements, device) -> o.vals = [A w | B w | C w] (n each, standard form): the values of ax, bx, cx at the
// nodes 1..n (CombinePolynomials' sum_i w_i alpha_i(x) evaluated there, r1csqap.go:191-210)
static void r1cs_values_dev(Ctx& c, R1csObj& o, const uint32_t* w_dev) {
  const size_t n = o.n, m = o.m;
  o.w_mont.ensure(m * 32); o.vals.ensure(3 * n * 32);
  GS_HIP(hipMemcpyAsync(o.w_mont.p, w_dev, m * 32, hipMemcpyDeviceToDevice, c.stream));
  poly_canon_dev(c, o.w_mont.as<uint32_t>(), m, 1);                                         // w -> Montgomery
  for (int k = 0; k < 3; ++k)
    spmv_dev(c, o.rowptr[k].as<uint32_t>(), o.col[k].as<uint32_t>(), o.val[k].as<uint32_t>(), o.w_mont.as<uint32_t>(), n, m,
             o.vals.as<uint32_t>() + k * n * 8);
}

// w -> o.coef = [ax | bx | cx] (n each) and px_out (2n - 1), canonical standard form
static void r1cs_px_dev(Ctx& c, R1csObj& o, const uint32_t* w_dev, uint32_t* px_out) {
  const size_t n = o.n, npx = 2 * n - 1;
  r1cs_values_dev(c, o, w_dev);
  o.coef.ensure(3 * n * 32); o.prod.ensure(npx * 32);
  interpolate_dev(c, o.vals.as<uint32_t>(), n, 3, o.coef.as<uint32_t>());
  uint32_t* A = o.coef.as<uint32_t>();
  uint32_t* B = A + n * 8;
  uint32_t* C = B + n * 8;
  poly_mul_dev(c, A, n, Form::Std, B, n, Form::Std, o.prod.as<uint32_t>());
  poly_addsub_dev(c, o.prod.as<uint32_t>(), npx, C, n, true, px_out);
  poly_canon_dev(c, px_out, npx, 0);
  poly_canon_dev(c, A, 3 * n, 0);
}

int gs_r1cs_to_px(size_t n, size_t m,
                  const uint32_t* a_rowptr, const uint32_t* a_col, const uint64_t* a_val,
                  const uint32_t* b_rowptr, const uint32_t* b_col, const uint64_t* b_val,
                  const uint32_t* c_rowptr, const uint32_t* c_col, const uint64_t* c_val,
                  const uint64_t* w, uint64_t* ax, uint64_t* bx, uint64_t* cx, uint64_t* px) {
  return guarded([&](Ctx& c) -> int {
    if (!w || !px) return fail(GS_ERR_ARG, "gs_r1cs_to_px: null argument");
    const uint32_t* rp[3] = {a_rowptr, b_rowptr, c_rowptr};
    const uint32_t* cl[3] = {a_col, b_col, c_col};
    const uint64_t* vl[3] = {a_val, b_val, c_val};
    R1csObj o;
    const int rc = r1cs_upload_impl(c, n, m, rp, cl, vl, o);
    if (rc != GS_OK) return rc;
    const size_t npx = 2 * n - 1;
    const uint32_t* dw = upload_tmp(c, prove_state(c).up_w, w, m);
    prove_state(c).up_o.ensure(npx * 32);
    r1cs_px_dev(c, o, dw, prove_state(c).up_o.as<uint32_t>());
    const uint32_t* A = o.coef.as<uint32_t>();
    if (ax) GS_HIP(hipMemcpyAsync(ax, A, n * 32, hipMemcpyDeviceToHost, c.stream));
    if (bx) GS_HIP(hipMemcpyAsync(bx, A + n * 8, n * 32, hipMemcpyDeviceToHost, c.stream));
    if (cx) GS_HIP(hipMemcpyAsync(cx, A + 2 * n * 8, n * 32, hipMemcpyDeviceToHost, c.stream));
    GS_HIP(hipMemcpyAsync(px, prove_state(c).up_o.p, npx * 32, hipMemcpyDeviceToHost, c.stream));
    GS_HIP(hipStreamSynchronize(c.stream));
    return GS_OK;
  });
}

// The per-circuit / per-proof split of the same computation: the R1CS is uploaded and validated once ...
int gs_r1cs_upload(size_t n, size_t m,
                   const uint32_t* a_rowptr, const uint32_t* a_col, const uint64_t* a_val,
                   const uint32_t* b_rowptr, const uint32_t* b_col, const uint64_t* b_val,
                   const uint32_t* c_rowptr, const uint32_t* c_col, const uint64_t* c_val, gs_handle* out) {
  return guarded([&](Ctx& c) -> int {
    if (!out) return fail(GS_ERR_ARG, "gs_r1cs_upload: null output");
    const uint32_t* rp[3] = {a_rowptr, b_rowptr, c_rowptr};
    const uint32_t* cl[3] = {a_col, b_col, c_col};
    const uint64_t* vl[3] = {a_val, b_val, c_val};
    auto o = std::make_unique<R1csObj>();
    const int rc = r1cs_upload_impl(c, n, m, rp, cl, vl, *o);
    if (rc != GS_OK) return rc;
    *out = c.put(std::move(o));
    return GS_OK;
  });
}

// ... and every proof only turns its resident witness into the resident px (nothing crosses PCIe).  *px_inout: 0 to create the
// 2n - 1 coefficient vector, or a handle from an earlier call to overwrite.
int gs_r1cs_px(gs_handle hr1cs, gs_handle hw, gs_handle* px_inout) {
  return guarded([&](Ctx& c) -> int {
    R1csObj* o = c.get<R1csObj>(hr1cs, Kind::R1cs);
    Scalars* w = c.get<Scalars>(hw, Kind::Scalars);
    if (!o || !w || !px_inout) return fail(GS_ERR_ARG, "gs_r1cs_px: bad handle");
    if (w->n != o->m) return fail(GS_ERR_SHAPE, "len(w) = %zu but the system has %zu variables", w->n, o->m);
    const size_t npx = 2 * o->n - 1;
    Scalars* px = nullptr;
    if (*px_inout) {
      px = c.get<Scalars>(*px_inout, Kind::Scalars);
      if (!px || px->n != npx) return fail(GS_ERR_ARG, "gs_r1cs_px: the px handle does not hold 2n - 1 = %zu coefficients", npx);
    } else {
      auto fresh = std::make_unique<Scalars>();
      fresh->n = npx;
      fresh->buf.alloc(npx * 32);
      px = fresh.get();
      *px_inout = c.put(std::move(fresh));
    }
    // On the stream that carries every proof's polynomial stage: stream order keeps the engine's workspaces consistent, so
    // the px of proof k+1 may be computed while proofs are in flight (it queues behind their H(x), not behind their MSMs).
    StreamScope sc(c, c.aux_stream[1]);
    PhaseTimer t(c.stream);
    r1cs_px_dev(c, *o, w->buf.as<uint32_t>(), px->buf.as<uint32_t>());
    t.stop();
    GS_HIP(hipStreamSynchronize(c.stream));
    if (!c.any_inflight()) reset_timing(c);
    c.timing.poly_ms = t.ms();
    c.timing.total_ms = c.timing.poly_ms;
    return GS_OK;
  }, true, true, hr1cs);
}

// R1CS + witness -> proof in one call: the px stage (three interpolations and a product, r1csqap.go:161-210) runs on the aux
// stream BEHIND which H(x) waits anyway, while the main stream is already accumulating the four sums over w, which do not
// need px.  *px_inout as in gs_r1cs_px (px stays available to the caller, as CombinePolynomials returns it).
int gs_groth16_prove_r1cs(gs_handle hpk, gs_handle hr1cs, gs_handle hw, gs_handle* px_inout, const uint64_t r[4], const uint64_t s[4],
                          uint64_t out_proof[32], int inf[3]) {
  return guarded([&](Ctx& c) -> int {
    GrothPkObj* pk = c.get<GrothPkObj>(hpk, Kind::GrothPk);
    R1csObj* o = c.get<R1csObj>(hr1cs, Kind::R1cs);
    Scalars* w = c.get<Scalars>(hw, Kind::Scalars);
    if (!pk || !o || !w || !px_inout) return fail(GS_ERR_ARG, "gs_groth16_prove_r1cs: bad handle");
    if (!r || !s || !out_proof || !inf) return fail(GS_ERR_ARG, "null argument");
    if (w->n != o->m) return fail(GS_ERR_SHAPE, "len(w) = %zu but the system has %zu variables", w->n, o->m);
    const size_t npx = 2 * o->n - 1;
    Scalars* px = nullptr;
    if (*px_inout) {
      px = c.get<Scalars>(*px_inout, Kind::Scalars);
      if (!px || px->n != npx) return fail(GS_ERR_ARG, "gs_groth16_prove_r1cs: the px handle does not hold 2n - 1 = %zu coefficients", npx);
    } else {
      auto fresh = std::make_unique<Scalars>();
      fresh->n = npx;
      fresh->buf.alloc(npx * 32);
      px = fresh.get();
      *px_inout = c.put(std::move(fresh));
    }
    reset_timing(c);
    DevScalars dp{px->buf.as<uint32_t>(), npx};
    const uint32_t* wdev = w->buf.as<uint32_t>();
    uint32_t* pxdev = px->buf.as<uint32_t>();
    dp.produce = [o, wdev, pxdev](Ctx& cc) { r1cs_px_dev(cc, *o, wdev, pxdev); };
    return groth16_prove_impl(c, pk, DevScalars{wdev, w->n}, dp, r, s, out_proof, inf);
  }, true, false, hpk);
}

// Witness -> proof without ever forming px (the fast form of CombinePolynomials + DivisorPolynomial, r1csqap.go:191-216, for a
// witness that satisfies the R1CS -- the only case in which a proof means anything):
//   * key with an evaluation-basis copy of PowersTauDelta (gs_groth16_setup builds one): the values of H at the nodes n+1..2n come
//     out of ONE batched convolution of the constraint values [A w | B w | C w], and the h-MSM runs over those values.  No
//     interpolation, no Taylor shift, no host wait: the violated-constraint count is read when the proof is collected.
//   * key without one: H's coefficients from the same values by one interpolation and a Taylor shift (poly.h, hx_direct_dev).
// When a constraint is violated the call takes the exact route of gs_groth16_prove_r1cs and returns the same (meaningless) proof
// the reference would.  Same result as gs_r1cs_px + gs_groth16_prove_resident either way.
static bool hx_shape(size_t n, size_t nz) { return n >= 2 && nz >= 1 && (nz - 1 == n - 1 || nz - 1 == n); }

// the three ways to the h-scalars of a witness proof, in the order the prover tries them
static DevScalars witness_scalars(R1csObj* o, const uint32_t* wdev, size_t nz, bool eval, std::function<uint32_t*(Ctx&)> px_buffer) {
  const size_t npx = 2 * o->n - 1, dz = nz - 1;
  DevScalars dp{nullptr, npx};
  if (eval)
    dp.produce_hv = [o, wdev, dz](Ctx& cc, uint32_t* hv, uint32_t* bad) {
      r1cs_values_dev(cc, *o, wdev);
      r1cs_check_dev(cc, o->vals.as<uint32_t>(), o->n, dz, bad);
      hx_values_dev(cc, o->vals.as<uint32_t>(), o->n, dz, hv);
    };
  dp.produce_hx = [o, wdev, dz](Ctx& cc, uint32_t* hx) { r1cs_values_dev(cc, *o, wdev); return hx_direct_dev(cc, o->vals.as<uint32_t>(), o->n, dz, hx); };
  dp.produce = [o, wdev, px_buffer](Ctx& cc) { r1cs_px_dev(cc, *o, wdev, px_buffer(cc)); };
  return dp;
}
// px of the exact route: one buffer per slot (the three tickets and the blocking entry points), allocated only if that route is
// ever taken.  Round 3 handed every in-flight ticket -- and the blocking entry points' host uploads -- the SAME staging buffer and
// leaned on all polynomial stages being serialised on one stream; a slot's buffer is only ever touched by the
// operation that owns the slot, and a slot is re-used only after its operation was collected.
static uint32_t* exact_px_buffer(Ctx& c, R1csObj* o, int slot) {
  const size_t npx = 2 * o->n - 1;
  o->prod.ensure(npx * 32);
  DevBuf& b = prove_state(c).exact_px[slot % Ctx::kSlots];
  b.ensure(npx * 32);
  return b.as<uint32_t>();
}

int gs_groth16_prove_witness(gs_handle hpk, gs_handle hr1cs, gs_handle hw, const uint64_t r[4], const uint64_t s[4], uint64_t out_proof[32], int inf[3]) {
  return guarded([&](Ctx& c) -> int {
    GrothPkObj* pk = c.get<GrothPkObj>(hpk, Kind::GrothPk);
    R1csObj* o = c.get<R1csObj>(hr1cs, Kind::R1cs);
    Scalars* w = c.get<Scalars>(hw, Kind::Scalars);
    if (!pk || !o || !w) return fail(GS_ERR_ARG, "gs_groth16_prove_witness: bad handle");
    if (!r || !s || !out_proof || !inf) return fail(GS_ERR_ARG, "null argument");
    if (w->n != o->m) return fail(GS_ERR_SHAPE, "len(w) = %zu but the system has %zu variables", w->n, o->m);
    reset_timing(c);
    const uint32_t* wdev = w->buf.as<uint32_t>();
    const bool eval = c.eval_basis && pk->shard_count == 1 && pk->n_eval == o->n && hx_shape(o->n, pk->nz);
    uint32_t* pxdev = exact_px_buffer(c, o, c.blocking_slot());
    DevScalars dp = witness_scalars(o, wdev, pk->nz, eval, [pxdev](Ctx&) { return pxdev; });
    dp.p = pxdev;
    return groth16_prove_impl(c, pk, DevScalars{wdev, w->n}, dp, r, s, out_proof, inf);
  }, true, false, hpk);
}

// The same, pipelined: a ticket for gs_groth16_prove_end (which also runs the exact route, blocking, should the witness turn out
// to violate a constraint).  With an evaluation-basis key nothing in here waits for the device.
static int groth16_witness_begin_impl(Ctx& c, const char* fn, gs_handle hpk, gs_handle hr1cs, gs_handle hw, const uint64_t* w_host, size_t nw_host,
                                      const uint64_t r[4], const uint64_t s[4], uint64_t* ticket) {
  GrothPkObj* pk = c.get<GrothPkObj>(hpk, Kind::GrothPk);
  R1csObj* o = c.get<R1csObj>(hr1cs, Kind::R1cs);
  Scalars* w = w_host ? nullptr : c.get<Scalars>(hw, Kind::Scalars);
  if (!pk || !o || (!w_host && !w)) return fail(GS_ERR_ARG, "%s: bad handle", fn);
  if (!r || !s || !ticket) return fail(GS_ERR_ARG, "null argument");
  const size_t nw = w_host ? nw_host : w->n;
  if (nw != o->m) return fail(GS_ERR_SHAPE, "len(w) = %zu but the system has %zu variables", nw, o->m);
  if (nw != pk->nvars) return fail(GS_ERR_SHAPE, "len(w) = %zu but the key has %zu variables", nw, pk->nvars);
  if (pk->shard_count != 1) return fail(GS_ERR_ARG, "%s: the key is a slice", fn);
  const int parity = c.free_parity();
  if (parity < 0) return fail(GS_ERR_BUSY, "%s: three operations are already outstanding; call gs_groth16_prove_end first", fn);
  const bool eval = c.eval_basis && pk->n_eval == o->n && hx_shape(o->n, pk->nz);
  auto st = std::make_unique<GrothInFlight>();
  memcpy(st->r, r, 32); memcpy(st->s, s, 32);
  st->with_tail = true;
  GrothInFlight* raw = st.get();
  raw->pk = pk;
  raw->keep = {c.share<Object>(hpk, Kind::GrothPk), c.share<Object>(hr1cs, Kind::R1cs)};
  if (w) raw->keep.push_back(c.share<Object>(hw, Kind::Scalars));
  raw->fpre = std::async(std::launch::async, [raw] { groth16_tail_pre(raw->pk, raw->r, raw->s, raw->pre); });
  raw->early.pk = pk; raw->early.r = raw->r; raw->early.s = raw->s; raw->early.pre = &raw->pre; raw->early.fpre = &raw->fpre;
  // (a host witness lives in the slot's buffer until the ticket is collected -- the exact-route retry below still finds it there)
  const DevScalars dw = w_host ? stage_slot_w(c, parity, w_host, nw, raw->in) : DevScalars{w->buf.as<uint32_t>(), nw};
  const uint32_t* wdev = dw.p;
  const size_t nz = pk->nz;
  DevScalars dp = witness_scalars(o, wdev, nz, eval, [o, parity](Ctx& cc) { return exact_px_buffer(cc, o, parity); });
  if (!eval) {                               // the monomial route may have to write px at once (its check is a host wait inside enqueue)
    dp.p = exact_px_buffer(c, o, parity);
  } else {
    dp.produce_hx = nullptr; dp.produce = nullptr;
    raw->exact_route = [pk, o, wdev, nw, nz](Ctx& cc, GrothSums& sums) {
      uint32_t* pxdev = exact_px_buffer(cc, o, cc.blocking_slot());     // the retry is a blocking proof at collection time
      DevScalars ex = witness_scalars(o, wdev, nz, false, [pxdev](Ctx&) { return pxdev; });
      ex.p = pxdev;
      return groth16_sums_impl(cc, pk, DevScalars{wdev, nw}, ex, Shard{}, sums);
    };
  }
  const int rc = groth16_enqueue(c, pk, dw, dp, Shard{}, parity, false, true, *raw);
  if (rc != GS_OK) return rc;
  if (raw->in.th2d) raw->in.th2d->stop();
  mark_ticket_reads(raw->streams, w, nullptr);
  st->ticket = c.new_ticket();
  *ticket = st->ticket;
  c.inflight[parity] = std::move(st);
  return GS_OK;
}

int gs_groth16_prove_witness_begin(gs_handle hpk, gs_handle hr1cs, gs_handle hw, const uint64_t r[4], const uint64_t s[4], uint64_t* ticket) {
  return guarded([&](Ctx& c) -> int {
    return groth16_witness_begin_impl(c, "gs_groth16_prove_witness_begin", hpk, hr1cs, hw, nullptr, 0, r, s, ticket);
  }, true, true, hpk);
}

// The reference's call shape for a server that keeps circuit and key resident: every call brings a NEW witness in host memory
// (cli/main.go:480-501 computes w per proof) and nothing else -- a host-buffer ticket (see stage_slot_w).  Collect with gs_groth16_prove_end.
int gs_groth16_prove_witness_host_begin(gs_handle hpk, gs_handle hr1cs, const uint64_t* w, size_t nw, const uint64_t r[4], const uint64_t s[4],
                                        uint64_t* ticket) {
  return guarded([&](Ctx& c) -> int {
    if (!w) return fail(GS_ERR_ARG, "gs_groth16_prove_witness_host_begin: null witness");
    if (nw >= (1ull << 31)) return fail(GS_ERR_ARG, "gs_groth16_prove_witness_host_begin: too many scalars");
    return groth16_witness_begin_impl(c, "gs_groth16_prove_witness_host_begin", hpk, hr1cs, 0, w, nw, r, s, ticket);
  }, true, true, hpk);
}

// ... and the blocking form: witness in host memory -> proof (the upload is part of the call, as in gs_groth16_prove).
int gs_groth16_prove_witness_host(gs_handle hpk, gs_handle hr1cs, const uint64_t* w, size_t nw, const uint64_t r[4], const uint64_t s[4],
                                  uint64_t out_proof[32], int inf[3]) {
  return guarded([&](Ctx& c) -> int {
    GrothPkObj* pk = c.get<GrothPkObj>(hpk, Kind::GrothPk);
    R1csObj* o = c.get<R1csObj>(hr1cs, Kind::R1cs);
    if (!pk || !o) return fail(GS_ERR_ARG, "gs_groth16_prove_witness_host: bad handle");
    if (!w || !r || !s || !out_proof || !inf) return fail(GS_ERR_ARG, "null argument");
    if (nw != o->m) return fail(GS_ERR_SHAPE, "len(w) = %zu but the system has %zu variables", nw, o->m);
    reset_timing(c);
    const uint32_t* wdev = upload_tmp(c, prove_state(c).up_w, w, nw);
    const bool eval = c.eval_basis && pk->shard_count == 1 && pk->n_eval == o->n && hx_shape(o->n, pk->nz);
    uint32_t* pxdev = exact_px_buffer(c, o, c.blocking_slot());
    DevScalars dp = witness_scalars(o, wdev, pk->nz, eval, [pxdev](Ctx&) { return pxdev; });
    dp.p = pxdev;
    return groth16_prove_impl(c, pk, DevScalars{wdev, nw}, dp, r, s, out_proof, inf);
  }, true, false, hpk);
}

// snark.GenerateProofs straight from the witness (the Pinocchio twin of gs_groth16_prove_witness): CombinePolynomials + Div
// (r1csqap.go:191-216, snark.go:280) collapse into H's values (evaluation-basis key) or H(x) from the constraint values; the exact
// px route when a constraint is violated.
int gs_pinocchio_prove_witness(gs_handle hpk, gs_handle hr1cs, gs_handle hw, uint64_t out_proof[72], int inf[8]) {
  return guarded([&](Ctx& c) -> int {
    PinocchioPkObj* pk = c.get<PinocchioPkObj>(hpk, Kind::PinocchioPk);
    R1csObj* o = c.get<R1csObj>(hr1cs, Kind::R1cs);
    Scalars* w = c.get<Scalars>(hw, Kind::Scalars);
    if (!pk || !o || !w) return fail(GS_ERR_ARG, "gs_pinocchio_prove_witness: bad handle");
    if (!out_proof || !inf) return fail(GS_ERR_ARG, "null argument");
    if (w->n != o->m) return fail(GS_ERR_SHAPE, "len(w) = %zu but the system has %zu variables", w->n, o->m);
    if (pk->nz == 0) return fail(GS_ERR_SHAPE, "the key has no Z");
    reset_timing(c);
    const uint32_t* wdev = w->buf.as<uint32_t>();
    const bool eval = c.eval_basis && pk->n_eval == o->n && hx_shape(o->n, pk->nz);
    uint32_t* pxdev = exact_px_buffer(c, o, c.blocking_slot());
    DevScalars dp = witness_scalars(o, wdev, pk->nz, eval, [pxdev](Ctx&) { return pxdev; });
    dp.p = pxdev;
    return pinocchio_prove_impl(c, pk, DevScalars{wdev, w->n}, dp, out_proof, inf);
  }, true, false, hpk);
}

static int pinocchio_witness_begin_impl(Ctx& c, const char* fn, gs_handle hpk, gs_handle hr1cs, gs_handle hw, const uint64_t* w_host, size_t nw_host,
                                        uint64_t* ticket) {
  PinocchioPkObj* pk = c.get<PinocchioPkObj>(hpk, Kind::PinocchioPk);
  R1csObj* o = c.get<R1csObj>(hr1cs, Kind::R1cs);
  Scalars* w = w_host ? nullptr : c.get<Scalars>(hw, Kind::Scalars);
  if (!pk || !o || (!w_host && !w) || !ticket) return fail(GS_ERR_ARG, "%s: bad handle or null ticket", fn);
  const size_t nw = w_host ? nw_host : w->n;
  if (nw != o->m) return fail(GS_ERR_SHAPE, "len(w) = %zu but the system has %zu variables", nw, o->m);
  if (nw != pk->nvars) return fail(GS_ERR_SHAPE, "len(w) = %zu but the key has %zu variables", nw, pk->nvars);
  if (pk->nz == 0) return fail(GS_ERR_SHAPE, "the key has no Z");
  const int parity = c.free_parity();
  if (parity < 0) return fail(GS_ERR_BUSY, "%s: three operations are already outstanding; call gs_pinocchio_prove_end first", fn);
  const bool eval = c.eval_basis && pk->n_eval == o->n && hx_shape(o->n, pk->nz);
  auto st = std::make_unique<PinInFlight>();
  st->keep = {c.share<Object>(hpk, Kind::PinocchioPk), c.share<Object>(hr1cs, Kind::R1cs)};
  if (w) st->keep.push_back(c.share<Object>(hw, Kind::Scalars));
  const DevScalars dw = w_host ? stage_slot_w(c, parity, w_host, nw, st->in) : DevScalars{w->buf.as<uint32_t>(), nw};
  const uint32_t* wdev = dw.p;
  const size_t nz = pk->nz;
  DevScalars dp = witness_scalars(o, wdev, nz, eval, [o, parity](Ctx& cc) { return exact_px_buffer(cc, o, parity); });
  if (!eval) {
    dp.p = exact_px_buffer(c, o, parity);
  } else {
    dp.produce_hx = nullptr; dp.produce = nullptr;
    st->exact_route = [pk, o, wdev, nw, nz](Ctx& cc, uint64_t* out, int* inf) {
      uint32_t* pxdev = exact_px_buffer(cc, o, cc.blocking_slot());     // the retry is a blocking proof at collection time
      DevScalars ex = witness_scalars(o, wdev, nz, false, [pxdev](Ctx&) { return pxdev; });
      ex.p = pxdev;
      return pinocchio_prove_impl(cc, pk, DevScalars{wdev, nw}, ex, out, inf);
    };
  }
  const int rc = pinocchio_enqueue(c, pk, dw, dp, Shard{}, parity, false, true, *st);
  if (rc != GS_OK) return rc;
  if (st->in.th2d) st->in.th2d->stop();
  mark_ticket_reads(st->streams, w, nullptr);
  st->ticket = c.new_ticket();
  *ticket = st->ticket;
  c.inflight[parity] = std::move(st);
  return GS_OK;
}

int gs_pinocchio_prove_witness_begin(gs_handle hpk, gs_handle hr1cs, gs_handle hw, uint64_t* ticket) {
  return guarded([&](Ctx& c) -> int {
    return pinocchio_witness_begin_impl(c, "gs_pinocchio_prove_witness_begin", hpk, hr1cs, hw, nullptr, 0, ticket);
  }, true, true, hpk);
}

// host-buffer tickets of snark.GenerateProofs (see gs_groth16_prove_witness_host_begin / gs_groth16_prove_host_begin)
int gs_pinocchio_prove_witness_host_begin(gs_handle hpk, gs_handle hr1cs, const uint64_t* w, size_t nw, uint64_t* ticket) {
  return guarded([&](Ctx& c) -> int {
    if (!w) return fail(GS_ERR_ARG, "gs_pinocchio_prove_witness_host_begin: null witness");
    if (nw >= (1ull << 31)) return fail(GS_ERR_ARG, "gs_pinocchio_prove_witness_host_begin: too many scalars");
    return pinocchio_witness_begin_impl(c, "gs_pinocchio_prove_witness_host_begin", hpk, hr1cs, 0, w, nw, ticket);
  }, true, true, hpk);
}

int gs_pinocchio_prove_witness_host(gs_handle hpk, gs_handle hr1cs, const uint64_t* w, size_t nw, uint64_t out_proof[72], int inf[8]) {
  return guarded([&](Ctx& c) -> int {
    PinocchioPkObj* pk = c.get<PinocchioPkObj>(hpk, Kind::PinocchioPk);
    R1csObj* o = c.get<R1csObj>(hr1cs, Kind::R1cs);
    if (!pk || !o) return fail(GS_ERR_ARG, "gs_pinocchio_prove_witness_host: bad handle");
    if (!w || !out_proof || !inf) return fail(GS_ERR_ARG, "null argument");
    if (nw != o->m) return fail(GS_ERR_SHAPE, "len(w) = %zu but the system has %zu variables", nw, o->m);
    if (pk->nz == 0) return fail(GS_ERR_SHAPE, "the key has no Z");
    reset_timing(c);
    const uint32_t* wdev = upload_tmp(c, prove_state(c).up_w, w, nw);
    const bool eval = c.eval_basis && pk->n_eval == o->n && hx_shape(o->n, pk->nz);
    uint32_t* pxdev = exact_px_buffer(c, o, c.blocking_slot());
    DevScalars dp = witness_scalars(o, wdev, pk->nz, eval, [pxdev](Ctx&) { return pxdev; });
    dp.p = pxdev;
    return pinocchio_prove_impl(c, pk, DevScalars{wdev, nw}, dp, out_proof, inf);
  }, true, false, hpk);
}

}  // extern "C"
