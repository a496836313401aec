// Several MI355X behind the C ABI (BASELINE configs[3] and [4]).
//
// The reference's prover sums independent terms (groth16/groth16.go:243-250,269-271): the term ranges are cut into
// contiguous shards, every device runs the whole Pippenger pipeline on its shard, and ONE small record per device
// -- the five partial sums of a proof (416 bytes) or one partial point of an MSM (72 / 136 bytes) -- is exchanged.
// RCCL has no curve-point reduction, so the exchange is an ncclAllGather of the records as bytes followed by
// world - 1 complete additions on the host core (a literal all-reduce cannot add curve points).
//
// Two deployment shapes share the code:
//   * one process, N logical devices (gs_init(devices, N); a Go host with one goroutine per device):
//     gs_groth16_prove_multi / gs_msm_g1_multi / gs_groth16_prove_batch drive the devices from N host threads.  The records
//     already sit in this process's memory; with gs_comm_init_local() they nevertheless travel through ncclAllGather
//     (one RCCL rank per distinct physical device), which is how the xGMI path is exercised in one process.
//   * one process per GPU (bench.py under torch.distributed.run, or N Go processes): gs_comm_init_rank() joins a
//     communicator whose unique id the host language distributed, gs_groth16_prove_sharded / gs_msm_g1_sharded compute
//     this rank's shard and gather inside the library.
// Batches of independent proofs (configs[4]) need no collective at all: gs_groth16_prove_batch round-robins them.
#include <rccl/rccl.h>

#include <algorithm>
#include <atomic>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include "runtime.h"

using namespace gs;

namespace {

struct Comm {
  std::mutex mu;
  bool local = false;                    // true: every rank lives in this process (one per distinct physical device)
  int nranks = 0, rank = -1;             // rank mode: this process's rank
  std::vector<int> phys;                 // HIP ordinal of each local rank
  std::vector<ncclComm_t> comms;         // one per local rank
  std::vector<hipStream_t> streams;
  std::vector<void*> send, recv;         // device staging per local rank
  size_t cap = 0;                        // bytes per rank the staging buffers hold
  uint64_t collectives = 0;              // ncclAllGather calls completed (tests and bench.py read it)
  bool active() const { return !comms.empty(); }
};
Comm& comm() {
  static Comm c;
  return c;
}

#define GS_NCCL(x)                                                                                         \
  do {                                                                                                     \
    ncclResult_t _r = (x);                                                                                 \
    if (_r != ncclSuccess) return fail(GS_ERR_HIP, "RCCL error %d (%s) at %s", (int)_r, ncclGetErrorString(_r), #x); \
  } while (0)
#define GS_HIPRC(x)                                                                                        \
  do {                                                                                                     \
    hipError_t _e = (x);                                                                                   \
    if (_e != hipSuccess) return fail(GS_ERR_HIP, "HIP error %d (%s) at %s", (int)_e, hipGetErrorString(_e), #x); \
  } while (0)

void comm_release(Comm& cm) {
  for (size_t i = 0; i < cm.comms.size(); ++i) {
    (void)hipSetDevice(cm.phys[i]);
    if (cm.comms[i]) (void)ncclCommDestroy(cm.comms[i]);
    if (i < cm.streams.size() && cm.streams[i]) (void)hipStreamDestroy(cm.streams[i]);
    if (i < cm.send.size() && cm.send[i]) (void)hipFree(cm.send[i]);
    if (i < cm.recv.size() && cm.recv[i]) (void)hipFree(cm.recv[i]);
  }
  cm.comms.clear(); cm.streams.clear(); cm.send.clear(); cm.recv.clear(); cm.phys.clear();
  cm.cap = 0; cm.nranks = 0; cm.rank = -1; cm.local = false;
}

int comm_staging(Comm& cm, size_t bytes_per_rank) {
  if (bytes_per_rank <= cm.cap) return GS_OK;
  const size_t cap = std::max<size_t>(4096, bytes_per_rank * 2);
  for (size_t i = 0; i < cm.comms.size(); ++i) {
    GS_HIPRC(hipSetDevice(cm.phys[i]));
    if (cm.send[i]) (void)hipFree(cm.send[i]);
    if (cm.recv[i]) (void)hipFree(cm.recv[i]);
    cm.send[i] = cm.recv[i] = nullptr;
    GS_HIPRC(hipMalloc(&cm.send[i], cap));
    GS_HIPRC(hipMalloc(&cm.recv[i], cap * (size_t)cm.nranks));
  }
  cm.cap = cap;
  return GS_OK;
}

// All ranks of this process contribute `bytes` each (send: one block per LOCAL rank, in local-rank order); every rank
// receives nranks blocks; recv gets local rank 0's copy.  Host buffers, staged through device memory: the payload is a few
// hundred bytes and the call is latency bound either way.
int comm_allgather(Comm& cm, const uint8_t* send, size_t bytes, uint8_t* recv) {
  if (!cm.active()) return fail(GS_ERR_ARG, "no communicator: call gs_comm_init_local or gs_comm_init_rank first");
  if (bytes == 0) return GS_OK;
  if (int rc = comm_staging(cm, bytes)) return rc;
  const size_t nloc = cm.comms.size();
  for (size_t i = 0; i < nloc; ++i) {
    GS_HIPRC(hipSetDevice(cm.phys[i]));
    GS_HIPRC(hipMemcpyAsync(cm.send[i], send + i * bytes, bytes, hipMemcpyHostToDevice, cm.streams[i]));
  }
  if (nloc > 1) GS_NCCL(ncclGroupStart());
  for (size_t i = 0; i < nloc; ++i) {
    GS_HIPRC(hipSetDevice(cm.phys[i]));
    GS_NCCL(ncclAllGather(cm.send[i], cm.recv[i], bytes, ncclUint8, cm.comms[i], cm.streams[i]));
  }
  if (nloc > 1) GS_NCCL(ncclGroupEnd());
  GS_HIPRC(hipSetDevice(cm.phys[0]));
  GS_HIPRC(hipMemcpyAsync(recv, cm.recv[0], bytes * (size_t)cm.nranks, hipMemcpyDeviceToHost, cm.streams[0]));
  for (size_t i = 0; i < nloc; ++i) {
    GS_HIPRC(hipSetDevice(cm.phys[i]));
    GS_HIPRC(hipStreamSynchronize(cm.streams[i]));
  }
  cm.collectives += 1;
  return GS_OK;
}

// ---- records ------------------------------------------------------------------------------------------------------
constexpr size_t kProofRecordWords = 52;       // 48 words of sums | inf[5], shard index, 0, 0 as u32 = 4 words: 416 bytes
struct ProofRecord { uint64_t sums[48]; uint32_t inf[5]; uint32_t shard; uint32_t pad[2]; };
static_assert(sizeof(ProofRecord) == kProofRecordWords * 8, "the exchanged record is 416 bytes");

struct PinRecord { uint64_t sums[72]; uint32_t inf[8]; uint32_t shard; uint32_t pad; };      // a Pinocchio rank's eight partial sums
static_assert(sizeof(PinRecord) == 616, "the exchanged Pinocchio record is 616 bytes");

template <int W> struct PointRecord { uint64_t p[W]; uint32_t inf; uint32_t shard; };
static_assert(sizeof(PointRecord<8>) == 72 && sizeof(PointRecord<16>) == 136, "72 / 136 byte partial-point records");

struct Result { int rc = GS_OK; std::string msg; };

// run f(d) for d < n on n host threads (one per logical device), collecting status + thread-local message
template <class F>
int on_devices(int n, F&& f, const char* what) {
  std::vector<Result> res(n);
  std::vector<std::thread> th;
  for (int d = 0; d < n; ++d)
    th.emplace_back([&, d] {
      res[d].rc = f(d);
      if (res[d].rc != GS_OK) res[d].msg = gs_last_error();
    });
  for (auto& t : th) t.join();
  for (int d = 0; d < n; ++d)
    if (res[d].rc != GS_OK) return fail(res[d].rc, "%s, logical device / shard %d: %s", what, d, res[d].msg.c_str());
  return GS_OK;
}

// Exchange one record per logical device.  Through RCCL when a local communicator exists and the logical devices spread
// evenly over its ranks; otherwise the records are simply already here (same process).  out: n records ordered by shard.
template <class R>
int exchange(const std::vector<R>& mine, const std::vector<int>& phys_of, std::vector<R>& out, int* used_rccl) {
  const size_t n = mine.size();
  out.assign(n, R{});
  Comm& cm = comm();
  std::lock_guard<std::mutex> lk(cm.mu);
  bool rccl = cm.active() && cm.local;
  std::vector<std::vector<R>> per(cm.phys.size());
  if (rccl) {
    for (size_t d = 0; d < n; ++d) {
      auto it = std::find(cm.phys.begin(), cm.phys.end(), phys_of[d]);
      if (it == cm.phys.end()) { rccl = false; break; }
      per[it - cm.phys.begin()].push_back(mine[d]);
    }
    for (auto& v : per) if (v.size() != per[0].size() || v.empty()) rccl = false;
  }
  if (used_rccl) *used_rccl = rccl ? 1 : 0;
  if (!rccl) { for (const R& r : mine) out[r.shard] = r; return GS_OK; }
  const size_t k = per[0].size(), bytes = k * sizeof(R);
  std::vector<uint8_t> send(per.size() * bytes), recv((size_t)cm.nranks * bytes);
  for (size_t p = 0; p < per.size(); ++p) memcpy(send.data() + p * bytes, per[p].data(), bytes);
  if (int rc = comm_allgather(cm, send.data(), bytes, recv.data())) return rc;
  const R* got = reinterpret_cast<const R*>(recv.data());
  for (size_t i = 0; i < n; ++i) {
    if (got[i].shard >= n) return fail(GS_ERR_HIP, "gathered record %zu carries shard index %u of %zu", i, got[i].shard, n);
    out[got[i].shard] = got[i];
  }
  return GS_OK;
}

int sum_proof_records(const std::vector<ProofRecord>& recs, uint64_t sums[48], int inf[5]) {
  const size_t n = recs.size();
  std::vector<uint64_t> g1(n * 8), g2(n * 16);
  std::vector<int> fl(n);
  const int g1off[4] = {0, 8, 32, 40}, g1idx[4] = {0, 1, 3, 4};
  for (int k = 0; k < 4; ++k) {
    for (size_t i = 0; i < n; ++i) { memcpy(&g1[i * 8], recs[i].sums + g1off[k], 64); fl[i] = (int)recs[i].inf[g1idx[k]]; }
    if (int rc = gs_g1_sum_affine(g1.data(), fl.data(), n, sums + g1off[k], &inf[g1idx[k]])) return rc;
  }
  for (size_t i = 0; i < n; ++i) { memcpy(&g2[i * 16], recs[i].sums + 16, 128); fl[i] = (int)recs[i].inf[2]; }
  return gs_g2_sum_affine(g2.data(), fl.data(), n, sums + 16, &inf[2]);
}

int check_devices(const gs_handle* a, const gs_handle* b, const gs_handle* c, int ndev, const char* fn, std::vector<int>& phys_of) {
  if (!a || ndev < 1 || ndev > kMaxLogicalDevices) return fail(GS_ERR_ARG, "%s: need 1 .. %d devices and their handles", fn, kMaxLogicalDevices);
  phys_of.assign(ndev, -1);
  for (int d = 0; d < ndev; ++d) {
    const int ld = handle_device(a[d]);
    std::shared_ptr<Ctx> cx = ctx_ref(ld);
    if (!cx) return fail(GS_ERR_ARG, "%s: shard %d: handle names logical device %d, which gs_init did not create", fn, d, ld);
    if ((b && handle_device(b[d]) != ld) || (c && handle_device(c[d]) != ld))
      return fail(GS_ERR_ARG, "%s: shard %d: its handles live on different logical devices", fn, d);
    phys_of[d] = cx->device;             // (`device` is written once, before the context is published)
  }
  return GS_OK;
}

}  // namespace

// Pinocchio: records -> proof, and the two deployment shapes as templates over "how a rank gets its eight sums"
static int pin_records_to_proof(const std::vector<PinRecord>& all, uint64_t out_proof[72], int inf[8]) {
  const size_t n = all.size();
  std::vector<uint64_t> sums(n * 72);
  std::vector<int> fl(n * 8);
  for (size_t i = 0; i < n; ++i) {
    memcpy(&sums[i * 72], all[i].sums, sizeof all[i].sums);
    for (int k = 0; k < 8; ++k) fl[i * 8 + k] = (int)all[i].inf[k];
  }
  return gs_pinocchio_combine(sums.data(), fl.data(), n, out_proof, inf);
}

template <class Partials>
static int pinocchio_multi(const gs_handle* pk, const gs_handle* w, const gs_handle* third, int ndev, uint64_t out_proof[72], int inf[8], int* used_rccl,
                           const char* fn, Partials&& partials) {
  std::vector<int> phys_of;
  if (int rc = check_devices(pk, w, third, ndev, fn, phys_of)) return rc;
  if (!w || !third || !out_proof || !inf) return fail(GS_ERR_ARG, "%s: null argument", fn);
  std::vector<PinRecord> mine(ndev), all;
  if (int rc = on_devices(ndev, [&](int d) {
        int f[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        const int rc = partials(d, mine[d].sums, f);
        for (int k = 0; k < 8; ++k) mine[d].inf[k] = (uint32_t)f[k];
        mine[d].shard = (uint32_t)d; mine[d].pad = 0;
        return rc;
      }, fn)) return rc;
  if (int rc = exchange(mine, phys_of, all, used_rccl)) return rc;
  return pin_records_to_proof(all, out_proof, inf);
}

template <class Partials>
static int pinocchio_sharded(uint64_t out_proof[72], int inf[8], const char* fn, Partials&& partials) {
  if (!out_proof || !inf) return fail(GS_ERR_ARG, "%s: null argument", fn);
  int nranks = 0, rank = -1, local = 0;
  gs_comm_info(&nranks, &rank, &local, nullptr);
  if (nranks < 1 || local) return fail(GS_ERR_ARG, "%s: needs the communicator of gs_comm_init_rank (one process per GPU)", fn);
  PinRecord mine{};
  int f[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  if (int rc = partials((size_t)rank, (size_t)nranks, mine.sums, f)) return rc;
  for (int k = 0; k < 8; ++k) mine.inf[k] = (uint32_t)f[k];
  mine.shard = (uint32_t)rank;
  std::vector<PinRecord> all(nranks);
  if (int rc = gs_comm_allgather(&mine, sizeof mine, all.data())) return rc;
  for (int i = 0; i < nranks; ++i)
    if (all[i].shard != (uint32_t)i) return fail(GS_ERR_HIP, "gathered record %d carries shard index %u", i, all[i].shard);
  return pin_records_to_proof(all, out_proof, inf);
}

namespace gs {
void multi_shutdown() {
  Comm& cm = comm();
  std::lock_guard<std::mutex> lk(cm.mu);
  comm_release(cm);
}
}  // namespace gs

extern "C" {

// ---- communicator ---------------------------------------------------------------------------------------------------
int gs_comm_unique_id(uint8_t out[128]) {
  if (!out) return fail(GS_ERR_ARG, "gs_comm_unique_id: null output");
  static_assert(sizeof(ncclUniqueId) == 128, "ncclUniqueId is 128 bytes");
  ncclUniqueId id;
  GS_NCCL(ncclGetUniqueId(&id));
  memcpy(out, &id, 128);
  return GS_OK;
}

int gs_comm_init_rank(const uint8_t id_bytes[128], int nranks, int rank) {
  if (!id_bytes || nranks < 1 || rank < 0 || rank >= nranks) return fail(GS_ERR_ARG, "gs_comm_init_rank: bad id / rank %d of %d", rank, nranks);
  std::shared_ptr<Ctx> pc = ctx_ref(current_logical());
  if (!pc) return fail(GS_ERR_NOT_INIT, "gs_comm_init_rank: gs_init first (the calling thread's current logical device joins)");
  Ctx& c = *pc;
  Comm& cm = comm();
  std::lock_guard<std::mutex> lk(cm.mu);
  if (cm.active()) return fail(GS_ERR_ARG, "a communicator already exists: gs_comm_destroy first");
  ncclUniqueId id;
  memcpy(&id, id_bytes, 128);
  GS_HIPRC(hipSetDevice(c.device));
  cm.local = false; cm.nranks = nranks; cm.rank = rank;
  cm.phys = {c.device};
  cm.comms.assign(1, nullptr); cm.streams.assign(1, nullptr); cm.send.assign(1, nullptr); cm.recv.assign(1, nullptr);
  ncclResult_t r = ncclCommInitRank(&cm.comms[0], nranks, id, rank);
  if (r != ncclSuccess) { cm.comms[0] = nullptr; comm_release(cm); return fail(GS_ERR_HIP, "ncclCommInitRank(rank %d of %d): %s", rank, nranks, ncclGetErrorString(r)); }
  if (hipStreamCreateWithFlags(&cm.streams[0], hipStreamNonBlocking) != hipSuccess) { comm_release(cm); return fail(GS_ERR_HIP, "stream for the communicator"); }
  return GS_OK;
}

int gs_comm_init_local(void) {
  std::vector<int> phys;
  {
    Registry& r = registry();
    std::lock_guard<std::mutex> rl(r.mu);
    if (r.ctxs.empty()) return fail(GS_ERR_NOT_INIT, "gs_comm_init_local: gs_init first");
    for (auto& c : r.ctxs) if (std::find(phys.begin(), phys.end(), c->device) == phys.end()) phys.push_back(c->device);
  }
  Comm& cm = comm();
  std::lock_guard<std::mutex> lk(cm.mu);
  if (cm.active()) return fail(GS_ERR_ARG, "a communicator already exists: gs_comm_destroy first");
  cm.local = true; cm.nranks = (int)phys.size(); cm.rank = 0; cm.phys = phys;
  cm.comms.assign(phys.size(), nullptr); cm.streams.assign(phys.size(), nullptr);
  cm.send.assign(phys.size(), nullptr); cm.recv.assign(phys.size(), nullptr);
  ncclResult_t rr = ncclCommInitAll(cm.comms.data(), (int)phys.size(), phys.data());
  if (rr != ncclSuccess) { for (auto& x : cm.comms) x = nullptr; comm_release(cm); return fail(GS_ERR_HIP, "ncclCommInitAll over %zu device(s): %s", phys.size(), ncclGetErrorString(rr)); }
  for (size_t i = 0; i < phys.size(); ++i) {
    if (hipSetDevice(phys[i]) != hipSuccess || hipStreamCreateWithFlags(&cm.streams[i], hipStreamNonBlocking) != hipSuccess) {
      comm_release(cm);
      return fail(GS_ERR_HIP, "stream for the communicator on device %d", phys[i]);
    }
  }
  return GS_OK;
}

void gs_comm_destroy(void) { multi_shutdown(); }

int gs_comm_info(int* nranks, int* rank, int* local, uint64_t* collectives) {
  Comm& cm = comm();
  std::lock_guard<std::mutex> lk(cm.mu);
  if (nranks) *nranks = cm.active() ? cm.nranks : 0;
  if (rank) *rank = cm.active() ? cm.rank : -1;
  if (local) *local = cm.local ? 1 : 0;
  if (collectives) *collectives = cm.collectives;
  return GS_OK;
}

int gs_comm_allgather(const void* send, size_t bytes, void* recv) {
  if (!send || !recv) return fail(GS_ERR_ARG, "gs_comm_allgather: null buffer");
  Comm& cm = comm();
  std::lock_guard<std::mutex> lk(cm.mu);
  return comm_allgather(cm, static_cast<const uint8_t*>(send), bytes, static_cast<uint8_t*>(recv));
}

// ---- one MSM over the logical devices of this process (configs[3]) ------------------------------------------------------
// bases[d] / scalars[d]: shard d of the term range (the same contiguous split on both), resident on one logical device.
static int msm_multi(bool g2, const gs_handle* bases, const gs_handle* scalars, int ndev, uint64_t* out_affine, int* is_inf, int* used_rccl) {
  std::vector<int> phys_of;
  if (int rc = check_devices(bases, scalars, nullptr, ndev, "gs_msm_multi", phys_of)) return rc;
  if (!scalars || !out_affine || !is_inf) return fail(GS_ERR_ARG, "gs_msm_multi: null argument");
  std::vector<size_t> len(ndev);
  for (int d = 0; d < ndev; ++d) {
    size_t nb = 0;
    if (int rc = gs_len(scalars[d], &len[d])) return rc;
    if (int rc = gs_len(bases[d], &nb)) return rc;
    if (nb != len[d]) return fail(GS_ERR_ARG, "gs_msm_multi: shard %d has %zu points but %zu scalars", d, nb, len[d]);
  }
  if (!g2) {
    std::vector<PointRecord<8>> mine(ndev), all;
    if (int rc = on_devices(ndev, [&](int d) {
          int inf = 0;
          const int rc = gs_msm_g1_resident(bases[d], 0, scalars[d], 0, len[d], mine[d].p, &inf);
          mine[d].inf = (uint32_t)inf; mine[d].shard = (uint32_t)d;
          return rc;
        }, "gs_msm_g1_multi")) return rc;
    if (int rc = exchange(mine, phys_of, all, used_rccl)) return rc;
    std::vector<uint64_t> pts((size_t)ndev * 8);
    std::vector<int> fl(ndev);
    for (int d = 0; d < ndev; ++d) { memcpy(&pts[(size_t)d * 8], all[d].p, 64); fl[d] = (int)all[d].inf; }
    return gs_g1_sum_affine(pts.data(), fl.data(), ndev, out_affine, is_inf);
  }
  std::vector<PointRecord<16>> mine(ndev), all;
  if (int rc = on_devices(ndev, [&](int d) {
        int inf = 0;
        const int rc = gs_msm_g2_resident(bases[d], 0, scalars[d], 0, len[d], mine[d].p, &inf);
        mine[d].inf = (uint32_t)inf; mine[d].shard = (uint32_t)d;
        return rc;
      }, "gs_msm_g2_multi")) return rc;
  if (int rc = exchange(mine, phys_of, all, used_rccl)) return rc;
  std::vector<uint64_t> pts((size_t)ndev * 16);
  std::vector<int> fl(ndev);
  for (int d = 0; d < ndev; ++d) { memcpy(&pts[(size_t)d * 16], all[d].p, 128); fl[d] = (int)all[d].inf; }
  return gs_g2_sum_affine(pts.data(), fl.data(), ndev, out_affine, is_inf);
}
int gs_msm_g1_multi(const gs_handle* bases, const gs_handle* scalars, int ndev, uint64_t out_affine[8], int* is_inf, int* used_rccl) {
  return msm_multi(false, bases, scalars, ndev, out_affine, is_inf, used_rccl);
}
int gs_msm_g2_multi(const gs_handle* bases, const gs_handle* scalars, int ndev, uint64_t out_affine[16], int* is_inf, int* used_rccl) {
  return msm_multi(true, bases, scalars, ndev, out_affine, is_inf, used_rccl);
}

// ---- one proof over the logical devices of this process -------------------------------------------------------------------
// pk[d]: the full key or slice d of ndev (gs_groth16_pk_shard_to), w[d] / px[d]: replicas of the witness and of P(x), all
// resident on one logical device per d.  Device d sums shard d of the term ranges (gs_groth16_prove_partials), the 416-byte
// records are exchanged, added, and the O(1) tail (groth16.go:253-275) runs once.  Same proof as gs_groth16_prove_resident.
int gs_groth16_prove_multi(const gs_handle* pk, const gs_handle* w, const gs_handle* px, int ndev, const uint64_t r[4], const uint64_t s[4],
                           uint64_t out_proof[32], int inf[3], int* used_rccl) {
  std::vector<int> phys_of;
  if (int rc = check_devices(pk, w, px, ndev, "gs_groth16_prove_multi", phys_of)) return rc;
  if (!w || !px || !r || !s || !out_proof || !inf) return fail(GS_ERR_ARG, "gs_groth16_prove_multi: null argument");
  std::vector<ProofRecord> mine(ndev), all;
  if (int rc = on_devices(ndev, [&](int d) {
        int f[5] = {0, 0, 0, 0, 0};
        const int rc = gs_groth16_prove_partials(pk[d], w[d], px[d], (size_t)d, (size_t)ndev, mine[d].sums, f);
        for (int k = 0; k < 5; ++k) mine[d].inf[k] = (uint32_t)f[k];
        mine[d].shard = (uint32_t)d; mine[d].pad[0] = mine[d].pad[1] = 0;
        return rc;
      }, "gs_groth16_prove_multi")) return rc;
  if (int rc = exchange(mine, phys_of, all, used_rccl)) return rc;
  uint64_t sums[48];
  int sinf[5];
  if (int rc = sum_proof_records(all, sums, sinf)) return rc;
  return gs_groth16_finish(pk[0], sums, sinf, r, s, out_proof, inf);
}

// ---- one process per GPU: this rank's shard, gathered over the communicator of gs_comm_init_rank ------------------------
int gs_groth16_prove_sharded(gs_handle pk, gs_handle w, gs_handle px, const uint64_t r[4], const uint64_t s[4], uint64_t out_proof[32], int inf[3]) {
  if (!r || !s || !out_proof || !inf) return fail(GS_ERR_ARG, "gs_groth16_prove_sharded: null argument");
  int nranks = 0, rank = -1, local = 0;
  gs_comm_info(&nranks, &rank, &local, nullptr);
  if (nranks < 1 || local) return fail(GS_ERR_ARG, "gs_groth16_prove_sharded: needs the communicator of gs_comm_init_rank (one process per GPU)");
  ProofRecord mine{};
  int f[5] = {0, 0, 0, 0, 0};
  if (int rc = gs_groth16_prove_partials(pk, w, px, (size_t)rank, (size_t)nranks, mine.sums, f)) return rc;
  for (int k = 0; k < 5; ++k) mine.inf[k] = (uint32_t)f[k];
  mine.shard = (uint32_t)rank;
  std::vector<ProofRecord> all(nranks);
  if (int rc = gs_comm_allgather(&mine, sizeof mine, all.data())) return rc;
  for (int i = 0; i < nranks; ++i)
    if (all[i].shard != (uint32_t)i) return fail(GS_ERR_HIP, "gathered record %d carries shard index %u", i, all[i].shard);
  uint64_t sums[48];
  int sinf[5];
  if (int rc = sum_proof_records(all, sums, sinf)) return rc;
  return gs_groth16_finish(pk, sums, sinf, r, s, out_proof, inf);
}

// ---- the same two, with the polynomial stage on ONE rank per proof (prove.hip: gs_groth16_witness_values) ------------------------
// hv[d]: device d's slice of H's values (the range of its key slice's evaluation-basis array), scattered by the proof's owner.
int gs_groth16_prove_multi_values(const gs_handle* pk, const gs_handle* w, const gs_handle* hv, int ndev, const uint64_t r[4], const uint64_t s[4],
                                  uint64_t out_proof[32], int inf[3], int* used_rccl) {
  std::vector<int> phys_of;
  if (int rc = check_devices(pk, w, hv, ndev, "gs_groth16_prove_multi_values", phys_of)) return rc;
  if (!w || !hv || !r || !s || !out_proof || !inf) return fail(GS_ERR_ARG, "gs_groth16_prove_multi_values: null argument");
  std::vector<ProofRecord> mine(ndev), all;
  if (int rc = on_devices(ndev, [&](int d) {
        int f[5] = {0, 0, 0, 0, 0};
        const int rc = gs_groth16_prove_partials_values(pk[d], w[d], hv[d], (size_t)d, (size_t)ndev, mine[d].sums, f);
        for (int k = 0; k < 5; ++k) mine[d].inf[k] = (uint32_t)f[k];
        mine[d].shard = (uint32_t)d; mine[d].pad[0] = mine[d].pad[1] = 0;
        return rc;
      }, "gs_groth16_prove_multi_values")) return rc;
  if (int rc = exchange(mine, phys_of, all, used_rccl)) return rc;
  uint64_t sums[48];
  int sinf[5];
  if (int rc = sum_proof_records(all, sums, sinf)) return rc;
  return gs_groth16_finish(pk[0], sums, sinf, r, s, out_proof, inf);
}

int gs_groth16_prove_sharded_values(gs_handle pk, gs_handle w, gs_handle hv_slice, const uint64_t r[4], const uint64_t s[4], uint64_t out_proof[32], int inf[3]) {
  if (!r || !s || !out_proof || !inf) return fail(GS_ERR_ARG, "gs_groth16_prove_sharded_values: null argument");
  int nranks = 0, rank = -1, local = 0;
  gs_comm_info(&nranks, &rank, &local, nullptr);
  if (nranks < 1 || local) return fail(GS_ERR_ARG, "gs_groth16_prove_sharded_values: needs the communicator of gs_comm_init_rank (one process per GPU)");
  ProofRecord mine{};
  int f[5] = {0, 0, 0, 0, 0};
  if (int rc = gs_groth16_prove_partials_values(pk, w, hv_slice, (size_t)rank, (size_t)nranks, mine.sums, f)) return rc;
  for (int k = 0; k < 5; ++k) mine.inf[k] = (uint32_t)f[k];
  mine.shard = (uint32_t)rank;
  std::vector<ProofRecord> all(nranks);
  if (int rc = gs_comm_allgather(&mine, sizeof mine, all.data())) return rc;
  for (int i = 0; i < nranks; ++i)
    if (all[i].shard != (uint32_t)i) return fail(GS_ERR_HIP, "gathered record %d carries shard index %u", i, all[i].shard);
  uint64_t sums[48];
  int sinf[5];
  if (int rc = sum_proof_records(all, sums, sinf)) return rc;
  return gs_groth16_finish(pk, sums, sinf, r, s, out_proof, inf);
}

// ---- Pinocchio (snark.go:254-289): the same four shapes; a proof is the sum of the ranks' eight partial points, there is no tail ----
// one process, ndev logical devices: pk[d] the full key or slice d (gs_pinocchio_pk_shard_to), w[d] / px[d] replicas
int gs_pinocchio_prove_multi(const gs_handle* pk, const gs_handle* w, const gs_handle* px, int ndev, uint64_t out_proof[72], int inf[8], int* used_rccl) {
  return pinocchio_multi(pk, w, px, ndev, out_proof, inf, used_rccl, "gs_pinocchio_prove_multi",
                         [&](int d, uint64_t* sums, int* f) { return gs_pinocchio_prove_partials(pk[d], w[d], px[d], (size_t)d, (size_t)ndev, sums, f); });
}
// ... hv[d]: device d's slice of H's values (gs_pinocchio_witness_values on the proof's owner, then the scatter)
int gs_pinocchio_prove_multi_values(const gs_handle* pk, const gs_handle* w, const gs_handle* hv, int ndev, uint64_t out_proof[72], int inf[8], int* used_rccl) {
  return pinocchio_multi(pk, w, hv, ndev, out_proof, inf, used_rccl, "gs_pinocchio_prove_multi_values",
                         [&](int d, uint64_t* sums, int* f) { return gs_pinocchio_prove_partials_values(pk[d], w[d], hv[d], (size_t)d, (size_t)ndev, sums, f); });
}
// one process per GPU: this rank's shard, the 616-byte records gathered over the communicator of gs_comm_init_rank
int gs_pinocchio_prove_sharded(gs_handle pk, gs_handle w, gs_handle px, uint64_t out_proof[72], int inf[8]) {
  return pinocchio_sharded(out_proof, inf, "gs_pinocchio_prove_sharded",
                           [&](size_t rank, size_t nranks, uint64_t* sums, int* f) { return gs_pinocchio_prove_partials(pk, w, px, rank, nranks, sums, f); });
}
int gs_pinocchio_prove_sharded_values(gs_handle pk, gs_handle w, gs_handle hv_slice, uint64_t out_proof[72], int inf[8]) {
  return pinocchio_sharded(out_proof, inf, "gs_pinocchio_prove_sharded_values",
                           [&](size_t rank, size_t nranks, uint64_t* sums, int* f) { return gs_pinocchio_prove_partials_values(pk, w, hv_slice, rank, nranks, sums, f); });
}

// The owner's scatter between processes: rank `root` holds `total` scalars (handle `full`, ignored on the other ranks); every rank
// -- the root included -- ends up with its slice of the contiguous split of [0, total) (first ranks one longer: the split every
// sharded entry point uses) in *slice_inout (0 = create).  ncclSend / ncclRecv in one group on the communicator's stream: the
// root's 32 (n / N) bytes per peer leave over its xGMI links in parallel (n = 2^22 over 8 ranks = 16 MiB per peer).
int gs_scalars_scatter(gs_handle full, size_t total, int root, gs_handle* slice_inout) {
  if (!slice_inout) return fail(GS_ERR_ARG, "gs_scalars_scatter: null output");
  Comm& cm = comm();
  std::lock_guard<std::mutex> lk(cm.mu);
  if (!cm.active() || cm.local) return fail(GS_ERR_ARG, "gs_scalars_scatter: needs the communicator of gs_comm_init_rank (one process per GPU)");
  if (root < 0 || root >= cm.nranks) return fail(GS_ERR_ARG, "gs_scalars_scatter: root %d of %d ranks", root, cm.nranks);
  auto range = [&](int k, size_t& lo, size_t& hi) {
    const size_t q = total / (size_t)cm.nranks, rem = total % (size_t)cm.nranks;
    lo = (size_t)k * q + std::min<size_t>((size_t)k, rem);
    hi = lo + q + ((size_t)k < rem ? 1 : 0);
  };
  size_t mylo, myhi;
  range(cm.rank, mylo, myhi);
  const size_t mine = myhi - mylo;
  uint32_t* dst = nullptr;
  const uint32_t* src = nullptr;
  // resolve the handles on their context (objects are kept alive by the caller for the duration of the call)
  int rc = guarded([&](Ctx& c) -> int {
    Scalars* out = nullptr;
    if (*slice_inout) {
      out = c.get<Scalars>(*slice_inout, Kind::Scalars);
      if (!out || out->n != mine) return fail(GS_ERR_ARG, "gs_scalars_scatter: the output handle does not hold this rank's %zu values", mine);
    } else {
      auto fresh = std::make_unique<Scalars>();
      fresh->n = mine;
      fresh->buf.alloc(std::max<size_t>(mine, 1) * 32);
      out = fresh.get();
      *slice_inout = c.put(std::move(fresh));
    }
    dst = out->buf.as<uint32_t>();
    if (cm.rank == root) {
      Scalars* f = c.get<Scalars>(full, Kind::Scalars);
      if (!f || f->n != total) return fail(GS_ERR_ARG, "gs_scalars_scatter: the root's vector must hold the %zu values", total);
      src = f->buf.as<uint32_t>();
    }
    c.drain();                                          // whatever produced the root's vector / still reads the old slice
    return GS_OK;
  }, true, true, *slice_inout ? *slice_inout : (cm.rank == root ? full : 0));
  if (rc != GS_OK) return rc;
  GS_HIPRC(hipSetDevice(cm.phys[0]));
  GS_NCCL(ncclGroupStart());
  if (cm.rank == root) {
    for (int k = 0; k < cm.nranks; ++k) {
      size_t lo, hi;
      range(k, lo, hi);
      if (k == root || hi == lo) continue;
      GS_NCCL(ncclSend(src + lo * 8, (hi - lo) * 32, ncclUint8, k, cm.comms[0], cm.streams[0]));
    }
  } else if (mine) {
    GS_NCCL(ncclRecv(dst, mine * 32, ncclUint8, root, cm.comms[0], cm.streams[0]));
  }
  GS_NCCL(ncclGroupEnd());
  if (cm.rank == root && mine) GS_HIPRC(hipMemcpyAsync(dst, src + mylo * 8, mine * 32, hipMemcpyDeviceToDevice, cm.streams[0]));
  GS_HIPRC(hipStreamSynchronize(cm.streams[0]));
  cm.collectives += 1;
  return GS_OK;
}

static int msm_sharded(bool g2, gs_handle bases, gs_handle scalars, uint64_t* out_affine, int* is_inf) {
  if (!out_affine || !is_inf) return fail(GS_ERR_ARG, "gs_msm_sharded: null output");
  int nranks = 0, rank = -1, local = 0;
  gs_comm_info(&nranks, &rank, &local, nullptr);
  if (nranks < 1 || local) return fail(GS_ERR_ARG, "gs_msm_sharded: needs the communicator of gs_comm_init_rank (one process per GPU)");
  size_t n = 0, nb = 0;
  if (int rc = gs_len(scalars, &n)) return rc;
  if (int rc = gs_len(bases, &nb)) return rc;
  if (n != nb) return fail(GS_ERR_ARG, "gs_msm_sharded: %zu points but %zu scalars in this rank's shard", nb, n);
  const size_t words = g2 ? 16 : 8, rec = words * 8 + 8;
  std::vector<uint8_t> mine(rec), all(rec * (size_t)nranks);
  int inf = 0;
  const int rc = g2 ? gs_msm_g2_resident(bases, 0, scalars, 0, n, reinterpret_cast<uint64_t*>(mine.data()), &inf)
                    : gs_msm_g1_resident(bases, 0, scalars, 0, n, reinterpret_cast<uint64_t*>(mine.data()), &inf);
  if (rc) return rc;
  const uint32_t tag[2] = {(uint32_t)inf, (uint32_t)rank};
  memcpy(mine.data() + words * 8, tag, 8);
  if (int rc2 = gs_comm_allgather(mine.data(), rec, all.data())) return rc2;
  std::vector<uint64_t> pts((size_t)nranks * words);
  std::vector<int> fl(nranks);
  for (int i = 0; i < nranks; ++i) {
    memcpy(&pts[(size_t)i * words], all.data() + (size_t)i * rec, words * 8);
    uint32_t t2[2];
    memcpy(t2, all.data() + (size_t)i * rec + words * 8, 8);
    if (t2[1] != (uint32_t)i) return fail(GS_ERR_HIP, "gathered record %d carries rank %u", i, t2[1]);
    fl[i] = (int)t2[0];
  }
  return g2 ? gs_g2_sum_affine(pts.data(), fl.data(), nranks, out_affine, is_inf) : gs_g1_sum_affine(pts.data(), fl.data(), nranks, out_affine, is_inf);
}
int gs_msm_g1_sharded(gs_handle bases, gs_handle scalars, uint64_t out_affine[8], int* is_inf) { return msm_sharded(false, bases, scalars, out_affine, is_inf); }
int gs_msm_g2_sharded(gs_handle bases, gs_handle scalars, uint64_t out_affine[16], int* is_inf) { return msm_sharded(true, bases, scalars, out_affine, is_inf); }

// ---- a batch of independent proofs, one proof per device at a time (configs[4]) -----------------------------------------------
// Proof i reads w[i] / px[i] and runs on the logical device those handles live on, with the key pk_of_device[that device]
// (one resident key per logical device; pass 0 for devices the batch does not use).  Every device streams its proofs
// through the pipelined prover (three in flight); there is no collective.  r, s: nproofs x 4 words.
int gs_groth16_prove_batch(const gs_handle* pk_of_device, int ndev, const gs_handle* w, const gs_handle* px, size_t nproofs,
                           const uint64_t* r, const uint64_t* s, uint64_t* out_proofs /* nproofs x 32 */, int* inf /* nproofs x 3 */) {
  if (!pk_of_device || ndev < 1 || ndev > kMaxLogicalDevices || (nproofs && (!w || !px || !r || !s || !out_proofs || !inf)))
    return fail(GS_ERR_ARG, "gs_groth16_prove_batch: null argument");
  std::vector<std::vector<size_t>> work(ndev);
  for (size_t i = 0; i < nproofs; ++i) {
    const int ld = handle_device(w[i]);
    if (ld >= ndev || handle_device(px[i]) != ld || !pk_of_device[ld] || handle_device(pk_of_device[ld]) != ld)
      return fail(GS_ERR_ARG, "gs_groth16_prove_batch: proof %zu: w / px / key do not share one logical device below %d", i, ndev);
    work[ld].push_back(i);
  }
  return on_devices(ndev, [&](int d) -> int {
    const std::vector<size_t>& q = work[d];
    std::vector<uint64_t> tickets(q.size(), 0);
    const size_t depth = Ctx::kMaxInFlight;
    size_t begun = 0, collected = 0;
    int rc = GS_OK;
    for (size_t k = 0; k < q.size() + depth && rc == GS_OK; ++k) {
      if (k >= depth && collected < begun) {
        const size_t i = q[collected];
        rc = gs_groth16_prove_end(tickets[collected], out_proofs + i * 32, inf + i * 3);
        ++collected;                                  // collected or not, that ticket is gone (prove_end consumed the slot)
        if (rc != GS_OK) break;
      }
      if (k < q.size()) {
        const size_t i = q[k];
        rc = gs_groth16_prove_begin(pk_of_device[d], w[i], px[i], r + i * 4, s + i * 4, &tickets[k]);
        if (rc == GS_OK) ++begun;
      }
    }
    if (rc != GS_OK) {
      // Leave no ticket behind: an uncollected ticket keeps its slot, its key and its vectors, and every later
      // gs_*_begin on this device would answer GS_ERR_BUSY until gs_shutdown.  The error that is reported is the first one.
      const std::string first = gs_last_error();
      for (size_t t = collected; t < begun; ++t) (void)gs_ticket_cancel(tickets[t]);
      last_error_ref() = first;
    }
    return rc;
  }, "gs_groth16_prove_batch");
}

// snark.GenerateProofs for a batch of independent witnesses (as gs_groth16_prove_batch; the proofs are deterministic: no r, s)
int gs_pinocchio_prove_batch(const gs_handle* pk_of_device, int ndev, const gs_handle* w, const gs_handle* px, size_t nproofs,
                             uint64_t* out_proofs /* nproofs x 72 */, int* inf /* nproofs x 8 */) {
  if (!pk_of_device || ndev < 1 || ndev > kMaxLogicalDevices || (nproofs && (!w || !px || !out_proofs || !inf)))
    return fail(GS_ERR_ARG, "gs_pinocchio_prove_batch: null argument");
  std::vector<std::vector<size_t>> work(ndev);
  for (size_t i = 0; i < nproofs; ++i) {
    const int ld = handle_device(w[i]);
    if (ld >= ndev || handle_device(px[i]) != ld || !pk_of_device[ld] || handle_device(pk_of_device[ld]) != ld)
      return fail(GS_ERR_ARG, "gs_pinocchio_prove_batch: proof %zu: w / px / key do not share one logical device below %d", i, ndev);
    work[ld].push_back(i);
  }
  return on_devices(ndev, [&](int d) -> int {
    const std::vector<size_t>& q = work[d];
    std::vector<uint64_t> tickets(q.size(), 0);
    const size_t depth = Ctx::kMaxInFlight;
    size_t begun = 0, collected = 0;
    int rc = GS_OK;
    for (size_t k = 0; k < q.size() + depth && rc == GS_OK; ++k) {
      if (k >= depth && collected < begun) {
        const size_t i = q[collected];
        rc = gs_pinocchio_prove_end(tickets[collected], out_proofs + i * 72, inf + i * 8);
        ++collected;
        if (rc != GS_OK) break;
      }
      if (k < q.size()) {
        const size_t i = q[k];
        rc = gs_pinocchio_prove_begin(pk_of_device[d], w[i], px[i], &tickets[k]);
        if (rc == GS_OK) ++begun;
      }
    }
    if (rc != GS_OK) {                                  // leave no ticket behind (as gs_groth16_prove_batch)
      const std::string first = gs_last_error();
      for (size_t t = collected; t < begun; ++t) (void)gs_ticket_cancel(tickets[t]);
      last_error_ref() = first;
    }
    return rc;
  }, "gs_pinocchio_prove_batch");
}

}  // extern "C"
