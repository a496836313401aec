// Memory accounting and eviction (include/gosnark_hip.h, "device memory").  The window tables are the library's one large,
// rebuildable cost (15 rows per base array at c = 17: 5.6 GiB per 2^20 Groth16 key against 0.4 GiB of key data), so a host
// that keeps several keys resident needs to see them and to be able to drop them without dropping the key.
#include "msm.h"
#include "poly.h"
#include "prove.h"
#include "runtime.h"

#include <algorithm>

using namespace gs;

namespace {

uint64_t divisor_bytes(const Divisor& d) { return d.b_std.bytes + d.inv_rev_mont.bytes + d.inv_spec.bytes; }

template <class F>
void for_each_table(Object* o, F f) {
  switch (o->kind) {
    case Kind::G1Bases: case Kind::G2Bases: {
      auto* b = static_cast<Bases*>(o);
      if (b->table) f(*static_cast<BaseTable*>(b->table.get()));
      break;
    }
    case Kind::GrothPk: {
      auto* k = static_cast<GrothPkObj*>(o);
      for (BaseTable* t : {&k->t_at, &k->t_bacgamma1, &k->t_bacdelta, &k->t_ptd, &k->t_bacgamma2, &k->t_ptd_eval}) f(*t);
      break;
    }
    case Kind::PinocchioPk: {
      auto* k = static_cast<PinocchioPkObj*>(o);
      for (BaseTable* t : {&k->t_a, &k->t_ap, &k->t_bp, &k->t_c, &k->t_cp, &k->t_kp, &k->t_g1t, &k->t_b2, &k->t_g1t_eval}) f(*t);
      break;
    }
    default: break;
  }
}

uint64_t object_bytes(Object* o) {
  switch (o->kind) {
    case Kind::G1Bases: case Kind::G2Bases: return static_cast<Bases*>(o)->buf.bytes;
    case Kind::Scalars: return static_cast<Scalars*>(o)->buf.bytes;
    case Kind::GrothPk: {
      auto* k = static_cast<GrothPkObj*>(o);
      return k->at.bytes + k->bacgamma1.bytes + k->bacdelta.bytes + k->ptd.bytes + k->bacgamma2.bytes + k->ptd_eval.bytes + divisor_bytes(k->z);
    }
    case Kind::PinocchioPk: {
      auto* k = static_cast<PinocchioPkObj*>(o);
      return k->a.bytes + k->ap.bytes + k->bp.bytes + k->c.bytes + k->cp.bytes + k->kp.bytes + k->g1t.bytes + k->b2.bytes + k->g1t_eval.bytes + divisor_bytes(k->z);
    }
    case Kind::R1cs: {
      auto* r = static_cast<R1csObj*>(o);
      uint64_t t = r->w_mont.bytes + r->vals.bytes + r->coef.bytes + r->prod.bytes;
      for (int i = 0; i < 3; ++i) t += r->rowptr[i].bytes + r->col[i].bytes + r->val[i].bytes;
      return t;
    }
  }
  return 0;
}

uint64_t table_bytes(Object* o) {
  uint64_t t = 0;
  for_each_table(o, [&](BaseTable& b) { t += b.rows.bytes + b.pending.bytes; });
  return t;
}

}  // namespace

// Out of device memory (runtime.h, dev_malloc): drop window tables of the calling thread's context, least recently used first,
// until `need` bytes are free.  Never a table of an object that an outstanding ticket holds (its kernels may still read it), never
// one the running call stamped (prepare_tables stamps every table of the call before anything is built), never one that is being
// built.  Everything else is idle: blocking calls have finished with the device when they return.  A dropped table is rebuilt -- or
// its array summed table-free -- by the handle's next use (gs_set_table_policy).
bool gs::evict_tables_for(size_t need) {
  Ctx* c = current_ctx();
  if (!c) return false;
  struct Cand { BaseTable* t; uint64_t stamp; };
  std::vector<Cand> cands;
  for (auto& kv : c->objs) {
    Object* o = kv.second.get();
    bool held = false;
    for (auto& f : c->inflight)
      if (f) for (auto& k : f->keep) held = held || k.get() == o;
    if (held) continue;
    for_each_table(o, [&](BaseTable& b) {
      if (b.rows.p && !b.pending.p && b.last_use != c->call_clock) cands.push_back(Cand{&b, b.last_use});
    });
  }
  std::sort(cands.begin(), cands.end(), [](const Cand& a, const Cand& b) { return a.stamp < b.stamp; });
  size_t freed = 0;
  for (auto& cd : cands) {
    if (freed >= need) break;
    freed += cd.t->rows.bytes;
    cd.t->drop();
    cd.t->uses = 0;
    c->evictions += 1;
  }
  return freed > 0;
}

extern "C" {

int gs_memory_query(gs_memory* out) {
  return guarded([&](Ctx& c) -> int {
    if (!out) return fail(GS_ERR_ARG, "gs_memory_query: null output");
    *out = gs_memory{};
    size_t free_b = 0, total_b = 0;
    GS_HIP(hipMemGetInfo(&free_b, &total_b));
    out->device_total_bytes = total_b;
    out->device_free_bytes = free_b;
    out->library_bytes = devbuf_bytes().load();
    for (auto& kv : c.objs) {
      out->object_bytes += object_bytes(kv.second.get());
      out->table_bytes += table_bytes(kv.second.get());
      out->objects += 1;
    }
    for (int i = 0; i < Ctx::kWsSets; ++i)
      out->workspace_bytes += c.ws_buckets[i].bytes + c.ws_chunks[i].bytes + c.ws_partials[i].bytes + c.ws_out[i].bytes;
    out->workspace_bytes += c.ws_misc.bytes + c.g1_pow2.bytes + c.g2_pow2.bytes;
    out->evictions = c.evictions;
    return GS_OK;
  }, true, true);
}

int gs_handle_bytes(gs_handle h, uint64_t* object_b, uint64_t* table_b) {
  return guarded([&](Ctx& c) -> int {
    auto it = c.objs.find(h);
    if (it == c.objs.end()) return fail(GS_ERR_ARG, "gs_handle_bytes: bad handle");
    if (object_b) *object_b = object_bytes(it->second.get());
    if (table_b) *table_b = table_bytes(it->second.get());
    return GS_OK;
  }, true, true, h);
}

// Queues behind outstanding tickets (they read the tables), then frees; the next proof / MSM on the handle rebuilds them.
int gs_release_tables(gs_handle h) {
  return guarded([&](Ctx& c) -> int {
    auto it = c.objs.find(h);
    if (it == c.objs.end()) return fail(GS_ERR_ARG, "gs_release_tables: bad handle");
    c.drain();
    for_each_table(it->second.get(), [&](BaseTable& b) { table_settle(c, b, false); b.drop(); b.uses = 0; });
    return GS_OK;
  }, true, false, h);
}

// When base arrays get their window tables: 0 = auto (default: table-free until an array's second use, then a build in the
// background), 1 = always (inside the first call that needs them: ~140 ms per 2^20 Groth16 key), 2 = never.  Every logical device.
int gs_set_table_policy(int policy) {
  if (policy < 0 || policy > 2) return fail(GS_ERR_ARG, "gs_set_table_policy: 0 (auto), 1 (always) or 2 (never)");
  Registry& r = registry();
  std::lock_guard<std::mutex> rl(r.mu);
  if (r.ctxs.empty()) return fail(GS_ERR_NOT_INIT, "gs_init has not been called (or failed)");
  for (auto& pc : r.ctxs) {
    std::lock_guard<FairMutex> lk(pc->mu);
    pc->table_policy = policy;
  }
  return GS_OK;
}

// Build the window tables of a key or base array NOW (a server that loads a key it will prove with for hours; bench.py's steady
// state): blocking, whatever the policy.  route (keys only): 0 = everything the key can use, 1 = the arrays of the px routes
// (PowersTauDelta / G1T), 2 = those of the witness routes (the evaluation-basis array when the key has one).  The widths are the
// ones a full-range proof / MSM picks; a later call with another width (gs_set_window_bits, a shard of a full key) rebuilds or goes
// table-free as the policy says.
int gs_build_tables(gs_handle h, int route) {
  return guarded([&](Ctx& c) -> int {
    auto it = c.objs.find(h);
    if (it == c.objs.end()) return fail(GS_ERR_ARG, "gs_build_tables: bad handle");
    if (route < 0 || route > 2) return fail(GS_ERR_ARG, "gs_build_tables: route 0, 1 or 2");
    Object* o = it->second.get();
    auto build = [&](BaseTable& t, const DevBuf& pts, size_t n, bool g2) {
      if (!n) return;
      t.last_use = c.call_clock;
      const int cb = choose_window_bits((uint32_t)n, c.window_bits);
      if (t.pending.p) table_settle(c, t, t.pending_c == cb && t.pending_n == n);
      if (g2) ensure_table_g2(c, t, pts.as<uint32_t>(), n, cb); else ensure_table_g1(c, t, pts.as<uint32_t>(), n, cb);
    };
    switch (o->kind) {
      case Kind::G1Bases: case Kind::G2Bases: {
        auto* b = static_cast<Bases*>(o);
        if (!b->table) b->table = std::make_shared<BaseTable>();
        build(*static_cast<BaseTable*>(b->table.get()), b->buf, b->n, o->kind == Kind::G2Bases);
        return GS_OK;
      }
      case Kind::GrothPk: {
        auto* k = static_cast<GrothPkObj*>(o);
        for_each_table(o, [&](BaseTable& t) { t.last_use = c.call_clock; });      // none of them is this call's eviction victim
        build(k->t_at, k->at, k->n_w, false); build(k->t_bacgamma1, k->bacgamma1, k->n_w, false); build(k->t_bacdelta, k->bacdelta, k->n_w, false);
        build(k->t_bacgamma2, k->bacgamma2, k->n_w, true);
        if (route != 2) build(k->t_ptd, k->ptd, k->n_h, false);
        if (route != 1 && k->n_e) build(k->t_ptd_eval, k->ptd_eval, k->n_e, false);
        return GS_OK;
      }
      case Kind::PinocchioPk: {
        auto* k = static_cast<PinocchioPkObj*>(o);
        for_each_table(o, [&](BaseTable& t) { t.last_use = c.call_clock; });
        build(k->t_a, k->a, k->n_w, false); build(k->t_ap, k->ap, k->n_w, false); build(k->t_bp, k->bp, k->n_w, false); build(k->t_c, k->c, k->n_w, false);
        build(k->t_cp, k->cp, k->n_w, false); build(k->t_kp, k->kp, k->n_w, false); build(k->t_b2, k->b2, k->n_w, true);
        if (route != 2) build(k->t_g1t, k->g1t, k->n_h, false);
        if (route != 1 && k->n_e) build(k->t_g1t_eval, k->g1t_eval, k->n_e, false);
        return GS_OK;
      }
      default: return fail(GS_ERR_ARG, "gs_build_tables: the handle has no base arrays");
    }
  }, true, false, h);
}

// Development / test hook: cap the device bytes this library may hold (0 = no cap).  An allocation beyond it behaves exactly like
// hipErrorOutOfMemory: least-recently-used window tables are evicted, one retry, then GS_ERR_HIP.
int gs_set_memory_limit(uint64_t bytes) {
  devbuf_limit().store(bytes);
  return GS_OK;
}

// sizeof the two structs this library writes through caller pointers (a binding compiled against another header revision can tell)
int gs_abi_sizes(size_t* timing_bytes, size_t* memory_bytes) {
  if (timing_bytes) *timing_bytes = sizeof(gs_timing);
  if (memory_bytes) *memory_bytes = sizeof(gs_memory);
  return GS_OK;
}

// Drop every cached workspace of the current logical device (bucket sets, plan buffers, NTT twiddles, node trees, factorial
// tables, fixed-base tables): all of it is rebuilt on demand.  Keys, base arrays, scalars and their tables stay.
int gs_trim(void) {
  return guarded([&](Ctx& c) -> int {
    c.drain();
    if (c.table_stream) GS_HIP(hipStreamSynchronize(c.table_stream));      // a background table build uses the engine's scratch slab
    for (int i = 0; i < Ctx::kWsSets; ++i) { c.ws_buckets[i].release(); c.ws_chunks[i].release(); c.ws_partials[i].release(); c.ws_out[i].release(); }
    c.ws_misc.release(); c.g1_pow2.release(); c.g2_pow2.release();
    c.msm_state.reset(); c.poly_state.reset(); c.prove_state.reset();
    return GS_OK;
  }, true, false);
}

}  // extern "C"
