// Memory accounting and eviction (include/gosnark_hip.h, "device memory").  The window tables are the library's one large,
// rebuildable cost (15 rows per base array at c = 17: 5.6 GiB per 2^20 Groth16 key against 0.4 GiB of key data), so a host
// that keeps several keys resident needs to see them and to be able to drop them without dropping the key.
#include "msm.h"
#include "poly.h"
#include "prove.h"
#include "runtime.h"

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
  for_each_table(o, [&](BaseTable& b) { t += b.rows.bytes; });
  return t;
}

}  // namespace

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
    for_each_table(it->second.get(), [&](BaseTable& b) { b.rows.release(); b.n = 0; b.c = 0; b.W = 0; });
    return GS_OK;
  }, true, false, h);
}

// Drop every cached workspace of the current logical device (bucket sets, plan buffers, NTT twiddles, node trees, factorial
// tables, fixed-base tables): all of it is rebuilt on demand.  Keys, base arrays, scalars and their tables stay.
int gs_trim(void) {
  return guarded([&](Ctx& c) -> int {
    c.drain();
    for (int i = 0; i < Ctx::kWsSets; ++i) { c.ws_buckets[i].release(); c.ws_chunks[i].release(); c.ws_partials[i].release(); c.ws_out[i].release(); }
    c.ws_misc.release(); c.g1_pow2.release(); c.g2_pow2.release();
    c.msm_state.reset(); c.poly_state.reset(); c.prove_state.reset();
    return GS_OK;
  }, true, false);
}

}  // extern "C"
