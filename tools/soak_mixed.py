"""Mixed soak (dev tool, round 4): RANDOM interleavings of every pipelined operation on ONE context, up to three tickets in flight,
every result compared with the answer of a blocking call made before the soak started.

What it is for: round 4 moved work between streams by rule (plan(w) on its own stream for witness-route / large proofs, MSM tickets
above 3 * 2^20 terms on their slot's stream, tail kernels in two forms by size, tail streams swapped per operation).  Each rule was
tested with runs of the SAME operation; this mixes them -- a Groth16 ticket from px behind a witness-route ticket behind a 2^22-term MSM
ticket behind a Pinocchio ticket ... -- so that a buffer or a stream handed from one kind of operation to another too early shows up
as a wrong proof.

Round 5 adds what that round added: host-buffer tickets (a witness / w + px from host memory into the slot's own buffers), in-place
gs_scalars_update under outstanding tickets, and -- between operations, at random -- the table policy (auto / always / never),
gs_release_tables and gs_build_tables of random keys and base arrays, so that operations switch between the window-table and the
table-free route while background table builds are in flight and tables come and go under queued tickets.
Usage: python tools/soak_mixed.py [seconds] [seed]"""
import os
import random
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gosnark_amd  # noqa: F401,E402
from gosnark_amd import capi, groth16, r1csqap, snark, synth  # noqa: E402

seconds = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
rng = random.Random(seed)
capi.init()
t_setup = time.perf_counter()

ops = []          # (name, begin() -> ticket, end(ticket) -> result, expected)
handles = []      # keys and base arrays whose window tables the soak builds / releases at random


def same_proof(p, q):
    return (p.PiA, p.PiB, p.PiC) == (q.PiA, q.PiB, q.PiC)


def same_pinocchio(p, q):
    return all(getattr(p, f) == getattr(q, f) for f in ("PiA", "PiAp", "PiB", "PiBp", "PiC", "PiCp", "PiH", "PiKp"))


for log2n, kind in ((16, "sqchain"), (18, "sqchain"), (18, "realistic"), (20, "sqchain")):
    inst = synth.sqchain_setup_instance(1 << log2n, 100 + log2n) if kind == "sqchain" else synth.realistic_setup_instance(1 << log2n, 200 + log2n)
    pk = inst.device_pk()
    dr = r1csqap.DeviceR1CS(*inst.r1cs, inst.m)
    r_, s_ = synth.field_elems(2, 7 + log2n)
    want = groth16.prove_resident(pk, inst.w, inst.px, r_, s_)
    want_w = groth16.prove_from_witness(pk, dr, inst.w, r_, s_)
    assert same_proof(want, want_w), "px route and witness route disagree at 2^%d (%s)" % (log2n, kind)
    ops.append(("groth16 px 2^%d %s" % (log2n, kind), (lambda pk=pk, i=inst, r=r_, s=s_: groth16.prove_begin(pk, i.w, i.px, r, s)), groth16.prove_end, want, same_proof))
    ops.append(("groth16 witness 2^%d %s" % (log2n, kind), (lambda pk=pk, dr=dr, i=inst, r=r_, s=s_: groth16.prove_witness_begin(pk, dr, i.w, r, s)), groth16.prove_end, want, same_proof))
    ops.append(("groth16 host witness 2^%d %s" % (log2n, kind), (lambda pk=pk, dr=dr, i=inst, r=r_, s=s_: groth16.prove_witness_host_begin(pk, dr, i.w_host, r, s)), groth16.prove_end, want, same_proof))
    ops.append(("groth16 host w+px 2^%d %s" % (log2n, kind), (lambda pk=pk, i=inst, r=r_, s=s_: groth16.prove_host_begin(pk, i.w_host, i.px_host, r, s)), groth16.prove_end, want, same_proof))
    if log2n <= 18:      # in-place update of a private copy of the witness right behind the ticket that reads it (same values: the proof must not change)
        wcopy = capi.scalars_upload(inst.w_host)

        def upd(pk=pk, dr=dr, i=inst, r=r_, s=s_, h=wcopy):
            t = groth16.prove_witness_begin(pk, dr, h, r, s)
            capi.scalars_update(h, i.w_host)
            return t
        ops.append(("groth16 witness + update 2^%d %s" % (log2n, kind), upd, groth16.prove_end, want, same_proof))
    handles.append(pk.handle)

pin = synth.sqchain_pinocchio_instance(1 << 16, 12)
ppk = pin.device_pk()
want_p = snark.prove_resident(ppk, pin.w, pin.px)
ops.append(("pinocchio 2^16", (lambda: snark.prove_begin(ppk, pin.w, pin.px)), snark.prove_end, want_p, same_pinocchio))
pdr = r1csqap.DeviceR1CS(*pin.r1cs, pin.m)
ops.append(("pinocchio host witness 2^16", (lambda: snark.prove_witness_host_begin(ppk, pdr, pin.w_host)), snark.prove_end, want_p, same_pinocchio))
handles.append(ppk.handle)

NB = 1 << 22
g1 = capi.g1_fixed_base(synth.scalars_u64(NB, 31))
sc = capi.scalars_upload(synth.scalars_u64(NB, 32))
g2 = capi.g2_fixed_base(synth.scalars_u64(1 << 18, 33))
for n, off in ((1 << 12, 5), (1 << 16, 1000), (1 << 18, 77), (1 << 20, 3), ((3 << 20) - 1, 0), (3 << 20, 1), (NB, 0)):
    want_m = capi.msm_resident(g1, sc, n, off, off)
    ops.append(("msm g1 %d terms" % n, (lambda n=n, off=off: capi.msm_begin(g1, sc, n, off, off)), capi.msm_end, want_m, lambda x, y: x == y))
for n in (1 << 14, 1 << 18):
    want_m = capi.msm_resident(g2, sc, n, 0, 9, g2=True)
    ops.append(("msm g2 %d terms" % n, (lambda n=n: capi.msm_begin(g2, sc, n, 0, 9, g2=True)), capi.msm_end, want_m, lambda x, y: x == y))
handles += [g1, g2]
print("setup %.1f s, %d kinds of operation" % (time.perf_counter() - t_setup, len(ops)), flush=True)
policy_events = {"auto": 0, "always": 0, "never": 0, "release": 0, "build": 0}

counts = {name: 0 for name, *_ in ops}
inflight = []
t0 = time.perf_counter()
done = 0
while time.perf_counter() - t0 < seconds or inflight:
    more = time.perf_counter() - t0 < seconds
    # keep a random number (1..3) of tickets outstanding; sometimes drain completely, sometimes repeat one kind
    target = rng.choice((1, 2, 3, 3, 3)) if more else 0
    if more and rng.random() < 0.08:             # tables come and go, the policy changes -- with whatever is in flight
        ev = rng.choice(("auto", "auto", "always", "never", "release", "release", "build"))
        if ev in ("auto", "always", "never"):
            capi.set_table_policy(ev)
        elif ev == "release":
            capi.release_tables(rng.choice(handles))
        else:
            h = rng.choice(handles[:-2])         # (a key: building the 2^22-point array's table costs 4 GiB and a second)
            capi.build_tables(h, rng.choice((0, 1, 2)))
        policy_events[ev] += 1
    while more and len(inflight) < target:
        op = rng.choice(ops)
        for _ in range(rng.choice((1, 1, 2, 3))):
            if len(inflight) < 3:
                inflight.append((op, op[1]()))
    if inflight:
        op, ticket = inflight.pop(0)
        got = op[2](ticket)
        if not op[4](got, op[3]):
            print("MISMATCH after %d operations: %s (seed %d); in flight behind it: %s" % (done, op[0], seed, [o[0] for o, _ in inflight]), flush=True)
            sys.exit(1)
        counts[op[0]] += 1
        done += 1
        if done % 500 == 0:
            print("%6d operations, %.0f s" % (done, time.perf_counter() - t0), flush=True)
print("OK: %d operations in %.0f s, every result equal to its blocking twin" % (done, time.perf_counter() - t0))
for k, v in counts.items():
    print("   %-40s %d" % (k, v))
print("   table events:", policy_events, "| evictions:", capi.memory_query()["evictions"])
