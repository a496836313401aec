"""Time from a fresh resident key to its first proofs -- dev tool.
    python tools/time_first_proof.py [policy=auto] [log2n=20] [proofs=40] [log2n of the warm-up key=10]
Prints every blocking proof's wall time and the window width it ran on (gs_timing.window_bits: the table-free route's differs from
the table route's), i.e. the whole warm-up transient of a key under the table policy: first proof, the proofs that share the chip
with the background builds (GS_TABLE_BG_SLAB_LOG2), the switch-over."""
import sys, time
sys.path.insert(0, "/root/repo")
import torch
import gosnark_amd
from gosnark_amd import capi, synth, groth16
policy = sys.argv[1] if len(sys.argv) > 1 else "auto"
logn = int(sys.argv[2]) if len(sys.argv) > 2 else 20
count = int(sys.argv[3]) if len(sys.argv) > 3 else 40
capi.init()
capi.set_table_policy(policy)
warm_logn = int(sys.argv[4]) if len(sys.argv) > 4 else 10      # 20: the process has proven at this size before (workspaces exist): what is left is per KEY
warm = synth.sqchain_setup_instance(1 << warm_logn, 1)
groth16.prove_resident(warm.device_pk(), warm.w, warm.px, *synth.field_elems(2, 4))
inst = synth.sqchain_setup_instance(1 << logn, 3)
r, s = synth.field_elems(2, 5)
pk = inst.device_pk()
torch.cuda.synchronize()
t0 = time.perf_counter()
rows, first = [], None
for i in range(count):
    t = time.perf_counter()
    p = groth16.prove_resident(pk, inst.w, inst.px, r, s)
    dt = (time.perf_counter() - t) * 1e3
    first = first or p
    assert (p.PiA, p.PiB, p.PiC) == (first.PiA, first.PiB, first.PiC)
    rows.append((dt, capi.last_timing()["window_bits"], capi.handle_bytes(pk.handle)[1] >> 20))
print("policy %s, 2^%d: proof ms (window bits, table MiB):" % (policy, logn), " ".join("%.1f(%d,%d)" % x for x in rows))
switch = next((i for i, x in enumerate(rows) if x[1] != rows[0][1]), None)
print("  first %.1f ms, second %.1f ms, sum until the tables serve %.1f ms (proof #%s), steady %.2f ms" % (
    rows[0][0], rows[1][0], sum(x[0] for x in rows[: (switch or 0)]), switch, min(x[0] for x in rows)), flush=True)
