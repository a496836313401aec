"""Experiment: do independent accumulation kernels overlap usefully at small sizes?  The same GPU listed as 1, 2, 3 logical
devices (each with its own stream set, three proofs in flight) proving the same 2^k instance through gs_groth16_prove_batch."""
import sys, time
sys.path.insert(0, ".")
import gosnark_amd  # noqa
from gosnark_amd import capi, groth16, synth

logn = int(sys.argv[1]) if len(sys.argv) > 1 else 16
P = int(sys.argv[2]) if len(sys.argv) > 2 else 240
maxdev = 3
capi.init([0] * maxdev)
capi.set_device(0)
inst = synth.sqchain_setup_instance(1 << logn, 0x7700 + logn)
rs = [tuple(synth.field_elems(2, 100 + i)) for i in range(P)]
for ndev in (1, 2, 3, 1, 2, 3):
    pks = [groth16.ShardPkTo(inst.device_pk(), 0, 1, d) for d in range(ndev)] + [None] * 0
    ws = [capi.scalars_clone(inst.w, i % ndev) for i in range(P)]
    pxs = [capi.scalars_clone(inst.px, i % ndev) for i in range(P)]
    groth16.prove_batch(pks, ws[:6 * ndev], pxs[:6 * ndev], rs[:6 * ndev])        # warm up: tables, workspaces
    t0 = time.perf_counter()
    got = groth16.prove_batch(pks, ws, pxs, rs)
    dt = (time.perf_counter() - t0) / P * 1e3
    print("2^%d, %d logical device(s) on one GPU: %.4f ms per proof, %.1f M constraints/s" % (logn, ndev, dt, (1 << logn) / dt / 1e3), flush=True)
    for h in ws + pxs:
        h.free()
    for k in pks:
        k.handle.free()
