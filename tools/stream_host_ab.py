"""A/B of how a host-buffer ticket's arrays reach its slot (GS_HOST_STAGE = 0 / 1 / 2, csrc/prove.hip) -- dev tool.
One process per mode (the switch is read once); prints the ms per proof of bench.py's distinct-witness streams."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1 and sys.argv[1] == "--child":
    sys.path.insert(0, ROOT)
    import bench
    from gosnark_amd import capi, synth
    capi.init(0)
    capi.set_table_policy("always")
    n = 1 << int(sys.argv[2])
    inst = synth.sqchain_setup_instance(n, 0x5EED0002)
    r_, s_ = synth.field_elems(2, 77)
    out = bench.stream_distinct_host(inst, inst.device_pk(), n, r_, s_, check=False)
    print(json.dumps({k: (v["ms_per_proof"] if isinstance(v, dict) and "ms_per_proof" in v else v) for k, v in out.items()
                      if k in ("witness_host", "px_host", "update", "resident", "px_resident_same_witness")}))
else:
    logn = sys.argv[1] if len(sys.argv) > 1 else "20"
    for rnd in range(2):
        for mode in os.environ.get("GS_AB_MODES", "0,1,2").split(","):
            env = dict(os.environ, GS_HOST_STAGE=mode)
            res = subprocess.run([sys.executable, os.path.abspath(__file__), "--child", logn], env=env, capture_output=True, text=True)
            line = [x for x in res.stdout.splitlines() if x.startswith("{")]
            print("GS_HOST_STAGE=%s GS_COPY_THREADS=%s round %d: %s" % (mode, os.environ.get("GS_COPY_THREADS", "4"), rnd, line[-1] if line else "FAILED " + res.stderr[-400:]), flush=True)
