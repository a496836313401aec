"""The shader clock at 20 us resolution while the library's streams run -- dev tool (round 6).

    python tools/clock_timeline.py [log2n = 20] [milliseconds = 60] [interval_us = 20]

tools/clock_probe.hip: one wave on its own high-priority stream reads s_memtime (ticks at sclk) against s_memrealtime (100 MHz) every
`interval_us`.  This script starts it in the steady state of each stream -- G1 MSM tickets, G2 MSM tickets, Groth16 proofs (three in
flight), Pinocchio proofs, gs_r1cs_px, blocking proofs -- and prints, per stream: the mean clock, its histogram in 50 MHz bins, the share
of the time below 2.2 GHz, and 12 ms of the timeline (one figure per 100 us, in units of 10 MHz) in which the proof period shows.
hwmon's 49 Hz sclk (tools/power_trace.py) averages over what this resolves; rocprofv3's GRBM_GUI_ACTIVE / duration
(profiles/r06_pmc_kernel_clocks.txt) needs a PMC pass that runs the kernels one after another.
"""
import ctypes
import os
import sys
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
LOGN = int(sys.argv[1]) if len(sys.argv) > 1 else 20
MS = float(sys.argv[2]) if len(sys.argv) > 2 else 60.0
INTERVAL_US = int(sys.argv[3]) if len(sys.argv) > 3 else 20


def main():
    import torch  # noqa: F401
    import gosnark_amd  # noqa: F401
    from gosnark_amd import capi, synth, groth16, snark
    lib = ctypes.CDLL(os.path.join(ROOT, "tools", "libclock_probe.so"))
    lib.clock_probe_start.argtypes = [ctypes.c_uint32, ctypes.c_uint32]
    lib.clock_probe_finish.argtypes = [ctypes.c_void_p]
    capi.init()
    capi.set_table_policy("always")
    n = 1 << LOGN
    inst = synth.sqchain_setup_instance(n, 3)
    pk = inst.device_pk()
    r, s = synth.field_elems(2, 5)
    groth16.prove_resident(pk, inst.w, inst.px, r, s)
    bases = capi.g1_fixed_base(synth.scalars_u64(n, 7))
    sc = capi.scalars_upload(synth.scalars_u64(n, 8))
    capi.msm_resident(bases, sc, n)
    n2 = n // 2
    bases2 = capi.g2_fixed_base(synth.scalars_u64(n2, 9))
    capi.msm_resident(bases2, sc, n2, g2=True)
    samples = int(MS * 1e3 / INTERVAL_US)

    def ticketed(begin, end):
        def run(stop):
            tickets, done, t0 = [], 0, time.perf_counter()
            while not stop.is_set():
                while len(tickets) < 3:
                    tickets.append(begin())
                end(tickets.pop(0))
                done += 1
            for t in tickets:
                end(t)
            return done, time.perf_counter() - t0
        return run

    def blocking(fn):
        def run(stop):
            done, t0 = 0, time.perf_counter()
            while not stop.is_set():
                fn()
                done += 1
            return done, time.perf_counter() - t0
        return run

    def measure(name, stream_fn):
        stop = threading.Event()
        result = {}
        th = None
        if stream_fn is not None:
            th = threading.Thread(target=lambda: result.update(r=stream_fn(stop)))
            th.start()
            time.sleep(1.5)                                         # the stream's steady state (clock ramp, power averaging)
        pairs = np.zeros(2 * samples, dtype=np.uint64)
        rc = lib.clock_probe_start(samples, INTERVAL_US)
        rc2 = lib.clock_probe_finish(pairs.ctypes.data) if rc == 0 else -9
        stop.set()
        if th:
            th.join()
        if rc or rc2:
            print("%s: probe failed (%d, %d)" % (name, rc, rc2)); return
        real, clk = pairs[0::2].astype(np.float64), pairs[1::2].astype(np.float64)
        dt = np.diff(real) * 1e-8                                   # seconds (100 MHz)
        ghz = np.diff(clk) / dt * 1e-9
        ghz = ghz[(dt > 0) & (ghz > 0.05) & (ghz < 3.0)]
        per = ("%d operations, %.3f ms each" % (result["r"][0], result["r"][1] * 1e3 / max(result["r"][0], 1))) if "r" in result else ""
        print("%-44s mean %.3f GHz | p5 %.3f p50 %.3f p95 %.3f | below 2.2 GHz %4.1f %% of the time, below 2.0: %4.1f %% | %d samples of %d us (median gap %.1f us) | %s" % (
            name, ghz.mean(), np.percentile(ghz, 5), np.percentile(ghz, 50), np.percentile(ghz, 95), 100.0 * (ghz < 2.2).mean(), 100.0 * (ghz < 2.0).mean(),
            len(ghz), INTERVAL_US, float(np.median(dt)) * 1e6, per))
        edges = np.arange(1.5, 2.5001, 0.05)
        hist, _ = np.histogram(np.clip(ghz, 1.5, 2.4999), bins=edges)
        print("      histogram, 50 MHz bins from 1.50 GHz (%% of samples): " + " ".join("%.0f" % (100.0 * h / max(len(ghz), 1)) for h in hist))
        per100 = max(1, int(round(100.0 / INTERVAL_US)))
        k = (len(ghz) // per100) * per100
        coarse = ghz[:k].reshape(-1, per100).mean(axis=1)[:120]
        print("      12 ms of it, one figure per 100 us, x 10 MHz: " + " ".join("%.0f" % (100.0 * v) for v in coarse))
        sys.stdout.flush()

    measure("idle (the probe alone)", None)
    measure("G1 MSM stream, 2^%d terms, three in flight" % LOGN, ticketed(lambda: capi.msm_begin(bases, sc, n), capi.msm_end))
    measure("G2 MSM stream, 2^%d terms, three in flight" % (LOGN - 1), ticketed(lambda: capi.msm_begin(bases2, sc, n2, g2=True), capi.msm_end))
    measure("Groth16 proof stream, 2^%d, three in flight" % LOGN, ticketed(lambda: groth16.prove_begin(pk, inst.w, inst.px, r, s), groth16.prove_end))
    measure("blocking Groth16 proofs back to back", blocking(lambda: groth16.prove_resident(pk, inst.w, inst.px, r, s)))
    from gosnark_amd import r1csqap
    dr = r1csqap.DeviceR1CS(*inst.r1cs, inst.m)
    state = {"ph": dr.ComputePxResident(inst.w)}

    def px_once():
        state["ph"] = dr.ComputePxResident(inst.w, state["ph"])
    measure("gs_r1cs_px stream (NTT passes only)", blocking(px_once))
    measure("witness -> proof stream, three in flight", ticketed(lambda: groth16.prove_witness_begin(pk, dr, inst.w, r, s), groth16.prove_end))
    pin = synth.sqchain_pinocchio_instance(n, 4)
    ppk = pin.device_pk()
    snark.prove_resident(ppk, pin.w, pin.px)
    measure("Pinocchio proof stream, three in flight", ticketed(lambda: snark.prove_begin(ppk, pin.w, pin.px), snark.prove_end))
    measure("idle again (the probe alone)", None)


if __name__ == "__main__":
    main()
