"""Socket power and shader clock of the GPU beside the loops the "power-bound" argument rests on -- dev tool.

    python tools/power_trace.py [seconds per phase = 4] [log2n = 20]

profiles/r05_pmc_sq_issue_breakdown.txt derives every clock from s_memtime spans inside the kernels.  This script is the independent
witness: a sampler thread reads the driver's own telemetry -- hwmon power1_average / power1_input (socket power, microwatts) and
freq1_input (sclk, Hz) under /sys/class/drm/card*/device/hwmon/, gpu_busy_percent, pp_dpm_sclk -- at ~50 Hz while the main thread runs,
each for `seconds`:
    idle | s_nop loop | v_mad_u64_u32 on 8 chains | fp29.h dots3 products (bare loop) | the G1 addition's instruction mix |
    v_fma_f64 | a stream of 2^log2n-term G1 MSMs (three in flight: k_bucket_accumulate<G1> is ~83 % of its device time) |
    a stream of 2^log2n-constraint Groth16 proofs (three in flight: the judged workload)
and prints, per phase, mean / p5 / p95 of power and sclk, the power cap (power1_cap) and the clock the ubench's own s_memtime span
gives for the same launches (the number DESIGN quotes), so the two can be compared line by line.  When sysfs has no such files the
script falls back to `amd-smi metric` / `rocm-smi` once per ~0.3 s and says so.  Nothing here changes a machine setting (a power-cap
A/B would: not attempted on the shared gpurun boxes).
"""
import glob
import json
import os
import re
import subprocess
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
STREAMS_ONLY = "--streams-only" in sys.argv       # only the library's streams (and a few more of them): which part of a proof pulls the PLL down?
_args = [a for a in sys.argv[1:] if not a.startswith("--")]
SECONDS = float(_args[0]) if len(_args) > 0 else 4.0
LOGN = int(_args[1]) if len(_args) > 1 else 20


def read_int(path):
    try:
        with open(path) as f:
            return int(f.read().split()[0])
    except (OSError, ValueError, IndexError):
        return None


def own_pci_bdf():
    """PCI address of HIP device 0 (the box shows every card of the node under /sys/class/drm, the process sees one of them)."""
    import ctypes
    try:
        hip = ctypes.CDLL("libamdhip64.so")
        buf = ctypes.create_string_buffer(64)
        if hip.hipDeviceGetPCIBusId(buf, 64, 0) == 0:
            return buf.value.decode().lower()
    except OSError:
        pass
    return None


class Sysfs:
    """The hwmon directory of the amdgpu card HIP device 0 is (matched by PCI address; first card that reports power otherwise)."""
    def __init__(self):
        self.power = self.sclk = self.mclk = self.cap = self.busy = self.temp = None
        self.bdf = own_pci_bdf()
        cards = sorted(glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*"))
        self.all_cards = []
        for hw in cards:
            dev = os.path.dirname(os.path.dirname(hw))
            self.all_cards.append((os.path.basename(os.path.realpath(dev)), hw))
        mine = [hw for bdf, hw in self.all_cards if self.bdf and bdf.lower() == self.bdf]
        for hw in (mine or cards):
            for name in ("power1_average", "power1_input"):
                p = os.path.join(hw, name)
                if read_int(p) is not None:
                    self.power = p
                    break
            if self.power:
                for attr, name in (("sclk", "freq1_input"), ("mclk", "freq2_input"), ("cap", "power1_cap"), ("temp", "temp2_input")):
                    p = os.path.join(hw, name)
                    if read_int(p) is not None:
                        setattr(self, attr, p)
                dev = os.path.dirname(os.path.dirname(hw))
                if read_int(os.path.join(dev, "gpu_busy_percent")) is not None:
                    self.busy = os.path.join(dev, "gpu_busy_percent")
                self.dev = dev
                self.matched = bool(mine)
                break

    def sample(self):
        return (time.perf_counter(), read_int(self.power) if self.power else None, read_int(self.sclk) if self.sclk else None,
                read_int(self.busy) if self.busy else None, read_int(self.mclk) if self.mclk else None) + tuple(read_int(p) for _, p in self.temps())

    def temps(self):
        """[(label, path)] of the card's temperature sensors (junction / memory / edge), found once"""
        if not hasattr(self, "_temps"):
            self._temps = []
            hw = os.path.dirname(self.power) if self.power else None
            for p in sorted(glob.glob(os.path.join(hw, "temp*_input"))) if hw else []:
                try:
                    label = open(p.replace("_input", "_label")).read().strip()
                except OSError:
                    label = os.path.basename(p)
                if read_int(p) is not None:
                    self._temps.append((label, p))
        return self._temps


def smi_sample():
    """Fallback: one amd-smi / rocm-smi call (slow: a few per second)."""
    t = time.perf_counter()
    for cmd in (["amd-smi", "metric", "-g", "0", "--power", "--clock", "--json"], ["rocm-smi", "-d", "0", "--showpower", "--showclocks", "--json"]):
        try:
            out = subprocess.run(cmd, capture_output=True, text=True, timeout=5).stdout
            txt = json.dumps(json.loads(out))
        except (OSError, ValueError, subprocess.TimeoutExpired):
            continue
        pw = re.search(r'"(?:socket_power|Average Graphics Package Power \(W\)|Current Socket Graphics Package Power \(W\))"\s*:\s*(?:\{"value":\s*)?"?([0-9.]+)', txt)
        ck = re.search(r'"(?:gfx_0|sclk clock speed:)"\s*:\s*(?:\{"clk":\s*\{"value":\s*)?"?\(?([0-9.]+)', txt)
        if pw or ck:
            return (t, int(float(pw.group(1)) * 1e6) if pw else None, int(float(ck.group(1)) * 1e6) if ck else None, None, None)
    return (t, None, None, None, None)


class Sampler(threading.Thread):
    def __init__(self, sysfs):
        super().__init__(daemon=True)
        self.sysfs, self.rows, self.stop_flag = sysfs, [], False
        self.fast = sysfs.power is not None or sysfs.sclk is not None

    def run(self):
        while not self.stop_flag:
            self.rows.append(self.sysfs.sample() if self.fast else smi_sample())
            time.sleep(0.02 if self.fast else 0.05)


def stats(vals, scale):
    vals = sorted(v * scale for v in vals if v is not None)
    if not vals:
        return "   n/a"
    pick = lambda q: vals[min(len(vals) - 1, int(q * len(vals)))]
    return "%7.1f (p5 %7.1f, p95 %7.1f, n %d)" % (sum(vals) / len(vals), pick(0.05), pick(0.95), len(vals))


def phase(sampler, name, fn):
    t0 = time.perf_counter()
    note = fn() or ""
    t1 = time.perf_counter()
    # the first 25 % of a phase is the transient (clock ramp, power averaging window): report the rest
    lo = t0 + 0.25 * (t1 - t0)
    rows = [r for r in sampler.rows if lo <= r[0] <= t1]
    temps = " ".join("%s %.0f C" % (label, max([r[5 + i] for r in rows if len(r) > 5 + i and r[5 + i] is not None] or [0]) * 1e-3)
                     for i, (label, _) in enumerate(sampler.sysfs.temps())) if sampler.fast else ""
    print("%-46s %5.1f s | power W %s | sclk MHz %s | busy %% %s | max %s" % (name, t1 - t0, stats([r[1] for r in rows], 1e-6), stats([r[2] for r in rows], 1e-6),
                                                                        stats([r[3] for r in rows], 1.0), temps))
    if note:
        print("      " + note.strip().replace("\n", "\n      "))
    sys.stdout.flush()


def ubench(binary, *args):
    def run():
        exe = os.path.join(ROOT, "tools", binary)
        if not os.path.exists(exe):
            return "(%s not built)" % binary
        out = subprocess.run([exe] + [str(a) for a in args], capture_output=True, text=True, timeout=120).stdout
        return "\n".join(l for l in out.splitlines() if "GHz" in l)
    return run


def main():
    fs = Sysfs()
    print("HIP device 0 is PCI %s; cards under /sys/class/drm: %s" % (fs.bdf, ", ".join("%s=%s W" % (b, (read_int(os.path.join(h, "power1_input")) or read_int(os.path.join(h, "power1_average")) or 0) // 1000000) for b, h in fs.all_cards)))
    print("telemetry (%s): power %s | sclk %s | busy %s | cap %s" % ("matched by PCI address" if getattr(fs, "matched", False) else "NOT matched: first card that reports power", fs.power, fs.sclk, fs.busy, fs.cap))
    if fs.cap:
        print("power cap: %.0f W (power1_cap); power1_cap_max %s, power1_cap_default %s" % (
            read_int(fs.cap) * 1e-6, read_int(fs.cap.replace("power1_cap", "power1_cap_max")), read_int(fs.cap.replace("power1_cap", "power1_cap_default"))))
    if getattr(fs, "dev", None):
        for name in ("pp_dpm_sclk", "pp_dpm_mclk", "pp_power_profile_mode", "power_dpm_force_performance_level"):
            try:
                print("%s: %s" % (name, " | ".join(open(os.path.join(fs.dev, name)).read().split("\n")[:12])))
            except OSError:
                pass
    for cmd in (["amd-smi", "static", "-g", "0", "--limit"], ["amd-smi", "metric", "-g", "0", "--power", "--clock"], ["rocm-smi", "--showpower", "--showclocks", "--showmaxpower"]):
        try:
            out = subprocess.run(cmd, capture_output=True, text=True, timeout=20)
            print("$ %s\n%s" % (" ".join(cmd), (out.stdout + out.stderr).strip()[:900]))
        except (OSError, subprocess.TimeoutExpired) as e:
            print("$ %s -> %s" % (" ".join(cmd), e))
    # (a power-cap A/B would mean changing a machine setting: not attempted -- gpurun boxes are shared and run as an ordinary user)
    sampler = Sampler(fs)
    if not sampler.fast:
        print("no hwmon power / clock files: sampling through amd-smi / rocm-smi (a few Hz)")
    sampler.start()
    phase(sampler, "idle", lambda: time.sleep(2.0))
    if not STREAMS_ONLY:
      phase(sampler, "s_nop loop (ubench_issue 11)", ubench("ubench_issue", 11, SECONDS))
      phase(sampler, "v_mad_u64_u32, 8 chains (ubench_issue 0)", ubench("ubench_issue", 0, SECONDS))
      phase(sampler, "fp29 dots3 products, bare loop (ubench_mulmod 2)", ubench("ubench_mulmod", 2, SECONDS))
      phase(sampler, "fp29 dots2 products, bare loop (ubench_mulmod 1)", ubench("ubench_mulmod", 1, SECONDS))
      phase(sampler, "G1-addition instruction mix (ubench_issue 9)", ubench("ubench_issue", 9, SECONDS))
      phase(sampler, "v_fma_f64 (ubench_issue 18)", ubench("ubench_issue", 18, SECONDS))
      phase(sampler, "v_and_b32 VOP2 (ubench_issue 6)", ubench("ubench_issue", 6, SECONDS))

    import torch  # noqa: F401  (device memory / streams for the library's Python side)
    import gosnark_amd  # noqa: F401
    from gosnark_amd import capi, synth, groth16
    capi.init()
    capi.set_table_policy("always")
    n = 1 << LOGN
    inst = synth.sqchain_setup_instance(n, 3)
    pk = inst.device_pk()
    r, s = synth.field_elems(2, 5)
    groth16.prove_resident(pk, inst.w, inst.px, r, s)           # builds the tables
    bases = capi.g1_fixed_base(synth.scalars_u64(n, 7))
    sc = capi.scalars_upload(synth.scalars_u64(n, 8))
    capi.msm_resident(bases, sc, n)

    def msm_stream():
        t_end = time.perf_counter() + SECONDS
        tickets, done, t0 = [], 0, time.perf_counter()
        while time.perf_counter() < t_end:
            while len(tickets) < 3:
                tickets.append(capi.msm_begin(bases, sc, n))
            capi.msm_end(tickets.pop(0))
            done += 1
        for t in tickets:
            capi.msm_end(t)
        tm = capi.last_timing()
        return "%d MSMs, %.3f ms each; last one: accumulate %.3f ms of %.3f ms device" % (done, (time.perf_counter() - t0) * 1e3 / max(done + len(tickets), 1),
                                                                                     tm["accumulate_ms"], tm["total_ms"])

    def proof_stream():
        t_end = time.perf_counter() + SECONDS
        tickets, done, t0 = [], 0, time.perf_counter()
        while time.perf_counter() < t_end:
            while len(tickets) < 3:
                tickets.append(groth16.prove_begin(pk, inst.w, inst.px, r, s))
            groth16.prove_end(tickets.pop(0))
            done += 1
        for t in tickets:
            groth16.prove_end(t)
        return "%d proofs, %.3f ms each" % (done, (time.perf_counter() - t0) * 1e3 / max(done + len(tickets), 1))

    phase(sampler, "G1 MSM stream, 2^%d terms, three in flight" % LOGN, msm_stream)
    # the G2 sum alone (39 % of a proof's device time; 256 VGPRs, two waves per SIMD): does IT pull the PLL down?
    bases2 = capi.g2_fixed_base(synth.scalars_u64(n // 2, 9))
    capi.msm_resident(bases2, sc, n // 2, g2=True)

    def msm2_stream():
        t_end = time.perf_counter() + SECONDS
        tickets, done, t0 = [], 0, time.perf_counter()
        while time.perf_counter() < t_end:
            while len(tickets) < 3:
                tickets.append(capi.msm_begin(bases2, sc, n // 2, g2=True))
            capi.msm_end(tickets.pop(0))
            done += 1
        for t in tickets:
            capi.msm_end(t)
        return "%d G2 MSMs of 2^%d terms, %.3f ms each" % (done, LOGN - 1, (time.perf_counter() - t0) * 1e3 / max(done + len(tickets), 1))
    phase(sampler, "G2 MSM stream, 2^%d terms, three in flight" % (LOGN - 1), msm2_stream)
    phase(sampler, "Groth16 proof stream, 2^%d, three in flight" % LOGN, proof_stream)
    if STREAMS_ONLY:
        from gosnark_amd import r1csqap
        dr = r1csqap.DeviceR1CS(*inst.r1cs, inst.m)

        def px_stream():                      # sparse mat-vecs + three interpolations + a size-2n product: NTT passes and point-wise kernels only
            t_end, done, t0 = time.perf_counter() + SECONDS, 0, time.perf_counter()
            ph = dr.ComputePxResident(inst.w)
            while time.perf_counter() < t_end:
                ph = dr.ComputePxResident(inst.w, ph)
                done += 1
            return "%d x gs_r1cs_px (NTT passes, no curve arithmetic), %.3f ms each" % (done, (time.perf_counter() - t0) * 1e3 / max(done, 1))

        def witness_stream():
            t_end = time.perf_counter() + SECONDS
            tickets, done, t0 = [], 0, time.perf_counter()
            while time.perf_counter() < t_end:
                while len(tickets) < 3:
                    tickets.append(groth16.prove_witness_begin(pk, dr, inst.w, r, s))
                groth16.prove_end(tickets.pop(0))
                done += 1
            for t in tickets:
                groth16.prove_end(t)
            return "%d witness -> proof, %.3f ms each" % (done, (time.perf_counter() - t0) * 1e3 / max(done + len(tickets), 1))

        def blocking_stream():
            t_end, done, t0 = time.perf_counter() + SECONDS, 0, time.perf_counter()
            while time.perf_counter() < t_end:
                groth16.prove_resident(pk, inst.w, inst.px, r, s)
                done += 1
            return "%d blocking proofs, %.3f ms each" % (done, (time.perf_counter() - t0) * 1e3 / max(done, 1))
        def alt_msm_stream():                 # G1 and G2 MSM tickets alternating: is it the CHANGE between the two accumulation kernels?
            t_end = time.perf_counter() + SECONDS
            tickets, done, t0, k = [], 0, time.perf_counter(), 0
            while time.perf_counter() < t_end:
                while len(tickets) < 3:
                    tickets.append(capi.msm_begin(bases2, sc, n // 2, g2=True) if k % 2 else capi.msm_begin(bases, sc, n))
                    k += 1
                capi.msm_end(tickets.pop(0))
                done += 1
            for t in tickets:
                capi.msm_end(t)
            return "%d MSMs (G1 2^%d / G2 2^%d alternating), %.3f ms each" % (done, LOGN, LOGN - 1, (time.perf_counter() - t0) * 1e3 / max(done + len(tickets), 1))

        def msm_px_stream():                  # G1 MSM tickets with a gs_r1cs_px between them: curve arithmetic and NTT passes alternating
            t_end = time.perf_counter() + SECONDS
            tickets, done, t0 = [], 0, time.perf_counter()
            ph = dr.ComputePxResident(inst.w)
            while time.perf_counter() < t_end:
                while len(tickets) < 3:
                    tickets.append(capi.msm_begin(bases, sc, n))
                ph = dr.ComputePxResident(inst.w, ph)
                capi.msm_end(tickets.pop(0))
                done += 1
            for t in tickets:
                capi.msm_end(t)
            return "%d x (G1 MSM + gs_r1cs_px), %.3f ms each" % (done, (time.perf_counter() - t0) * 1e3 / max(done + len(tickets), 1))

        def pin_stream():
            pin = synth.sqchain_pinocchio_instance(n, 4)
            from gosnark_amd import snark
            ppk = pin.device_pk()
            snark.prove_resident(ppk, pin.w, pin.px)
            t_end = time.perf_counter() + SECONDS
            tickets, done, t0 = [], 0, time.perf_counter()
            while time.perf_counter() < t_end:
                while len(tickets) < 3:
                    tickets.append(snark.prove_begin(ppk, pin.w, pin.px))
                snark.prove_end(tickets.pop(0))
                done += 1
            for t in tickets:
                snark.prove_end(t)
            return "%d Pinocchio proofs (7 G1 sums, 1 G2), %.3f ms each" % (done, (time.perf_counter() - t0) * 1e3 / max(done + len(tickets), 1))
        phase(sampler, "G1 / G2 MSM tickets alternating", alt_msm_stream)
        phase(sampler, "G1 MSM tickets + gs_r1cs_px alternating", msm_px_stream)
        phase(sampler, "gs_r1cs_px stream (NTT passes only)", px_stream)
        phase(sampler, "witness -> proof stream, three in flight", witness_stream)
        phase(sampler, "blocking proofs back to back", blocking_stream)
        phase(sampler, "Groth16 proof stream again", proof_stream)
        phase(sampler, "Pinocchio proof stream, three in flight", pin_stream)
    phase(sampler, "idle again", lambda: time.sleep(2.0))
    sampler.stop_flag = True
    sampler.join(timeout=2)
    gaps = [b[0] - a[0] for a, b in zip(sampler.rows, sampler.rows[1:])]
    if gaps:
        print("sampler: %d samples, median period %.1f ms (%.0f Hz)" % (len(sampler.rows), sorted(gaps)[len(gaps) // 2] * 1e3, 1.0 / max(sorted(gaps)[len(gaps) // 2], 1e-9)))


if __name__ == "__main__":
    main()
