import sys, time
sys.path.insert(0, "/root/repo")
import torch, gosnark_amd
from gosnark_amd import capi, synth
capi.init()
for logn in (14, 16, 18):
    n = 1 << logn
    bases = capi.g1_fixed_base(synth.scalars_u64(n, 1)); sc = capi.scalars_upload(synth.scalars_u64(n, 2))
    for c in (0, 11, 12, 13, 14, 15, 16):
        if c and c > logn + 1: continue
        capi.set_window_bits(c)
        ref = capi.msm_resident(bases, sc, n)
        for _ in range(3): capi.msm_resident(bases, sc, n)
        torch.cuda.synchronize(); t = time.perf_counter()
        for _ in range(20): capi.msm_resident(bases, sc, n)
        blk = (time.perf_counter() - t) / 20 * 1e3
        tick = []
        t = time.perf_counter()
        for _ in range(40):
            tick.append(capi.msm_begin(bases, sc, n))
            if len(tick) == 3: capi.msm_end(tick.pop(0))
        while tick: capi.msm_end(tick.pop(0))
        pipe = (time.perf_counter() - t) / 40 * 1e3
        print("2^%d c=%2d blocking %.3f ms pipelined %.3f ms" % (logn, c, blk, pipe), flush=True)
    capi.set_window_bits(0)
