"""Randomised stress of the MSM paths (dev tool; run on the GPU box).  Expected value: every base is b_j * G with a known b_j, so the sum
is (sum_i +-k_i b_idx(i) mod r) * G -- ONE scalar multiplication by the C oracle, complete by construction (the reference's naive Add loop
has no P == Q branch, g1.go:32-89, and returns garbage on exactly the degenerate sums this tool is after).  Inputs:
random sizes, skewed scalar distributions (zeros, ones, small values, r - 1, repeated values -> heavy buckets of every size), bases with
duplicates, negated pairs and points at infinity (partial sums that coincide or cancel: the doubling / infinity branches of the tail
kernels' memory-operand addition), G1 and G2, every window width the library may pick, blocking and pipelined.
    python tools/stress_msm_random.py [seconds] [seed]"""
import os
import random
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402

import gosnark_amd  # noqa: F401,E402
from gosnark_amd import capi  # noqa: E402
import gpu_util as U  # noqa: E402
from oracle import c_oracle as C  # noqa: E402
from oracle import ref_py as O  # noqa: E402

if os.environ.get("GS_STRESS_DUMP"):          # a hang shows where: dump every thread's stack after that many seconds and exit
    import faulthandler
    faulthandler.dump_traceback_later(float(os.environ["GS_STRESS_DUMP"]), exit=True)
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
rng = random.Random(seed)
capi.init(0)
t_end = time.time() + budget
cases = 0
while time.time() < t_end:
    g2 = rng.random() < 0.35
    n = rng.choice([1, 2, 3, 17, 64, 255, 256, 257, 1000, 3001, 4096, 9000] if not g2 else [1, 2, 5, 64, 257, 1000, 2500])
    nb = max(1, rng.choice([n, n, max(1, n // 2), max(1, n // 8), 3]))          # distinct bases: duplicates when nb < n
    ks_b = U.rand_scalars_u64(nb, rng.randrange(1 << 30))
    base_small = capi.g2_fixed_base(ks_b) if g2 else capi.g1_fixed_base(ks_b)
    pts = capi.g2_download(base_small) if g2 else capi.g1_download(base_small)
    idx = [rng.randrange(nb) for _ in range(n)]
    arr = pts[idx].copy()
    words = arr.shape[1]
    cw = words // 3
    bvals = capi.u64_to_ints(ks_b)
    coef = [bvals[j] for j in idx]                                               # point i = coef[i] * G
    for i in range(n):
        x = rng.random()
        if x < 0.03:
            arr[i] = 0                                                          # the point at infinity (Z = 0)
            coef[i] = 0
        elif x < 0.15:                                                           # -P: y -> q - y (each Fq component)
            ycoords = capi.u64_to_ints(arr[i, cw:2 * cw].reshape(-1, 4))
            arr[i, cw:2 * cw] = capi.ints_to_u64([(O.Q - y) % O.Q for y in ycoords]).reshape(-1)
            coef[i] = (O.R - coef[i]) % O.R
    uni = U.u64_rows_to_ints(U.rand_scalars_u64(n, rng.randrange(1 << 30)))
    mode = rng.random()
    common = [rng.randrange(1, 1 << rng.choice([1, 4, 16, 32, 64, 200, 253])) for _ in range(rng.choice([1, 2, 5]))]
    ks = []
    for i in range(n):
        x = rng.random()
        if mode < 0.3:
            ks.append(uni[i])
        elif x < 0.25:
            ks.append(0)
        elif x < 0.55:
            ks.append(1)
        elif x < 0.8:
            ks.append(rng.choice(common))
        elif x < 0.9:
            ks.append(O.R - 1 - rng.randrange(3))
        else:
            ks.append(uni[i])
    ksu = capi.ints_to_u64(ks)
    total = sum(k * cf for k, cf in zip(ks, coef)) % O.R
    want = (C.g2_affine(C.g2_mul_scalar(O.G2_GEN, total)) if g2 else C.g1_affine(C.g1_mul_scalar(O.G1_GEN, total))) if total else None
    bases = capi.g2_upload(arr) if g2 else capi.g1_upload(arr)
    sc = capi.scalars_upload(ksu)
    for c in (0, rng.choice([8, 9, 11, 13, 15, 16, 17, 18])):
        capi.set_window_bits(c)
        try:
            got = capi.msm(bases, ksu, g2=g2)
            assert got == want, ("blocking", g2, n, nb, c, seed, cases)
            ts = [capi.msm_begin(bases, sc, n, g2=g2) for _ in range(2)]
            for t in ts:
                assert capi.msm_end(t) == want, ("pipelined", g2, n, nb, c, seed, cases)
        finally:
            capi.set_window_bits(0)
    for h in (bases, sc, base_small):
        h.free()
    cases += 1
print("stress_msm_random: %d cases in %.0f s, seed %d: all equal the closed form" % (cases, budget, seed))
