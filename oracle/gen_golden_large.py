#!/usr/bin/env python3
"""ORACLE TOOLING (test infrastructure, not product code).

BASELINE configs[1] as worded -- "2^16 G1 Pippenger MSM, bit-exact vs the bn128.G1 loop" -- and a 2^16-constraint Groth16
proof: golden AFFINE outputs computed offline by the C restatement of the reference's naive loops (oracle/gs_oracle.c:
MulScalar = MSB-first double-and-add, Add = add-2007-bl, Div = schoolbook) on all host cores, on the same seeded inputs
the GPU tests rebuild (gosnark_amd.synth).  Takes ~10 minutes on 8 cores; run in the build container:

    python3 oracle/gen_golden_large.py [msm|msm20|msm22|partials20|prove20|pinocchio20|prove|pinocchio|all]

Writes tests/golden/oracle_msm_g1_2p16.json, oracle_msm_g1_2p20.json, oracle_msm_g1_2p22.json, oracle_groth_partials_2p20.json, oracle_groth_quotient_2p12.json / _2p18.json / _2p20.json, oracle_pinocchio_quotient_2p12.json / _2p20.json, tests/golden/oracle_groth_2p16.json and tests/golden/oracle_pinocchio_2p16.json.
"""
import json
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
from oracle import c_oracle as C, ref_py as O  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")
THREADS = os.cpu_count() or 1
_R_LIMBS = [(O.R >> (64 * i)) & 0xFFFFFFFFFFFFFFFF for i in range(4)]


def scalars_u64(n, seed):
    """Bit-identical to gosnark_amd.synth.scalars_u64 (kept separate: the oracle must not import the product package)."""
    rng = np.random.Generator(np.random.PCG64(seed))
    out = np.zeros((n, 4), dtype=np.uint64)
    todo = np.arange(n)
    while todo.size:
        cand = rng.integers(0, 2**64, size=(todo.size, 4), dtype=np.uint64)
        cand[:, 3] &= np.uint64(0x3FFFFFFFFFFFFFFF)
        lt = np.zeros(todo.size, dtype=bool)
        eq = np.ones(todo.size, dtype=bool)
        for i in (3, 2, 1, 0):
            lt |= eq & (cand[:, i] < np.uint64(_R_LIMBS[i]))
            eq &= cand[:, i] == np.uint64(_R_LIMBS[i])
        out[todo[lt]] = cand[lt]
        todo = todo[~lt]
    return out


def ints(a):
    raw = np.ascontiguousarray(a, dtype="<u8").tobytes()
    return [int.from_bytes(raw[i:i + 32], "little") for i in range(0, len(raw), 32)]


def msm_golden(logn=16, seed=0x60D0):
    n = 1 << logn
    t0 = time.time()
    bases = C.mul_scalar_batch(O.G1_GEN, scalars_u64(n, seed), threads=THREADS)           # P_i = k_i * G
    sc = scalars_u64(n, seed + 1)
    got = C.g1_affine(C.g1_msm_naive(bases, sc, threads=THREADS))                           # sum_i MulScalar(P_i, s_i)
    rec = {"what": "sum_i s_i * (k_i * G1) by the naive MulScalar / Add loop (bn128/g1.go:140-155, :32-89), affine",
           "n": n, "seed_bases": seed, "seed_scalars": seed + 1, "x": str(got[0]), "y": str(got[1]),
           "generator": "oracle/gen_golden_large.py msm (oracle/gs_oracle.c, %d threads, %.1f s)" % (THREADS, time.time() - t0)}
    with open(os.path.join(OUT, "oracle_msm_g1_2p%d.json" % logn), "w") as f:
        json.dump(rec, f, indent=1)
    print(rec)


def zpoly(deg):
    """prod_{i=1}^{deg} (x - i) over Fr by a product tree of schoolbook products (r1csqap.go:57-67 / groth16.go:122-131)."""
    polys = [C.poly_u64([(-i) % O.R, 1]) for i in range(1, deg + 1)]
    while len(polys) > 1:
        nxt = [C.poly_mul_u64(polys[i], polys[i + 1]) if i + 1 < len(polys) else polys[i] for i in range(0, len(polys), 2)]
        polys = nxt
    return polys[0]


def prove_golden(logn=16, seed=0x60D1):
    """The instance gosnark_amd.synth.RandomInstance(n, seed) defines (key points k_i * G, uniform w and px)."""
    n = 1 << logn
    m = n + 1
    t0 = time.time()
    fb1 = lambda cnt, sd: C.mul_scalar_batch(O.G1_GEN, scalars_u64(cnt, sd), threads=THREADS)      # noqa: E731
    at, bacgamma, bacdelta, ptd = fb1(m, seed + 1), fb1(m, seed + 2), fb1(m, seed + 3), fb1(n, seed + 4)
    bacgamma2 = C.mul_scalar_batch(O.G2_GEN, scalars_u64(m, seed + 5), g2=True, threads=THREADS)
    s1 = fb1(3, seed + 6)
    s2 = C.mul_scalar_batch(O.G2_GEN, scalars_u64(2, seed + 7), g2=True, threads=1)
    alpha, beta, delta = (tuple(ints(s1[i])) for i in range(3))
    g2pt = lambda row: (lambda v: ((v[0], v[1]), (v[2], v[3]), (v[4], v[5])))(ints(row))              # noqa: E731
    beta2, delta2 = g2pt(s2[0]), g2pt(s2[1])
    w = scalars_u64(m, seed + 8)
    w[0] = (1, 0, 0, 0)
    px = scalars_u64(2 * n - 1, seed + 9)
    print("key rebuilt on the CPU in %.0f s" % (time.time() - t0), flush=True)
    z = zpoly(m - 2)
    print("Z built, %.0f s" % (time.time() - t0), flush=True)
    hx, _ = C.poly_div_u64(px, z)                                                          # groth16.go:266 (quotient only)
    print("hx = px / Z done, %.0f s" % (time.time() - t0), flush=True)
    r, s = (int(x) % O.R for x in ints(scalars_u64(2, seed + 10)))
    G1, G2 = O.G1, O.G2
    # groth16.go:243-275, MSMs by the naive loop on all cores
    piA = C.g1_msm_naive(at, w, threads=THREADS)
    piB1 = C.g1_msm_naive(bacgamma, w, threads=THREADS)
    piB = C.g2_msm_naive(bacgamma2, w, threads=THREADS)
    piC = C.g1_msm_naive(bacdelta[2:], w[2:], threads=THREADS)                              # i > NPublic = 1
    hsum = C.g1_msm_naive(ptd[:hx.shape[0]], hx, threads=THREADS)
    print("MSMs done, %.0f s" % (time.time() - t0), flush=True)
    piA = G1.Add(G1.Add(piA, alpha), G1.MulScalar(delta, r))
    piB = G2.Add(G2.Add(piB, beta2), G2.MulScalar(delta2, s))
    piB1 = G1.Add(G1.Add(piB1, beta), G1.MulScalar(delta, s))
    piC = G1.Add(piC, hsum)
    piC = G1.Add(piC, G1.MulScalar(piA, s))
    piC = G1.Add(piC, G1.MulScalar(piB1, r))
    piC = G1.Add(piC, G1.Neg(G1.MulScalar(delta, r * s % O.R)))
    a, b, c = G1.Affine(piA), G2.Affine(piB), G1.Affine(piC)
    rec = {"what": "groth16.GenerateProofs (groth16.go:225-278) on gosnark_amd.synth.RandomInstance(n, seed), r and s from "
                   "scalars_u64(2, seed + 10); MSMs by the naive loops, hx by schoolbook Div; affine",
           "n": n, "seed": seed, "r": str(r), "s": str(s),
           "PiA": [str(a[0]), str(a[1])], "PiB": [[str(b[0][0]), str(b[0][1])], [str(b[1][0]), str(b[1][1])]], "PiC": [str(c[0]), str(c[1])],
           "generator": "oracle/gen_golden_large.py prove (oracle/gs_oracle.c + oracle/ref_py.py tail, %d threads, %.0f s)" % (THREADS, time.time() - t0)}
    with open(os.path.join(OUT, "oracle_groth_2p%d.json" % logn), "w") as f:
        json.dump(rec, f, indent=1)
    print(rec)


def partials_golden(logn=20, seed=0x60D4):
    """The four sums over w of groth16.GenerateProofs (groth16.go:243-250) at the headline size, on the key and witness
    gosnark_amd.synth.RandomInstance(n, seed) defines: what gs_groth16_prove_partials returns before the O(1) tail.  (The fifth sum
    needs hx = px / Z, and schoolbook Div at 2^20 is a day of CPU: it is pinned at 2^16 by prove_golden.)"""
    n = 1 << logn
    m = n + 1
    t0 = time.time()
    fb1 = lambda cnt, sd: C.mul_scalar_batch(O.G1_GEN, scalars_u64(cnt, sd), threads=THREADS)      # noqa: E731
    w = scalars_u64(m, seed + 8)
    w[0] = (1, 0, 0, 0)
    rec = {"what": "sum_i w_i * P_i for P = G1.At, G1.BACGamma, G2.BACGamma (all i) and BACDelta (i > NPublic = 1) of "
                   "gosnark_amd.synth.RandomInstance(n, seed), by the naive MulScalar / Add loops (groth16.go:243-250); affine",
           "n": n, "seed": seed}
    at = fb1(m, seed + 1)
    a = O.G1.Affine(C.g1_msm_naive(at, w, threads=THREADS)); del at
    rec["At"] = [str(a[0]), str(a[1])]
    print("At done, %.0f s" % (time.time() - t0), flush=True)
    bg = fb1(m, seed + 2)
    a = O.G1.Affine(C.g1_msm_naive(bg, w, threads=THREADS)); del bg
    rec["BACGamma1"] = [str(a[0]), str(a[1])]
    bd = fb1(m, seed + 3)
    a = O.G1.Affine(C.g1_msm_naive(bd[2:], w[2:], threads=THREADS)); del bd
    rec["BACDelta"] = [str(a[0]), str(a[1])]
    print("G1 sums done, %.0f s" % (time.time() - t0), flush=True)
    b2 = C.mul_scalar_batch(O.G2_GEN, scalars_u64(m, seed + 5), g2=True, threads=THREADS)
    b = O.G2.Affine(C.g2_msm_naive(b2, w, threads=THREADS))
    rec["BACGamma2"] = [[str(b[0][0]), str(b[0][1])], [str(b[1][0]), str(b[1][1])]]
    rec["generator"] = "oracle/gen_golden_large.py partials20 (oracle/gs_oracle.c, %d threads, %.0f s)" % (THREADS, time.time() - t0)
    with open(os.path.join(OUT, "oracle_groth_partials_2p%d.json" % logn), "w") as f:
        json.dump(rec, f, indent=1)
    print(rec["generator"])


def zpoly_fast(deg):
    """prod_{i=1}^{deg} (x - i): schoolbook products (oracle/gs_oracle.c) while the factors are short, the exact multi-prime product
    of oracle/crt_ntt.py above that."""
    from oracle import crt_ntt
    polys = [C.poly_u64([(-i) % O.R, 1]) for i in range(1, deg + 1)]
    while len(polys) > 1:
        big = polys[0].shape[0] > 512
        mul = crt_ntt.poly_mul_mod_r if big else C.poly_mul_u64
        polys = [mul(polys[i], polys[i + 1]) if i + 1 < len(polys) else polys[i] for i in range(0, len(polys), 2)]
    return polys[0]


def prove20_golden(logn=20, seed=0x60D5):
    """A COMPLETE Groth16 proof at the headline size from outside the library, on gosnark_amd.synth.QuotientInstance(n, seed): the key
    and witness of RandomInstance, and px = hx * Z + rem for seeded hx and rem, so that floor(px / Z) = hx is known without the
    schoolbook Div (a day of CPU at this size).  px itself is built by oracle/crt_ntt.py and its SHA-256 recorded."""
    import hashlib
    from oracle import crt_ntt
    n = 1 << logn
    m = n + 1
    t0 = time.time()
    z = zpoly_fast(m - 2)
    print("Z built, %.0f s" % (time.time() - t0), flush=True)
    hx = scalars_u64(n, seed + 11)
    rem = scalars_u64(n - 2, seed + 12)
    prod = crt_ntt.poly_mul_mod_r(hx, z)
    pi, ri = ints(prod), ints(rem)
    px = C.poly_u64([(pi[i] + (ri[i] if i < len(ri) else 0)) % O.R for i in range(len(pi))])
    sha = hashlib.sha256(np.ascontiguousarray(px, dtype="<u8").tobytes()).hexdigest()
    print("px = hx Z + rem built, %.0f s" % (time.time() - t0), flush=True)
    fb1 = lambda cnt, sd: C.mul_scalar_batch(O.G1_GEN, scalars_u64(cnt, sd), threads=THREADS)      # noqa: E731
    w = scalars_u64(m, seed + 8)
    w[0] = (1, 0, 0, 0)
    G1, G2 = O.G1, O.G2
    at = fb1(m, seed + 1)
    piA = C.g1_msm_naive(at, w, threads=THREADS); del at
    bg = fb1(m, seed + 2)
    piB1 = C.g1_msm_naive(bg, w, threads=THREADS); del bg
    bd = fb1(m, seed + 3)
    piC = C.g1_msm_naive(bd[2:], w[2:], threads=THREADS); del bd                             # i > NPublic = 1
    ptd = fb1(n, seed + 4)
    hsum = C.g1_msm_naive(ptd[:n], hx, threads=THREADS); del ptd
    print("G1 sums done, %.0f s" % (time.time() - t0), flush=True)
    b2 = C.mul_scalar_batch(O.G2_GEN, scalars_u64(m, seed + 5), g2=True, threads=THREADS)
    piB = C.g2_msm_naive(b2, w, threads=THREADS); del b2
    s1 = fb1(3, seed + 6)
    s2 = C.mul_scalar_batch(O.G2_GEN, scalars_u64(2, seed + 7), g2=True, threads=1)
    alpha, beta, delta = (tuple(ints(s1[i])) for i in range(3))
    g2pt = lambda row: (lambda v: ((v[0], v[1]), (v[2], v[3]), (v[4], v[5])))(ints(row))              # noqa: E731
    beta2, delta2 = g2pt(s2[0]), g2pt(s2[1])
    r, s = (int(x) % O.R for x in ints(scalars_u64(2, seed + 10)))
    piA = G1.Add(G1.Add(piA, alpha), G1.MulScalar(delta, r))
    piB = G2.Add(G2.Add(piB, beta2), G2.MulScalar(delta2, s))
    piB1 = G1.Add(G1.Add(piB1, beta), G1.MulScalar(delta, s))
    piC = G1.Add(piC, hsum)
    piC = G1.Add(piC, G1.MulScalar(piA, s))
    piC = G1.Add(piC, G1.MulScalar(piB1, r))
    piC = G1.Add(piC, G1.Neg(G1.MulScalar(delta, r * s % O.R)))
    a, b, c = G1.Affine(piA), G2.Affine(piB), G1.Affine(piC)
    rec = {"what": "groth16.GenerateProofs (groth16.go:225-278) on gosnark_amd.synth.QuotientInstance(n, seed): px = hx Z + rem built by "
                   "oracle/crt_ntt.py (SHA-256 below), so hx = floor(px / Z) is known; the five MSMs by the naive loops; affine",
           "n": n, "seed": seed, "r": str(r), "s": str(s), "px_sha256": sha,
           "PiA": [str(a[0]), str(a[1])], "PiB": [[str(b[0][0]), str(b[0][1])], [str(b[1][0]), str(b[1][1])]], "PiC": [str(c[0]), str(c[1])],
           "generator": "oracle/gen_golden_large.py prove20 (oracle/gs_oracle.c + oracle/crt_ntt.py + oracle/ref_py.py tail, %d threads, %.0f s)"
                        % (THREADS, time.time() - t0)}
    with open(os.path.join(OUT, "oracle_groth_quotient_2p%d.json" % logn), "w") as f:
        json.dump(rec, f, indent=1)
    print({k: v for k, v in rec.items() if k in ("n", "seed", "px_sha256", "generator")})


def pinocchio20_golden(logn=20, seed=0x60D7):
    """A COMPLETE snark.GenerateProofs output at the headline size from outside the library, on
    gosnark_amd.synth.QuotientPinocchioInstance(n, seed): px = hx Z + rem (oracle/crt_ntt.py), the eight sums by the naive loops."""
    import hashlib
    from oracle import crt_ntt
    n = 1 << logn
    m = n + 1
    t0 = time.time()
    z = zpoly_fast(m - 2)
    hx = scalars_u64(n, seed + 11)
    rem = scalars_u64(n - 2, seed + 12)
    pi, ri = ints(crt_ntt.poly_mul_mod_r(hx, z)), ints(rem)
    px = C.poly_u64([(pi[i] + (ri[i] if i < len(ri) else 0)) % O.R for i in range(len(pi))])
    sha = hashlib.sha256(np.ascontiguousarray(px, dtype="<u8").tobytes()).hexdigest()
    del pi, ri, px, z
    print("px = hx Z + rem built, %.0f s" % (time.time() - t0), flush=True)
    fb1 = lambda cnt, sd: C.mul_scalar_batch(O.G1_GEN, scalars_u64(cnt, sd), threads=THREADS)      # noqa: E731
    w = scalars_u64(m, seed + 9)
    w[0] = (1, 0, 0, 0)
    out = {}
    for i, k in enumerate(("A", "Ap", "Bp", "C", "Cp", "Kp")):
        arr = fb1(m, seed + 1 + i)
        lo = 2 if k in ("A", "Ap") else 0                                                   # i > NPublic = 1 only (snark.go:265-268)
        out["Pi" + k] = O.G1.Affine(C.g1_msm_naive(arr[lo:], w[lo:], threads=THREADS))
        del arr
        print("Pi%s done, %.0f s" % (k, time.time() - t0), flush=True)
    g1t = fb1(n, seed + 7)
    out["PiH"] = O.G1.Affine(C.g1_msm_naive(g1t, hx, threads=THREADS)); del g1t              # :284-286, len(hx) = n
    b2 = C.mul_scalar_batch(O.G2_GEN, scalars_u64(m, seed + 8), g2=True, threads=THREADS)
    out["PiB"] = O.G2.Affine(C.g2_msm_naive(b2, w, threads=THREADS)); del b2
    rec = {"what": "snark.GenerateProofs (snark.go:254-289) on gosnark_amd.synth.QuotientPinocchioInstance(n, seed): px = hx Z + rem built "
                   "by oracle/crt_ntt.py (SHA-256 below), so hx = floor(px / Z) is known; the eight sums by the naive loops; affine",
           "n": n, "seed": seed, "px_sha256": sha,
           "generator": "oracle/gen_golden_large.py pinocchio20 (oracle/gs_oracle.c + oracle/crt_ntt.py, %d threads, %.0f s)" % (THREADS, time.time() - t0)}
    for k, v in out.items():
        rec[k] = [[str(v[0][0]), str(v[0][1])], [str(v[1][0]), str(v[1][1])]] if k == "PiB" else [str(v[0]), str(v[1])]
    with open(os.path.join(OUT, "oracle_pinocchio_quotient_2p%d.json" % logn), "w") as f:
        json.dump(rec, f, indent=1)
    print({k: v for k, v in rec.items() if not k.startswith("Pi")})


def pinocchio_golden(logn=16, seed=0x60D2):
    """snark.GenerateProofs (snark.go:254-289) on the instance gosnark_amd.synth.RandomPinocchioInstance(n, seed) defines."""
    n = 1 << logn
    m = n + 1
    t0 = time.time()
    fb1 = lambda cnt, sd: C.mul_scalar_batch(O.G1_GEN, scalars_u64(cnt, sd), threads=THREADS)      # noqa: E731
    names = ("A", "Ap", "Bp", "C", "Cp", "Kp")
    arr = {k: fb1(m, seed + 1 + i) for i, k in enumerate(names)}
    g1t = fb1(n, seed + 7)
    b2 = C.mul_scalar_batch(O.G2_GEN, scalars_u64(m, seed + 8), g2=True, threads=THREADS)
    w = scalars_u64(m, seed + 9)
    w[0] = (1, 0, 0, 0)
    px = scalars_u64(2 * n - 1, seed + 10)
    print("key rebuilt on the CPU in %.0f s" % (time.time() - t0), flush=True)
    hx, _ = C.poly_div_u64(px, zpoly(m - 2))                                                # snark.go:280
    print("hx = px / Z done, %.0f s" % (time.time() - t0), flush=True)
    out = {}
    for k in ("A", "Ap"):                                                                   # i > NPublic = 1 only (:265-268)
        out["Pi" + k] = O.G1.Affine(C.g1_msm_naive(arr[k][2:], w[2:], threads=THREADS))
    for k in ("Bp", "C", "Cp", "Kp"):                                                       # all variables (:270-278)
        out["Pi" + k] = O.G1.Affine(C.g1_msm_naive(arr[k], w, threads=THREADS))
    out["PiB"] = O.G2.Affine(C.g2_msm_naive(b2, w, threads=THREADS))
    out["PiH"] = O.G1.Affine(C.g1_msm_naive(g1t[:hx.shape[0]], hx, threads=THREADS))        # :284-286
    print("MSMs done, %.0f s" % (time.time() - t0), flush=True)
    rec = {"what": "snark.GenerateProofs (snark.go:254-289) on gosnark_amd.synth.RandomPinocchioInstance(n, seed); MSMs by the naive "
                   "loops, hx by schoolbook Div; affine", "n": n, "seed": seed,
           "generator": "oracle/gen_golden_large.py pinocchio (oracle/gs_oracle.c, %d threads, %.0f s)" % (THREADS, time.time() - t0)}
    for k, v in out.items():
        rec[k] = [[str(v[0][0]), str(v[0][1])], [str(v[1][0]), str(v[1][1])]] if k == "PiB" else [str(v[0]), str(v[1])]
    with open(os.path.join(OUT, "oracle_pinocchio_2p%d.json" % logn), "w") as f:
        json.dump(rec, f, indent=1)
    print({k: v for k, v in rec.items() if not k.startswith("Pi")})


if __name__ == "__main__":
    what = sys.argv[1] if len(sys.argv) > 1 else "all"
    if what in ("pinocchio", "all"):
        pinocchio_golden()
    if what in ("msm", "all"):
        msm_golden()
    if what in ("pinocchio20", "all"):
        pinocchio20_golden(int(sys.argv[2]) if len(sys.argv) > 2 else 20)
    if what in ("prove20", "all"):
        prove20_golden(int(sys.argv[2]) if len(sys.argv) > 2 else 20)
    if what in ("partials20", "all"):
        partials_golden()
    if what in ("msm22", "all"):
        msm_golden(logn=22, seed=0x60D6)          # BASELINE configs[3]: the 2^22-term MSM that is sharded over 8 GPUs; ~6 minutes on 8 cores
    if what in ("msm20", "all"):
        msm_golden(logn=20, seed=0x60D3)          # the headline size: ~2 minutes on 8 cores
    if what in ("prove", "all"):
        prove_golden()
