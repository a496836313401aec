"""Shared helpers for the -m gpu parity tests (test infrastructure)."""
import random

import numpy as np

from oracle import ref_py as O


def splitmix64(seed):
    """SplitMix64 stream (deterministic synthetic inputs)."""
    x = seed & 0xFFFFFFFFFFFFFFFF
    while True:
        x = (x + 0x9E3779B97F4A7C15) & 0xFFFFFFFFFFFFFFFF
        z = x
        z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & 0xFFFFFFFFFFFFFFFF
        z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & 0xFFFFFFFFFFFFFFFF
        yield z ^ (z >> 31)


def rand_scalars_u64(n, seed):
    """n uniform elements of [0, r) as [n,4] uint64 (numpy, vectorised rejection)."""
    rng = np.random.Generator(np.random.PCG64(seed))
    out = np.zeros((n, 4), dtype=np.uint64)
    todo = np.arange(n)
    r_limbs = [(O.R >> (64 * i)) & 0xFFFFFFFFFFFFFFFF for i in range(4)]
    while todo.size:
        cand = rng.integers(0, 2**64, size=(todo.size, 4), dtype=np.uint64)
        cand[:, 3] &= np.uint64(0x3FFFFFFFFFFFFFFF)
        # keep candidates < r (lexicographic compare from the top limb)
        lt = np.zeros(todo.size, dtype=bool)
        eq = np.ones(todo.size, dtype=bool)
        for i in (3, 2, 1, 0):
            lt |= eq & (cand[:, i] < np.uint64(r_limbs[i]))
            eq &= cand[:, i] == np.uint64(r_limbs[i])
        out[todo[lt]] = cand[lt]
        todo = todo[~lt]
    return out


def u64_rows_to_ints(a):
    raw = np.ascontiguousarray(a, dtype="<u8").tobytes()
    return [int.from_bytes(raw[i:i + 32], "little") for i in range(0, len(raw), 32)]


def rand_g1_jac(rng, inf_prob=0.0):
    if rng.random() < inf_prob:
        return O.G1_ZERO
    return O.G1.MulScalar(O.G1_GEN, rng.randrange(1, O.R))


def rand_g2_jac(rng, inf_prob=0.0):
    if rng.random() < inf_prob:
        return O.G2_ZERO
    return O.G2.MulScalar(O.G2_GEN, rng.randrange(1, O.R))


def ref_msm_affine(G, pts, ks):
    """complete (mathematically correct) reference sum in affine form, small n"""
    z = O.G1_ZERO if G is O.G1 else O.G2_ZERO
    acc = z
    for p, k in zip(pts, ks):
        t = G.MulScalar(p, k % O.R)
        if not G.IsZero(acc) and not G.IsZero(t) and G.Equal(acc, t):
            acc = G.Double(acc)
        else:
            acc = G.Add(acc, t)
    return G.Affine(acc)
