#!/usr/bin/env python
"""What would it cost to DERIVE the evaluation-basis array of a foreign Groth16 key?

A key loaded from a file without the evaluation-basis section proves witnesses through H's coefficients (17.7 ms at 2^20 instead of
11.3).  The array is E[j-1] = l_j(tau) Z(tau)/delta G for the Lagrange basis l_j over the nodes n+1 .. 2n, and PowersTauDelta[i] =
tau^i Z(tau)/delta G (groth16.go:139-149), so  E[j-1] = sum_i coeff_i(l_j) PowersTauDelta[i]:  n MSMs of n terms over ONE base
array (its window table is built once).  l_j(x) = M(x) / ((x - node_j) M'(node_j)) with M = prod_k (x - node_k) = Z_2n / Z_n: the
row of scalars is a synthetic division away.  This prototype does exactly that through the product's own entry points (gs_zpoly,
gs_poly_div, gs_poly_eval, gs_msm_g1) -- O(n^2 W) bucket additions instead of the O(n log^2 n) scalar multiplications of a
transposed subproduct tree in the group -- checks the rows it derives against the array the device setup emitted, and times them:

    python tools/derive_eval_basis.py --log2n 10             every row (full check)
    python tools/derive_eval_basis.py --log2n 16 --rows 24   a sample of rows; the whole derivation is extrapolated from them

The per-row figure is split into the scalar row (host-driven polynomial division here; a synthetic-division kernel in a real
implementation) and the MSM, which is the part that cannot shrink."""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import gosnark_amd  # noqa: F401,E402
from gosnark_amd import capi, groth16, r1csqap, synth  # noqa: E402

R = groth16.R


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--log2n", type=int, default=10)
    ap.add_argument("--rows", type=int, default=0, help="rows to derive (0 = all n)")
    args = ap.parse_args()
    n = 1 << args.log2n
    capi.init(0)
    inst = synth.sqchain_setup_instance(n, 0xE7A1 + args.log2n)
    pk = inst.device_pk()
    ptd = groth16.ExportPkArray(pk, "PowersTauDelta")
    want = groth16.ExportPkArray(pk, "PowersTauDeltaEval")
    assert len(ptd) == n and len(want) == n
    bases = capi.g1_upload(capi.ints_to_u64([c for p in ptd for c in p]).reshape(-1, 12))
    pf = r1csqap.PolynomialField()
    t0 = time.perf_counter()
    M, rem = pf.Div(r1csqap.ZPoly(2 * n), r1csqap.ZPoly(n))            # prod_{k=1..n} (x - (n + k))
    assert not any(rem) and len(M) == n + 1 and M[n] == 1
    t_m = time.perf_counter() - t0
    rows = list(range(1, n + 1)) if not args.rows else sorted({1, n} | {1 + (k * 7919) % n for k in range(args.rows - 2)})
    capi.msm(bases, capi.ints_to_u64([1] * n))                          # window table of the base array
    t_row = t_msm = t_res = 0.0
    for j in rows:
        node = n + j
        t0 = time.perf_counter()
        q, r0 = pf.Div(M, [(-node) % R, 1])                              # M / (x - node_j): n coefficients
        assert not any(r0)
        dinv = pow(pf.Eval(q, node), R - 2, R)                           # 1 / M'(node_j)
        sc = capi.ints_to_u64([c * dinv % R for c in q])
        t1 = time.perf_counter()
        got = capi.msm(bases, sc)                                        # host-buffer entry point: 32 n bytes cross PCIe
        t2 = time.perf_counter()
        sh = capi.scalars_upload(sc)
        capi.msm_resident(bases, sh, n)
        t3 = time.perf_counter()
        capi.msm_resident(bases, sh, n)
        t4 = time.perf_counter()
        sh.free()
        t_row += t1 - t0
        t_msm += t2 - t1
        t_res += t4 - t3
        w = want[j - 1]
        assert got == (w[0], w[1]), "row %d differs from the array the setup emitted" % j
    k = len(rows)
    out = {"n": n, "rows_derived": k, "all_rows_equal_the_setup_array": True, "M_poly_s": t_m,
           "scalar_row_ms_host_driven": t_row / k * 1e3, "msm_ms_per_row_host_buffers": t_msm / k * 1e3, "msm_ms_per_row_resident_blocking": t_res / k * 1e3,
           "whole_derivation_extrapolated_s": {"msm_only_resident": t_res / k * n, "as_measured_here": (t_row + t_msm) / k * n + t_m,
                                               "row_scalars_through_hbm_at_4TBs": 2.0 * 32 * n * n / 4e12}}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
