"""CPU-only checks of the host-side logic around the C ABI (no GPU, no compute calls): limb packing, CSR builders,
the sqchain synthetic circuit of SURVEY.md 8d and the reference-shaped containers."""
import os

import numpy as np

import gosnark_amd  # noqa: F401
from gosnark_amd import capi, r1csqap, synth
from oracle import ref_py as O


def test_limb_packing_roundtrip_matches_big_int_bits():
    vals = [0, 1, O.R - 1, O.Q - 1, (1 << 256) - 1, 0x0123456789ABCDEF0FEDCBA987654321]
    arr = capi.ints_to_u64(vals)
    assert arr.shape == (len(vals), 4) and arr.dtype == np.uint64
    assert capi.u64_to_ints(arr) == vals
    # little-endian 64-bit words, exactly big.Int.Bits() padded to 4
    assert [int(x) for x in arr[5]] == [0x0FEDCBA987654321, 0x0123456789ABCDEF, 0, 0]
    g1 = capi.g1_points_to_u64([(1, 2, 3)])
    assert g1.shape == (1, 12) and int(g1[0, 0]) == 1 and int(g1[0, 4]) == 2 and int(g1[0, 8]) == 3
    g2 = capi.g2_points_to_u64([((1, 2), (3, 4), (5, 6))])
    assert g2.shape == (1, 24) and [int(g2[0, 4 * i]) for i in range(6)] == [1, 2, 3, 4, 5, 6]


def test_csr_from_rows():
    rp, cl, vl = r1csqap.csr_from_rows([{2: 5, 0: 1}, {}, {1: -1}])
    assert list(rp) == [0, 2, 2, 3] and list(cl) == [0, 2, 1]
    assert capi.u64_to_ints(vl) == [1, 5, O.R - 1]


def _dense(csr, n, m):
    rp, cl, vl = csr
    vals = capi.u64_to_ints(vl)
    mat = [[0] * m for _ in range(n)]
    for j in range(n):
        for e in range(int(rp[j]), int(rp[j + 1])):
            mat[j][int(cl[e])] = vals[e]
    return mat


def test_sqchain_circuit_is_satisfied_and_shaped_like_the_survey_says():
    for n in (1, 2, 8, 24):
        a, b, c, w = synth.sqchain_r1cs(n, 987654321)
        m = n + 1
        wi = capi.u64_to_ints(w)
        assert len(wi) == m and wi[0] == 1 and wi[1] == 987654321
        A, B, C = (_dense(x, n, m) for x in (a, b, c))
        dot = lambda row: sum(x * y for x, y in zip(row, wi)) % O.R   # noqa: E731
        for j in range(n):
            assert dot(A[j]) * dot(B[j]) % O.R == dot(C[j]), (n, j)
        assert int(a[0][n]) == n and int(b[0][n]) == n and int(c[0][n]) == 2 * n - 1      # nnz: A = n, B = n, C = 2n - 1
        if n >= 2:
            assert wi[2] == (wi[1] * wi[1] + 1) % O.R          # s_2 = s_1^2 + 1
        # the reference's dense flow accepts it (n <= 21: before NewPolZeroAt's int overflow): px / Z leaves no remainder
        if 2 <= n <= 8:
            al, be, ga, z = O.PF.R1CSToQAP(A, B, C)
            _, _, _, px = O.PF.CombinePolynomials(wi, al, be, ga)
            hx, rem = O.PF.Div(px, z)
            assert all(v == 0 for v in rem) and len(px) == 2 * n - 1 and len(z) == m - 1 and len(hx) == n


def test_reference_shaped_containers():
    from gosnark_amd import groth16, snark
    circ = groth16.Circuit(8, 1)
    assert (circ.NVars, circ.NPublic) == (8, 1)
    pk = groth16.Pk(BACDelta=[], Z=[1], G1_Alpha=(0, 0, 0), G1_Beta=(0, 0, 0), G1_Delta=(0, 0, 0), G1_At=[], G1_BACGamma=[],
                    G2_Beta=None, G2_Delta=None, G2_BACGamma=[], PowersTauDelta=[])
    assert pk._dev is None
    assert snark.Proof.FIELDS == ("PiA", "PiAp", "PiB", "PiBp", "PiC", "PiCp", "PiH", "PiKp")     # snark.go:59-69
    r = groth16.FqRRand()
    assert 0 <= r < O.R and r < (1 << 240)                       # 30 random bytes (fields/fq.go:116-132)


def test_pairing_header_constants_match_their_definitions():
    """csrc/pairing.h: the word constants equal tools/gen_constants.pairing_constants(), which also asserts that the
    final exponentiation's addition chain is exactly (q^4 - q^2 + 1)/r (so the value equals Fq12.Exp(f, FinalExp))."""
    import importlib.util
    import re
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("gen_constants", os.path.join(root, "tools", "gen_constants.py"))
    gen = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(gen)
    want = gen.pairing_constants()
    text = open(os.path.join(root, "go-snark-study_amd", "csrc", "pairing.h")).read()
    for name, words in want.items():
        m = re.search(r"static const uint64_t %s(?:\[\d+\])? = \{?([^;]*?)\}?;" % name, text)
        assert m, name
        got = [int(x, 16) for x in re.findall(r"0x([0-9a-fA-F]+)ULL", m.group(1))]
        assert got == words, name


def test_shard_split_is_the_same_everywhere():
    """The contiguous split (first n % N ranges one longer) is written three times: parallel.shard_range (exchange layer),
    groth16._shard_range (key slices, binary key files) and shard_range in csrc/prove.hip (gs_groth16_prove_partials, checked on
    the GPU by the slice tests).  They must agree or a key slice would not match its rank's term range."""
    from gosnark_amd import groth16, parallel
    for n in (0, 1, 7, 8, 9, 4098, (1 << 20) + 1):
        for world in (1, 2, 3, 8):
            ranges = [parallel.shard_range(n, world, r) for r in range(world)]
            assert ranges == [groth16._shard_range(n, world, r) for r in range(world)]
            assert ranges[0][0] == 0 and ranges[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(ranges, ranges[1:]))
            sizes = [hi - lo for lo, hi in ranges]
            assert max(sizes) - min(sizes) <= 1 and sizes == sorted(sizes, reverse=True)


def test_go_layer_matches_the_c_abi_mechanically(tmp_path):
    """tools/check_go_abi.py: every C.gs_* call of go/**/*.go has the arity and the per-position kind (handle /
    pointer / scalar type) of its prototype in include/gosnark_hip.h, every prototype is bound, and every INTEGRATION.md row's C driver
    calls the entry points its Go functions reach.  The checker must also FAIL when an argument is dropped or two are swapped."""
    import shutil
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import check_go_abi as chk
    errs, stats = chk.check()
    assert errs == [], errs
    assert stats["prototypes"] == stats["bound"] >= 124 and stats["go_calls"] >= stats["prototypes"]
    assert stats["unresolved_arguments"] == 0 and stats["integration_rows_checked"] >= 10
    shutil.copytree(chk.GO_DIR, tmp_path / "go")
    seams = tmp_path / "go" / "gosnarkhip" / "seams.go"
    good = seams.read_text()
    call = "C.gs_zpoly(C.size_t(deg), ptr(out))"
    assert call in good
    saved = chk.GO_DIR
    try:
        chk.GO_DIR = str(tmp_path / "go")
        seams.write_text(good.replace(call, "C.gs_zpoly(ptr(out))", 1))
        assert any("called with 1 arguments" in e for e in chk.check()[0])
        seams.write_text(good.replace(call, "C.gs_zpoly(ptr(out), C.size_t(deg))", 1))
        assert any("argument 1" in e and "wants a scalar" in e for e in chk.check()[0])
        seams.write_text(good.replace(call, "C.gs_zpoly(C.int(deg), ptr(out))", 1))
        assert any("is C.int, the header wants size_t" in e for e in chk.check()[0])
    finally:
        chk.GO_DIR = saved


def test_environment_knobs_are_validated_or_compiled_out(tmp_path):
    """go-snark-study_amd/csrc/knobs.h: in the product build (no -DGS_DEV_KNOBS) the tuning variables are not read
    at all; in a development build a value the algorithm cannot take (GS_REDUCE_L=3: not a power of two, GS_CHUNK=7: not a multiple
    of 4, out of range, not a number) falls back to the default; the always-on switches parse strictly.  And no source file of the
    library calls getenv() itself."""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    csrc = os.path.join(root, "go-snark-study_amd", "csrc")
    for f in os.listdir(csrc):
        if f.endswith((".hip", ".h")) and f != "knobs.h":
            assert "getenv" not in open(os.path.join(csrc, f)).read(), f
    src = os.path.join(root, "tests", "host", "knobs_host_test.cpp")
    lines = ["dev GS_REDUCE_L 4 1 32 1 1", "dev GS_CHUNK 0 4 1024 4 0", "dev GS_FOLD_MAX 0 1 256 1 0", "dev GS_AUTO_MAX_C 17 8 20 1 0",
             "run GS_TAIL_FLIP 1 0 2 1 0", "run GS_COPY_THREADS 4 1 64 1 0", "devflag GS_TABLE_PER_ROW 0 0 0 1 0"]
    hostile = {"GS_REDUCE_L": "3", "GS_CHUNK": "7", "GS_FOLD_MAX": "-5", "GS_AUTO_MAX_C": "99", "GS_TAIL_FLIP": "77", "GS_COPY_THREADS": "4x",
               "GS_TABLE_PER_ROW": "1"}
    benign = {"GS_REDUCE_L": "8", "GS_CHUNK": "48", "GS_FOLD_MAX": "16", "GS_AUTO_MAX_C": "19", "GS_TAIL_FLIP": "2", "GS_COPY_THREADS": "6",
              "GS_TABLE_PER_ROW": "1"}
    defaults = ["4", "0", "0", "17", "1", "4", "0"]
    for flags, env, want in (([], hostile, defaults), ([], benign, defaults[:4] + ["2", "6", "0"]),                       # product build
                             (["-DGS_DEV_KNOBS"], hostile, defaults[:6] + ["1"]), (["-DGS_DEV_KNOBS"], benign, ["8", "48", "16", "19", "2", "6", "1"])):
        exe = tmp_path / ("knobs" + ("_dev" if flags else ""))
        if not exe.exists():
            subprocess.check_call(["g++", "-std=c++17", "-O1", "-I", csrc] + flags + [src, "-o", str(exe)])
        out = subprocess.run([str(exe)], input="\n".join(lines) + "\n", capture_output=True, text=True, check=True,
                             env=dict(os.environ, **env)).stdout.split()
        assert out == want, (flags, env, out)


def test_heavy_bucket_slices_partition_every_span():
    """msm_kernels.h heavy_slice, compiled for the host (tests/host/heavy_slice_test.hip): the 16 slices of a heavy bucket are
    consecutive, disjoint and cover its chunks exactly once for every span from kHeavySpan + 2 chunks up."""
    import subprocess
    import hostbuild
    exe = hostbuild.build("heavy_slice_test")
    out = subprocess.run([exe], capture_output=True, text=True, check=True).stdout
    assert out.startswith("OK"), out
