"""Field arithmetic (go-snark-study_amd/csrc/fp29.h) instantiated on the HOST vs Python integers
(= the semantics of the reference's fields/fq.go:32-98).  Same source the kernels compile."""
import random

import pytest

import hostbuild
from oracle import ref_py as O

MODS = {"q": O.Q, "r": O.R}


@pytest.fixture(scope="module")
def exe():
    return hostbuild.build("fp_host_test")


def _vals(rng, p, n):
    edge = [0, 1, 2, p - 1, p - 2, (1 << 253), (1 << 29) - 1, 1 << 29, (1 << 232) - 1, p >> 1]
    return [rng.choice(edge) if rng.random() < 0.25 else rng.randrange(p) for _ in range(n)]


@pytest.mark.parametrize("f", ["q", "r"])
def test_field_ops_match_python(exe, f):
    p = MODS[f]
    rng = random.Random(1234 + ord(f))
    ops = {
        "mul": lambda a, b, c, d: a * b % p,
        "sqr": lambda a, b, c, d: a * a % p,
        "add": lambda a, b, c, d: (a + b) % p,
        "sub": lambda a, b, c, d: (a - b) % p,
        "neg": lambda a, b, c, d: (-a) % p,
        "dbl": lambda a, b, c, d: 2 * a % p,
        "muladd": lambda a, b, c, d: (a * b + c * d) % p,
        "roundtrip": lambda a, b, c, d: a % p,
        "subr": lambda a, b, c, d: (a + b + c - d) % p,
        "reduce2": lambda a, b, c, d: (a + b + c) % p,
        "chain": lambda a, b, c, d: ((2 * (((a + b) - c) * ((a - b) + 2 * d))) ** 2 - a * d) % p,
        "lazy12": lambda a, b, c, d: ((a + b + c + d + a + c) * (a + b + c + d + d)) % p,
    }
    lines, want = [], []
    for op, fn in ops.items():
        for _ in range(300):
            a, b, c, d = _vals(rng, p, 4)
            lines.append("%s %s %x %x %x %x" % (f, op, a, b, c, d))
            want.append(fn(a, b, c, d))
    got = hostbuild.run_lines(exe, lines)
    for line, g, w in zip(lines, got, want):
        assert int(g, 16) == w, line


@pytest.mark.parametrize("f", ["q", "r"])
def test_noncanonical_inputs_are_reduced(exe, f):
    p = MODS[f]
    rng = random.Random(99)
    lines, want = [], []
    for _ in range(200):
        a, b = rng.randrange(1 << 256), rng.randrange(1 << 256)
        lines.append("%s mul %x %x 0 0" % (f, a, b))
        want.append(a * b % p)
    got = hostbuild.run_lines(exe, lines)
    assert [int(g, 16) for g in got] == want


@pytest.mark.parametrize("f", ["q", "r"])
def test_inverse_fermat(exe, f):
    p = MODS[f]
    rng = random.Random(7)
    vals = [1, 2, p - 1] + [rng.randrange(1, p) for _ in range(40)]
    got = hostbuild.run_lines(exe, ["%s inv %x 0 0 0" % (f, a) for a in vals])
    for a, g in zip(vals, got):
        assert int(g, 16) == pow(a, -1, p)
    assert int(hostbuild.run_lines(exe, ["%s inv 0 0 0 0" % f])[0], 16) == 0


@pytest.mark.parametrize("f", ["q", "r"])
def test_is_zero_exact(exe, f):
    p = MODS[f]
    rng = random.Random(5)
    lines, want = [], []
    for _ in range(300):
        a = rng.randrange(p)
        k = rng.random()
        b = a if k < 0.4 else rng.randrange(p)
        c = (a + b) % p if rng.random() < 0.5 else rng.randrange(p)
        lines.append("%s iszero %x %x %x 0" % (f, a, b, c))
        want.append(("1" if a == b else "0") + ("1" if (a + b - c) % p == 0 else "0"))
    assert hostbuild.run_lines(exe, lines) == want


@pytest.mark.parametrize("f", ["q", "r"])
def test_is_zero_on_larger_multiples_of_p(exe, f):
    """is_zero = cheap limb-0 prefilter (maybe_zero: two candidates for k in value = k p) + exact check: values that vanish as
    k p for k = 3..9 in nearly-normal form, values that do not, and the prefilter never says no to a true zero."""
    p = MODS[f]
    rng = random.Random(6)
    lines, want = [], []
    for _ in range(400):
        a = rng.choice([0, 1, p - 1, p >> 1, (p >> 2) + 1]) if rng.random() < 0.2 else rng.randrange(p)
        b = 4 * a % p if rng.random() < 0.5 else rng.randrange(p)
        c = rng.randrange(p)
        d = (4 * a - b + c) % p if rng.random() < 0.5 else rng.randrange(p)
        lines.append("%s iszero4 %x %x %x %x" % (f, a, b, c, d))
        want.append(("1" if (4 * a - b) % p == 0 else "0") + ("1" if (4 * a - b + c - d) % p == 0 else "0") + "1")
    assert hostbuild.run_lines(exe, lines) == want


def test_host_copy_pool_copies_every_byte():
    """go-snark-study_amd/csrc/hostcopy.h: the persistent host threads that fill the pinned staging buffers of uploads from
    pageable memory -- 6000 back-to-back jobs around the threshold and up to 8 MiB, odd offsets; a late worker of job g - 1 must
    neither take nor skip a piece of job g."""
    import subprocess
    exe = hostbuild.build("hostcopy_test")
    out = subprocess.run([exe, "6000"], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and out.stdout.startswith("OK"), out.stdout + out.stderr


def test_context_lock_is_first_come_first_served():
    """go-snark-study_amd/csrc/runtime.h FairMutex (round 6): try_lock, service in the order of arrival, mutual exclusion, and no starvation of
    polite threads by two threads that release the lock and ask again at once -- the std::mutex of rounds 1-5 let two blocking-proof loops keep
    six other threads of tests/c/stream_stress.c at one operation each for a minute (profiles/r06_stream_stress_fair_lock.txt)."""
    import subprocess
    exe = hostbuild.build("fair_mutex_test")
    out = subprocess.run([exe, "400"], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0 and out.stdout.strip().endswith("OK"), out.stdout + out.stderr
