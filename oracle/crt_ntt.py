"""ORACLE TOOLING (test infrastructure, not product code): exact products of large polynomials over Fr by a route that shares
nothing with the library's NTT engine -- eighteen 31-bit NTT primes in numpy int64, Garner's mixed-radix reconstruction, reduction
mod r at the very end.  The library multiplies in Fr itself (one 254-bit Montgomery NTT, 9 x 29-bit limbs); this file never sees r
until the last line.  Used by oracle/gen_golden_large.py to build golden inputs at sizes where the reference's schoolbook Mul / Div
(r1csqap.go:57-84) would take days; checked against that schoolbook code (oracle/gs_oracle.c) at small sizes by
tests/test_oracle_crt_ntt.py."""
import numpy as np

R = 21888242871839275222246405745257275088548364400416034343698204186575808495617
LOG_MAX = 22                                   # transforms up to 2^22 points
NPRIMES = 18                                   # 18 x ~30.9 bits > 2^528 >= 2^20 * r^2


def _is_prime(n):
    if n < 2:
        return False
    for q in (2, 3, 5, 7, 11, 13, 17, 19, 23, 29, 31, 37):
        if n % q == 0:
            return n == q
    d, s = n - 1, 0
    while d % 2 == 0:
        d //= 2
        s += 1
    for a in (2, 3, 5, 7):                      # deterministic below 3.2e9
        x = pow(a, d, n)
        if x in (1, n - 1):
            continue
        for _ in range(s - 1):
            x = x * x % n
            if x == n - 1:
                break
        else:
            return False
    return True


def _generator(p):
    fac, m, q = [], p - 1, 2
    while q * q <= m:
        if m % q == 0:
            fac.append(q)
            while m % q == 0:
                m //= q
        q += 1
    if m > 1:
        fac.append(m)
    g = 2
    while any(pow(g, (p - 1) // f, p) == 1 for f in fac):
        g += 1
    return g


def _primes():
    out, k = [], (1 << 31) >> LOG_MAX
    while len(out) < NPRIMES:
        k -= 1
        p = (k << LOG_MAX) + 1
        if _is_prime(p):
            out.append(p)
    return out


PRIMES = _primes()
_GEN = {p: _generator(p) for p in PRIMES}
_TW = {}


def _twiddles(p, logn, inverse):
    key = (p, logn, inverse)
    if key not in _TW:
        n = 1 << logn
        w = pow(_GEN[p], (p - 1) // n, p)
        if inverse:
            w = pow(w, p - 2, p)
        tw = np.ones(n // 2, dtype=np.int64)
        # powers of w by doubling: tw[i] = w^i
        cur, filled = w, 1
        while filled < n // 2:
            tw[filled:2 * filled] = tw[:filled] * cur % p
            cur = cur * cur % p
            filled *= 2
        _TW[key] = tw
    return _TW[key]


def _bitrev(logn):
    n = 1 << logn
    idx = np.arange(n, dtype=np.int64)
    rev = np.zeros(n, dtype=np.int64)
    for b in range(logn):
        rev |= ((idx >> b) & 1) << (logn - 1 - b)
    return rev


_REV = {}


def _ntt(a, p, logn, inverse=False):
    """In-order in, in-order out; decimation in time on the bit-reversed input."""
    n = 1 << logn
    if logn not in _REV:
        _REV[logn] = _bitrev(logn)
    a = a[_REV[logn]]
    tw = _twiddles(p, logn, inverse)
    h = 1
    while h < n:
        v = a.reshape(n // (2 * h), 2, h)
        t = v[:, 1, :] * tw[::n // (2 * h)][None, :] % p
        u = v[:, 0, :].copy()
        v[:, 0, :] = (u + t) % p
        v[:, 1, :] = (u - t) % p
        h *= 2
    if inverse:
        a = a * pow(n, p - 2, p) % p
    return a


def _residues(limbs, p):
    """[n, 4] uint64 little-endian limbs -> the values mod p as int64."""
    acc = np.zeros(limbs.shape[0], dtype=np.int64)
    base = (1 << 64) % p
    for k in (3, 2, 1, 0):
        acc = (acc * base + (limbs[:, k] % np.uint64(p)).astype(np.int64)) % p
    return acc


def poly_mul_mod_r(a_limbs, b_limbs):
    """a, b: [na, 4] / [nb, 4] uint64 limbs of coefficients < r.  Returns the na + nb - 1 coefficients of a * b mod r, same layout."""
    a_limbs = np.ascontiguousarray(a_limbs, dtype=np.uint64).reshape(-1, 4)
    b_limbs = np.ascontiguousarray(b_limbs, dtype=np.uint64).reshape(-1, 4)
    na, nb = a_limbs.shape[0], b_limbs.shape[0]
    nout = na + nb - 1
    logn = max(1, (nout - 1).bit_length())
    assert logn <= LOG_MAX and min(na, nb) <= (1 << 20), "product too large for the prime set"
    n = 1 << logn
    res = []
    for p in PRIMES:
        fa = np.zeros(n, dtype=np.int64)
        fb = np.zeros(n, dtype=np.int64)
        fa[:na] = _residues(a_limbs, p)
        fb[:nb] = _residues(b_limbs, p)
        fc = _ntt(fa, p, logn) * _ntt(fb, p, logn) % p
        res.append(_ntt(fc, p, logn, inverse=True)[:nout])
    # Garner: x = c0 + c1 p0 + c2 p0 p1 + ... with 0 <= cj < pj, all in small modular arithmetic
    digits = []
    for j, p in enumerate(PRIMES):
        t = res[j].copy()
        # subtract the part already known, evaluated mod p by Horner over the mixed radix
        if j:
            known = digits[j - 1] % p
            for i in range(j - 2, -1, -1):
                known = (known * (PRIMES[i] % p) + digits[i]) % p
            minv = 1
            for i in range(j):
                minv = minv * PRIMES[i] % p
            t = (t - known) % p * pow(minv, p - 2, p) % p
        digits.append(t)
    # x mod r = sum_j cj * (p0 ... p_{j-1} mod r) mod r, accumulated on 16-bit limbs of the constants (47 + 5 bits per term)
    acc = np.zeros((nout, 17), dtype=np.int64)
    m = 1
    for j, p in enumerate(PRIMES):
        mr = m % R
        for k in range(16):
            acc[:, k] += digits[j] * ((mr >> (16 * k)) & 0xFFFF)
        m *= p
    # carry to clean 16-bit limbs (values < 2^(254 + 31 + 5)), then to Python ints for the final reduction
    carry = np.zeros(nout, dtype=np.int64)
    limbs16 = np.zeros((nout, 20), dtype=np.uint16)
    for k in range(20):
        v = (acc[:, k] if k < 17 else 0) + carry
        limbs16[:, k] = (v & 0xFFFF).astype(np.uint16)
        carry = v >> 16
    assert not carry.any()
    raw = limbs16.tobytes()
    out = np.zeros((nout, 4), dtype=np.uint64)
    vals = [int.from_bytes(raw[40 * i:40 * i + 40], "little") % R for i in range(nout)]
    out[:] = np.frombuffer(b"".join(v.to_bytes(32, "little") for v in vals), dtype="<u8").reshape(nout, 4)
    return out
