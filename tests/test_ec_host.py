"""G1/G2 group arithmetic (go-snark-study_amd/csrc/ec.h) instantiated on the HOST vs the oracle's
literal restatement of bn128/g1.go / g2.go, compared in affine normal form."""
import random

import pytest

import hostbuild
from oracle import ref_py as O


@pytest.fixture(scope="module")
def exe():
    return hostbuild.build("ec_host_test")


def rand_pt(rng, G, gen, inf_prob=0.1):
    if rng.random() < inf_prob:
        return (0, 0, 0) if G is O.G1 else ((0, 0), (0, 0), (0, 0))
    return G.MulScalar(gen, rng.randrange(1, O.R))


def fmt(G, p, k, sign):
    if G is O.G1:
        cs = "%x %x %x" % p
    else:
        cs = " ".join("%x %x" % c for c in p)
    return "%s %x %d" % (cs, k, sign)


def parse(G, line):
    if line.strip() == "inf":
        return None
    v = [int(x, 16) for x in line.split()]
    return (v[0], v[1]) if G is O.G1 else ((v[0], v[1]), (v[2], v[3]))


def ref_sum(G, terms):
    z = (0, 0, 0) if G is O.G1 else ((0, 0), (0, 0), (0, 0))
    acc = z
    for p, k, sign in terms:
        t = G.MulScalar(p, k)
        if sign:
            t = G.Neg(t)
        # the reference Add has no doubling branch: use Double when equal
        if not G.IsZero(acc) and not G.IsZero(t) and G.Equal(acc, t):
            acc = G.Double(acc)
        else:
            acc = G.Add(acc, t)
    return G.Affine(acc)


def ask(exe, g, op, terms, G):
    lines = ["%s %s %d" % (g, op, len(terms))] + [fmt(G, p, k, s) for p, k, s in terms]
    return parse(G, hostbuild.run_lines(exe, lines)[0])


@pytest.mark.parametrize("g", ["g1", "g2"])
def test_lincomb_full_width_scalars(exe, g):
    G, gen = (O.G1, O.G1_GEN) if g == "g1" else (O.G2, O.G2_GEN)
    rng = random.Random(11)
    for n in (1, 2, 5):
        terms = [(rand_pt(rng, G, gen), rng.randrange(O.R), rng.randrange(2)) for _ in range(n)]
        assert ask(exe, g, "lincomb", terms, G) == ref_sum(G, terms)


@pytest.mark.parametrize("g", ["g1", "g2"])
def test_madd_chain_with_negation_and_infinities(exe, g):
    G, gen = (O.G1, O.G1_GEN) if g == "g1" else (O.G2, O.G2_GEN)
    rng = random.Random(12)
    terms = [(rand_pt(rng, G, gen, 0.2), 1, rng.randrange(2)) for _ in range(24)]
    assert ask(exe, g, "maddsum", terms, G) == ref_sum(G, terms)


@pytest.mark.parametrize("g", ["g1", "g2"])
def test_complete_addition_doubling_and_cancellation(exe, g):
    """Cases the reference's Add gets wrong (g1.go:32-89 has no P==Q branch): P+P, P+(-P),
    and the same point in two different Jacobian representations."""
    G, gen = (O.G1, O.G1_GEN) if g == "g1" else (O.G2, O.G2_GEN)
    p = G.MulScalar(gen, 123456789)
    p_other_repr = G.Add(G.MulScalar(gen, 123456000), G.MulScalar(gen, 789))   # same point, different Z
    two_p = G.Affine(G.Double(p))
    assert ask(exe, g, "maddsum", [(p, 1, 0), (p, 1, 0)], G) == two_p
    assert ask(exe, g, "maddsum", [(p, 1, 0), (p_other_repr, 1, 0)], G) == two_p
    assert ask(exe, g, "maddsum", [(p, 1, 0), (p_other_repr, 1, 1)], G) is None
    assert ask(exe, g, "lincomb", [(p, 5, 0), (p_other_repr, 5, 0)], G) == G.Affine(G.MulScalar(gen, 1234567890))
    assert ask(exe, g, "lincomb", [(p, 5, 0), (p_other_repr, 5, 1)], G) is None
    three = ask(exe, g, "maddsum", [(p, 1, 0), (p, 1, 0), (p_other_repr, 1, 0)], G)
    assert three == G.Affine(G.MulScalar(p, 3))
    assert ask(exe, g, "maddsum", [], G) is None


@pytest.mark.parametrize("g", ["g1", "g2"])
def test_accumulation_kernel_accumulator_type(exe, g):
    """k_bucket_accumulate's own accumulator (G2: ec.h XyzzAcc, y typed below 2p so that R needs no reduction; G1: the plain Xyzz)
    through the same chains: random points with negations and infinities, then doubling (whose y must come back below 2p),
    cancellation, restart from infinity and further additions on top of a doubled point."""
    G, gen = (O.G1, O.G1_GEN) if g == "g1" else (O.G2, O.G2_GEN)
    rng = random.Random(14)
    terms = [(rand_pt(rng, G, gen, 0.2), 1, rng.randrange(2)) for _ in range(40)]
    assert ask(exe, g, "maddacc", terms, G) == ref_sum(G, terms)
    p = G.MulScalar(gen, 987654321)
    p2 = G.Add(G.MulScalar(gen, 987654000), G.MulScalar(gen, 321))            # same point, different Z
    q = G.MulScalar(gen, 55555)
    assert ask(exe, g, "maddacc", [(p, 1, 0), (p2, 1, 0)], G) == G.Affine(G.Double(p))
    assert ask(exe, g, "maddacc", [(p, 1, 0), (p2, 1, 1)], G) is None
    chain = [(p, 1, 0), (p2, 1, 0), (q, 1, 1), (p, 1, 0), (q, 1, 0), (p2, 1, 1), (p, 1, 1), (p2, 1, 1), (q, 1, 0), (q, 1, 0)]
    assert ask(exe, g, "maddacc", chain, G) == ref_sum(G, chain)
    assert ask(exe, g, "maddacc", [], G) is None


@pytest.mark.parametrize("g", ["g1", "g2"])
def test_small_scalar_mul(exe, g):
    G, gen = (O.G1, O.G1_GEN) if g == "g1" else (O.G2, O.G2_GEN)
    rng = random.Random(13)
    terms = [(rand_pt(rng, G, gen, 0.0), k, 0) for k in (0, 1, 2, 3, 65535, 32768, 0xffffffff)]
    assert ask(exe, g, "small", terms, G) == ref_sum(G, terms)


@pytest.mark.parametrize("g", ["g1", "g2"])
def test_jacobian_doubling_chain_of_the_table_builder(exe, g):
    """Round 6: the window-table builder walks every point through 14 x 17 Jacobian doublings (ec.h jac_dbl: 3M + 4S on a = 0, bounds kept
    where the next doubling squares them) instead of XYZZ ones.  2^k P for k = 0, 1, 2, 17, 34, 238, 255 doublings of random points (also
    negated, also the generator), summed, against the oracle's MulScalar."""
    G, gen = (O.G1, O.G1_GEN) if g == "g1" else (O.G2, O.G2_GEN)
    rng = random.Random(15)
    terms = [(rand_pt(rng, G, gen, 0.0), k, rng.randrange(2)) for k in (0, 1, 2, 17, 34, 238, 255)] + [(gen, 17, 0), (gen, 238, 1)]
    want = ref_sum(G, [(p, 1 << k, s) for p, k, s in terms])
    assert ask(exe, g, "jacdbl", terms, G) == want
    assert ask(exe, g, "jacdbl", [(rand_pt(rng, G, gen, 1.0), 5, 0)], G) is None          # infinity in, nothing out


def test_g1_77G_reference_kat(exe):
    """bn128/g1_test.go:29-30"""
    got = ask(exe, "g1", "lincomb", [(O.G1_GEN, 33, 0), (O.G1_GEN, 44, 0)], O.G1)
    assert got == (0x2f978c0ab89ebaa576866706b14787f360c4d6c3869efe5a72f7c3651a72ff00,
                   0x12e4ba7f0edca8b4fa668fe153aebd908d322dc26ad964d4cd314795844b62b2)
