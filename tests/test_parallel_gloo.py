"""Multi-process (gloo, world_size 2, CPU) test of the term-sharded MSM exchange (go-snark-study_amd/parallel.py):
every rank sums its shard with the oracle (standing in for its GPU), the partial points are all-gathered and
combined with the library's host-side complete addition (gs_g1_sum_affine / gs_g2_sum_affine need no device), and
the result must equal the oracle's MSM over the whole range."""
import os
import socket
import sys

import numpy as np
import pytest
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, n, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import gosnark_amd  # noqa: F401
        from gosnark_amd import parallel
        import gpu_util as U
        from oracle import c_oracle as C
        from oracle import ref_py as O
        # identical seeded inputs on every rank; each only touches its shard
        ks = U.rand_scalars_u64(n, 5)
        ks[3] = 0
        pts1 = np.zeros((n, 12), dtype=np.uint64)
        pts2 = np.zeros((n, 24), dtype=np.uint64)
        for i in range(n):
            pts1[i] = C._u64(C.g1_mul_scalar(O.G1_GEN, 1000 + 17 * i))
            p2 = C.g2_mul_scalar(O.G2_GEN, 2000 + 13 * i)
            pts2[i] = C._u64([c for xy in p2 for c in xy])
        lo, hi = parallel.shard_range(n, world, rank)
        part1 = C.g1_affine(C.g1_msm_naive(pts1[lo:hi], ks[lo:hi])) if hi > lo else None
        part2 = C.g2_affine(C.g2_msm_naive(pts2[lo:hi], ks[lo:hi])) if hi > lo else None
        got1 = parallel.msm_sharded(part1, g2=False)
        got2 = parallel.msm_sharded(part2, g2=True)
        # several partials in ONE gather (what a sharded Groth16 prove exchanges), with an infinity in the mix
        per_rank = parallel.allgather_points([part1, None, part2], [False, False, True])
        comb = parallel.combine_partials(per_rank, [False, False, True])
        want1 = C.g1_affine(C.g1_msm_naive(pts1, ks))
        want2 = C.g2_affine(C.g2_msm_naive(pts2, ks))
        ok = (got1 == want1 and got2 == want2 and comb[0] == want1 and comb[1] is None and comb[2] == want2
              and len(per_rank) == world)
        ret[rank] = bool(ok)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,n", [(2, 37), (3, 8)])
def test_sharded_msm_allgather_combine(world, n):
    ctx = mp.get_context("spawn")
    ret = ctx.Manager().dict()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n, ret)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(300)
        assert p.exitcode == 0
    assert all(ret.get(r) for r in range(world)), dict(ret)


def test_shard_ranges_partition_the_terms():
    from gosnark_amd import parallel
    import gosnark_amd  # noqa: F401
    for n in (0, 1, 7, 8, 1 << 22):
        for world in (1, 2, 3, 8):
            spans = [parallel.shard_range(n, world, r) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(spans[i][1] == spans[i + 1][0] for i in range(world - 1))
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1


def _scatter_worker(rank, world, port, total, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import gosnark_amd  # noqa: F401
        from gosnark_amd import parallel
        import gpu_util as U
        ok = True
        for proof in range(world + 1):                    # the owner rotates: every rank is the root once, rank 0 twice
            root = parallel.owner_of(proof, world)
            full = U.rand_scalars_u64(total, 100 + proof)          # identical on every rank (seeded): the reference for the check
            got = parallel.scatter_scalars(full if rank == root else None, total, root)
            lo, hi = parallel.shard_range(total, world, rank)
            ok = ok and got.shape == (hi - lo, 4) and np.array_equal(got, full[lo:hi])
        ret[rank] = bool(ok)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,total", [(2, 37), (3, 8), (3, 2)])
def test_values_route_scatter_with_rotating_owner(world, total):
    """The exchange of the strong-scaling route at world size 2 and 3 on CPU: the proof's owner scatters the
    values of H, every rank receives exactly its slice of the contiguous split -- ragged splits and a rank with an empty slice
    included -- whoever the owner is."""
    ctx = mp.get_context("spawn")
    ret = ctx.Manager().dict()
    port = _free_port()
    procs = [ctx.Process(target=_scatter_worker, args=(r, world, port, total, ret)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(300)
        assert p.exitcode == 0
    assert all(ret.get(r) for r in range(world)), dict(ret)
