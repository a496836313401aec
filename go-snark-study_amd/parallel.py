"""Multi-GPU layer: one process per GPU, torch.distributed (backend "nccl" = RCCL over xGMI on the
GPU box, "gloo" in the CPU tests).

An MSM is a sum over independent terms (groth16.go:243-250,269-271), so it shards by contiguous term
range: every rank runs the whole Pippenger pipeline on its resident shard and emits ONE partial point;
the only exchange is an all-gather of those partials (72 B per G1 point, 136 B per G2 point per rank:
affine words + an infinity flag) followed by world-1 local curve additions -- RCCL has no curve-point
reduction operator, so a literal all-reduce cannot add them.  Batches of independent
proofs (BASELINE configs[4]) need no collective at all: see bench.py."""
import os

import numpy as np
import torch
import torch.distributed as dist

from . import capi


def shard_range(n, world, rank):
    """Contiguous term range [lo, hi) of `rank` (first n % world ranks get one extra term)."""
    q, rem = divmod(n, world)
    lo = rank * q + min(rank, rem)
    return lo, lo + q + (1 if rank < rem else 0)


def _encode(point, g2):
    words = 16 if g2 else 8
    buf = np.zeros(words + 1, dtype=np.uint64)
    if point is None:
        buf[words] = 1
    else:
        flat = [point[0][0], point[0][1], point[1][0], point[1][1]] if g2 else [point[0], point[1]]
        buf[:words] = capi.ints_to_u64(flat).reshape(-1)
    return buf


def _decode(buf, g2):
    words = 16 if g2 else 8
    if int(buf[words]) != 0:
        return None
    v = capi.u64_to_ints(buf[:words])
    return ((v[0], v[1]), (v[2], v[3])) if g2 else (v[0], v[1])


def allgather_points(partials, g2_flags, group=None):
    """partials: this rank's affine points (tuples / None = infinity), g2_flags[i] tells the group of
    partials[i].  ONE all-gather of the packed bytes; returns per_rank[rank][i]."""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    packed = np.concatenate([_encode(p, g2) for p, g2 in zip(partials, g2_flags)]).view(np.uint8)
    # A lone rank has nothing to exchange -- unless GS_FORCE_COLLECTIVE=1 asks for the collective anyway, so that the
    # RCCL branch below is executed (and checked) on a 1-GPU box.
    if world == 1 and not (dist.is_initialized() and os.environ.get("GS_FORCE_COLLECTIVE") == "1"):
        gathered = [packed]
    else:
        dev = torch.device("cuda", torch.cuda.current_device()) if dist.get_backend(group) == "nccl" else torch.device("cpu")
        mine = torch.from_numpy(packed.copy()).to(dev)
        outs = [torch.empty_like(mine) for _ in range(world)]
        dist.all_gather(outs, mine, group=group)
        gathered = [o.cpu().numpy() for o in outs]
    res = []
    for raw in gathered:
        words = raw.view(np.uint64)
        pts, pos = [], 0
        for g2 in g2_flags:
            k = (16 if g2 else 8) + 1
            pts.append(_decode(words[pos:pos + k], g2))
            pos += k
        res.append(pts)
    return res


def combine_partials(per_rank, g2_flags):
    """Sum the ranks' partial points element-wise with the library's complete curve addition."""
    return [capi.sum_affine([per_rank[r][i] for r in range(len(per_rank))], g2=g2) for i, g2 in enumerate(g2_flags)]


def msm_sharded(local_partial, g2=False, group=None):
    """Finish a term-sharded MSM: `local_partial` is this rank's sum over its shard."""
    per_rank = allgather_points([local_partial], [g2], group)
    return combine_partials(per_rank, [g2])[0]


def msm_g1_sharded(bases, scalars, n_local, group=None):
    """Each rank holds `n_local` resident bases/scalars (its shard of the global term range)."""
    return msm_sharded(capi.msm_resident(bases, scalars, n_local), False, group)


def msm_g2_sharded(bases, scalars, n_local, group=None):
    return msm_sharded(capi.msm_resident(bases, scalars, n_local, g2=True), True, group)


# ---- values route: the owner of a proof scatters H's values ------------------------------------------------
def owner_of(proof_index, world):
    """The ranks take turns with the polynomial stage: proof i is owned by rank i mod world."""
    return proof_index % world


def scatter_scalars(full_u64, total, root, group=None):
    """torch.distributed twin of gs_scalars_scatter (same contiguous split: shard_range): rank `root` passes the [total, 4] uint64
    array, the others None; every rank returns ITS slice.  Backend nccl (= RCCL) on the GPU box, gloo in the CPU tests.  The
    library's own scatter (ncclSend / ncclRecv between resident vectors, no host round trip) is what bench.py uses; this one is
    for hosts that already drive their collectives through torch."""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    lo, hi = shard_range(total, world, rank)
    if world == 1:
        return np.ascontiguousarray(full_u64, dtype=np.uint64).reshape(-1, 4)[lo:hi].copy()
    dev = torch.device("cuda", torch.cuda.current_device()) if dist.get_backend(group) == "nccl" else torch.device("cpu")
    # dist.scatter wants equal shapes: pad every slice to the longest (ceil(total / world)) rows of 4 int64 words
    rows = -(-total // world) if total else 0
    mine = torch.zeros((max(rows, 1), 4), dtype=torch.int64, device=dev)
    chunks = None
    if rank == root:
        full = np.ascontiguousarray(full_u64, dtype=np.uint64).reshape(-1, 4)
        if full.shape[0] != total:
            raise ValueError("scatter_scalars: the root's vector has %d rows, total = %d" % (full.shape[0], total))
        chunks = []
        for r in range(world):
            a, b = shard_range(total, world, r)
            pad = np.zeros((max(rows, 1), 4), dtype=np.uint64)
            pad[:b - a] = full[a:b]
            chunks.append(torch.from_numpy(pad.view(np.int64)).to(dev))
    dist.scatter(mine, chunks, src=root, group=group)
    return mine.cpu().numpy().view(np.uint64)[:hi - lo].copy()
