"""Multi-GPU: contiguous slices of the batch, one process per GPU.

Problems are independent (the reference solves them one at a time,
src/grid_path_plan.py:59-79), so there is no exchange during the solve: rank r
solves problems [lo, hi) of the batch and the results are collected once at
the end.  The backend is whatever the process group was created with
("nccl" = RCCL over xGMI on MI355X; "gloo" in the CPU tests).
"""
import numpy as np

from .batch import PackedBatch


def shard_range(B, rank, world):
    """Contiguous slice of rank ``rank``: [g*ceil(B/G), min(B, (g+1)*ceil(B/G)))."""
    per = -(-B // world)
    lo = min(B, rank * per)
    hi = min(B, lo + per)
    return lo, hi


def solve_sharded(packed: PackedBatch, solve, gather=True):
    """Solve ``packed`` across the ranks of the default process group.

    ``solve(sub_batch) -> output dict`` runs on this rank's slice.  With
    ``gather`` every rank returns the full, globally ordered output dict
    (all_gather of equally padded slices); otherwise the rank's own slice.
    """
    import torch
    import torch.distributed as dist

    world = dist.get_world_size() if dist.is_initialized() else 1
    rank = dist.get_rank() if dist.is_initialized() else 0
    lo, hi = shard_range(packed.B, rank, world)
    out = solve(packed.slice(lo, hi))
    if not gather or world == 1:
        return out
    per = -(-packed.B // world)
    use_cuda = dist.get_backend() == "nccl"
    full = {}
    for key, arr in out.items():
        # res is [4, b]: put the problem axis first for the gather
        a = np.ascontiguousarray(arr.T) if key == "res" else arr
        pad = np.zeros((per,) + a.shape[1:], dtype=a.dtype)
        pad[:a.shape[0]] = a
        t = torch.from_numpy(pad)
        if use_cuda:
            t = t.cuda()
        parts = [torch.empty_like(t) for _ in range(world)]
        dist.all_gather(parts, t)
        cat = torch.cat(parts, 0)[:packed.B] if per * world >= packed.B else torch.cat(parts, 0)
        # drop the padding of every rank but keep order: rank g owns [g*per, g*per + n_g)
        pieces = []
        for g in range(world):
            glo, ghi = shard_range(packed.B, g, world)
            pieces.append(parts[g][:ghi - glo])
        cat = torch.cat(pieces, 0).cpu().numpy()
        full[key] = np.ascontiguousarray(cat.T) if key == "res" else cat
    return full
