"""Multi-GPU: contiguous slices of the batch, one process per GPU.

Problems are independent (the reference solves them one at a time,
src/grid_path_plan.py:59-79), so there is no exchange during the solve: rank r
solves problems [lo, hi) of the batch and the results are collected once at
the end, on rank 0 (``gather_to_root``) or on every rank.  The backend is
whatever the process group was created with ("nccl" = RCCL over xGMI on
MI355X, where the tensors stay in device memory; "gloo" in the CPU tests).
"""
import numpy as np

from .batch import PackedBatch

RESULT_KEYS = ("X", "U", "status", "sqp_iter", "qp_iter", "cost", "res")


def shard_range(B, rank, world):
    """Contiguous slice of rank ``rank``: [g*ceil(B/G), min(B, (g+1)*ceil(B/G)))."""
    per = -(-B // world)
    lo = min(B, rank * per)
    hi = min(B, lo + per)
    return lo, hi


def gather_to_root(out, B_total, dst=0, everyone=False):
    """Collect the ranks' result slices into the whole batch, in the caller's order.

    ``out``: this rank's output dict of torch tensors (device tensors under RCCL), problem axis
    first except ``res`` ([4, b]).  Returns the dict of whole-batch tensors on rank ``dst`` (on
    every rank with ``everyone``), None elsewhere.  One collective per array: a gather to ``dst``
    (each rank sends its slice once, xGMI point to point), or an all-gather.  Slices are padded to
    the common length ceil(B/G) for the collective and trimmed after it.
    """
    import torch
    import torch.distributed as dist

    world = dist.get_world_size()
    rank = dist.get_rank()
    per = -(-B_total // world)
    full = {} if (everyone or rank == dst) else None
    for key in RESULT_KEYS:
        t = out[key]
        a = t.t().contiguous() if key == "res" else t
        if a.shape[0] != per:
            pad = torch.zeros((per,) + tuple(a.shape[1:]), dtype=a.dtype, device=a.device)
            pad[:a.shape[0]] = a
            a = pad
        if everyone:
            parts = [torch.empty_like(a) for _ in range(world)]
            dist.all_gather(parts, a)
        else:
            parts = [torch.empty_like(a) for _ in range(world)] if rank == dst else None
            dist.gather(a, parts, dst=dst)
        if full is None:
            continue
        pieces = []
        for g in range(world):
            glo, ghi = shard_range(B_total, g, world)
            pieces.append(parts[g][:ghi - glo])
        cat = torch.cat(pieces, 0)
        full[key] = cat.t().contiguous() if key == "res" else cat
    return full


def solve_sharded(packed: PackedBatch, solve, gather=True, everyone=True):
    """Solve ``packed`` across the ranks of the default process group.

    ``solve(sub_batch) -> output dict`` (numpy arrays) runs on this rank's slice.  With ``gather``
    the globally ordered output dict is returned on every rank (``everyone``) or on rank 0 only
    (None elsewhere); otherwise the rank's own slice.
    """
    import torch
    import torch.distributed as dist

    world = dist.get_world_size() if dist.is_initialized() else 1
    rank = dist.get_rank() if dist.is_initialized() else 0
    lo, hi = shard_range(packed.B, rank, world)
    out = solve(packed.slice(lo, hi))
    if not gather or world == 1:
        return out
    use_cuda = dist.get_backend() == "nccl"
    tens = {k: torch.from_numpy(np.ascontiguousarray(out[k])) for k in RESULT_KEYS}
    if use_cuda:
        tens = {k: v.cuda() for k, v in tens.items()}
    full = gather_to_root(tens, packed.B, dst=0, everyone=everyone)
    if full is None:
        return None
    return {k: v.cpu().numpy() for k, v in full.items()}
