"""Experiment: does grouping the fleet by the previous tick's interior-point iteration counts shorten an RTI tick?
A wavefront's 64 vehicles are swept together until the slowest is done, so a tick costs each wavefront the
MAXIMUM count among its vehicles.  Orders tried, each tick, from the counts of the tick before:
  none      the fleet as it is
  sorted    ascending by count (the heavy ones end up together: tried before, slower)
  balanced  ascending by count, except that the heaviest B/64 vehicles are dealt one to a wavefront
  tiered<h> ascending by count, except that the heaviest h B/64 vehicles are dealt h to a wavefront
Fetch order of the library = the caller's (MPCX_DEAL=0), so that a wavefront takes 64 neighbours.
Usage on the GPU box:  python profiles/rti_balance_exp.py [vehicles] [ticks]
"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from mpc_amd import _lib  # noqa: E402
from mpc_amd.rti import RtiServer  # noqa: E402
from mpc_amd.synthetic import synthetic_batch  # noqa: E402

V = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
TICKS = int(sys.argv[2]) if len(sys.argv) > 2 else 200
pool = synthetic_batch(int(V / 0.08) + 4096)
full = _lib.solve_packed(pool)
ok = np.nonzero(full["status"] == 0)[0][:V]
assert ok.size == V
pick = lambda a, ax: np.ascontiguousarray(np.take(a, ok, axis=ax))
fleet = type(pool)(V, pool.N, pick(pool.pdata, 1), pick(pool.n_obs, 0), pick(pool.xguess, 2), pool.duration)


def permute(srv, perm):
    srv.pdata.copy_(srv.pdata[:, perm])
    srv.n_obs.copy_(srv.n_obs[perm])
    srv.xguess.copy_(srv.xguess[:, :, perm])
    srv.uguess.copy_(srv.uguess[:, :, perm])


def order_for(kind, counts):
    B = counts.numel()
    asc = torch.argsort(counts, stable=True)
    if kind == "sorted":
        return asc
    W = B // 64
    if kind == "balanced":
        body, heavy = asc[:B - W], asc[B - W:]
        out = torch.empty_like(asc)
        out.view(W, 64)[:, :63] = body.view(W, 63)
        out.view(W, 64)[:, 63] = heavy
        return out
    # tiered<h>: the heaviest h per wavefront are dealt round-robin (they end in single-problem mode, which a
    # wavefront enters once 8 or fewer of its vehicles are still iterating); the rest ascending
    h = int(kind[6:])
    body, heavy = asc[:B - h * W], asc[B - h * W:]
    out = torch.empty_like(asc)
    out.view(W, 64)[:, :64 - h] = body.view(W, 64 - h)
    out.view(W, 64)[:, 64 - h:] = heavy.view(h, W).t()
    return out


for deal, kind in (("1", "none"), ("0", "tiered8"), ("0", "tiered6"), ("0", "tiered4"), ("0", "tiered12")):
    os.environ["MPCX_DEAL"] = deal
    srv = RtiServer(fleet, X=full["X"][ok], U=full["U"][ok])
    stream = torch.cuda.current_stream()
    ms, perm_ms, wave_max = [], [], []
    for i in range(TICKS):
        e0, e1, e2 = (torch.cuda.Event(enable_timing=True) for _ in range(3))
        e0.record(stream)
        srv.tick()
        e1.record(stream)
        if kind != "none":
            permute(srv, order_for(kind, srv.qp_iter))
        e2.record(stream)
        torch.cuda.synchronize()
        ms.append(e0.elapsed_time(e1))
        perm_ms.append(e1.elapsed_time(e2))
        if i % 50 == 49:
            q = srv.qp_iter.view(-1, 64)
            wave_max.append(float(q.max(1).values.double().mean()))
    ms = np.array(ms[10:])
    print(f"deal={deal} order={kind:9s} tick p50 {np.percentile(ms, 50):6.2f} ms  p99 {np.percentile(ms, 99):6.2f}  max {ms.max():6.2f}   "
          f"reorder {np.median(perm_ms):5.2f} ms   mean IP iterations {float(srv.qp_iter.double().mean()):.2f}, "
          f"mean over wavefronts of the largest count {np.mean(wave_max):.2f}, failed {int((srv.status != 0).sum())}", flush=True)
