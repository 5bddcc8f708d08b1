"""Tick latency of the receding-horizon RTI server on one GPU, or on several with one process each.

Usage on the GPU box:  python profiles/rti_latency.py [vehicles] [ticks] [qp_tol preset: reference | loose]
  or, one rank per GPU:  python -m torch.distributed.run --nnodes=1 --nproc-per-node G --master-addr 127.0.0.1 \
                             --master-port P profiles/rti_latency.py [vehicles] [ticks]
Every rank serves a contiguous slice of the fleet (mpc_amd.parallel.shard_range) with no exchange
between ranks; the latency of a tick of the fleet is the slowest rank's, so the per-tick times are
reduced with MAX over the ranks (gloo, a vector of `ticks` doubles, after the run).
Fleet: the first `vehicles` problems of the synthetic benchmark distribution that a full
solve converges on, started from that solution.  A tick = one linearisation + one QP per
vehicle, the plant step and the shift, timed with HIP events on the launch stream (one pair around
the solve, one around shift + plant step), and the host's own clock around the enqueue of each tick;
the five slowest ticks are listed with that split.
"""
import json
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

from mpc_amd.parallel import shard_range  # noqa: E402

V = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
TICKS = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
QP_TOL = sys.argv[3] if len(sys.argv) > 3 else None
RANK, WORLD = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
DEVICE = int(os.environ.get("LOCAL_RANK", "0")) % max(torch.cuda.device_count(), 1)
torch.cuda.set_device(DEVICE)
if WORLD > 1:
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    dist.init_process_group("gloo", rank=RANK, world_size=WORLD)
pool = synthetic_batch(int(V / 0.08) + 4096)
t0 = time.time()
full = _lib.solve_packed(pool, device=DEVICE)
ok = np.nonzero(full["status"] == 0)[0][:V]
assert ok.size == V, f"only {ok.size} feasible problems in the pool"
lo, hi = shard_range(V, RANK, WORLD)
ok = ok[lo:hi]
V_ALL, V = V, hi - lo
pick = lambda a, ax: np.ascontiguousarray(np.take(a, ok, axis=ax))
fleet = type(pool)(V, pool.N, pick(pool.pdata, 1), pick(pool.n_obs, 0), pick(pool.xguess, 2), pool.duration)
srv = RtiServer(fleet, X=full["X"][ok], U=full["U"][ok], device=DEVICE, qp_tol=QP_TOL)
print(f"rank {RANK}: fleet of {V} vehicles ready ({time.time() - t0:.1f} s for the initial full solve of {pool.B})", flush=True)
stream = torch.cuda.current_stream()
ev = [tuple(torch.cuda.Event(enable_timing=True) for _ in range(3)) for _ in range(TICKS)]
qp = []
host_ms = np.zeros(TICKS)
for i, (e0, em, e1) in enumerate(ev):
    h0 = time.perf_counter()
    e0.record(stream)
    srv.solve()
    em.record(stream)
    srv.advance()
    e1.record(stream)
    host_ms[i] = 1e3 * (time.perf_counter() - h0)
    if i % 100 == 99:
        torch.cuda.synchronize()
        qp.append(float(srv.qp_iter.double().mean()))
        print(f"rank {RANK} tick {i + 1}: mean IP iterations {qp[-1]:.1f}, status!=0: {int((srv.status != 0).sum())}", flush=True)
torch.cuda.synchronize()
ms = np.array([e0.elapsed_time(e1) for e0, _, e1 in ev])
solve_ms = np.array([e0.elapsed_time(em) for e0, em, _ in ev])
slow = np.argsort(-ms)[:5]
slowest = [{"tick": int(i), "ms": float(ms[i]), "solve_ms": float(solve_ms[i]), "shift_plant_ms": float(ms[i] - solve_ms[i]),
            "host_enqueue_ms": float(host_ms[i])} for i in slow]
if WORLD > 1:
    t = torch.from_numpy(ms.copy())
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    ms = t.numpy()
x0 = srv.pdata[0:5].t().cpu().numpy()
d = np.hypot(x0[:, 0] - fleet.pdata[5], x0[:, 1] - fleet.pdata[6])
if WORLD > 1:
    dm = torch.tensor([float(d.sum())], dtype=torch.float64)
    dist.all_reduce(dm)
    mean_d = float(dm[0]) / V_ALL
    dist.destroy_process_group()
else:
    mean_d = float(d.mean())
if RANK == 0:
  print(json.dumps({"vehicles": V_ALL, "gpus": WORLD, "ticks": TICKS, "N": fleet.N,
                  "tick_ms_p50": float(np.percentile(ms, 50)), "tick_ms_p99": float(np.percentile(ms, 99)),
                  "tick_ms_max": float(ms.max()), "tick_ms_max_after_first": float(ms[1:].max()) if TICKS > 1 else None,
                  "solve_ms_p50": float(np.percentile(solve_ms, 50)), "host_enqueue_ms_p50": float(np.percentile(host_ms, 50)),
                  "qp_tol": QP_TOL or "reference", "mean_ip_iterations": qp[-1] if qp else None,
                  "slowest_ticks_rank0": slowest, "vehicle_ticks_per_s": V_ALL / (np.median(ms) * 1e-3),
                  "mean_distance_to_goal_m": mean_d}))
