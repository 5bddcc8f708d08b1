"""Wall time of small batches through the host entry point (uploads and downloads included).

Usage on the GPU box:  python profiles/latency.py
"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from mpc_amd import _lib  # noqa: E402
from mpc_amd.synthetic import synthetic_batch  # noqa: E402
from mpc_amd.grid_path_plan import grid_start_states  # noqa: E402
from mpc_amd.core_solver import EgoConfig, SimConfig  # noqa: E402
from mpc_amd.batch import pack_problems  # noqa: E402


def timed(packed, reps=3):
    best = 1e9
    for _ in range(reps):
        t0 = time.perf_counter()
        out = _lib.solve_packed(packed)
        best = min(best, time.perf_counter() - t0)
    return best, out


_lib.solve_packed(synthetic_batch(64))      # first call: module load
big = synthetic_batch(65536)
for B in (1, 8, 64, 512, 4096, 65536):
    dt, out = timed(big.slice(0, B))
    print(f"synthetic N=50  B={B:6d}  {dt * 1e3:9.2f} ms   {B / dt:10.0f} OCP/s   "
          f"converged {int((out['status'] == 0).sum())}  max qp_iter {int(out['qp_iter'].max())}")

egos = []
for (x, y, th) in grid_start_states(0.0, 0.0, 0.0, 40, 20, 4)[0]:
    e = EgoConfig()
    e.state_start = [x, y, th, 0.0, 0.0]
    e.state_final = [0.0, 0.0, 0.0, 0.0, 0.0]
    egos.append(e)
sim = SimConfig()
sim.duration = 80.0
packed = pack_problems(egos, sim)
dt, out = timed(packed, reps=2)
print(f"grid N={packed.N} B={packed.B}  {dt * 1e3:9.2f} ms  converged {int((out['status'] == 0).sum())}  "
      f"max qp_iter {int(out['qp_iter'].max())}")
