"""Wall time of the reference's full default grid sweep (21 x 21 x 8 - 8 start poses, N = 800)
through the batched driver's packing and the host entry point.
Usage on the GPU box:  python profiles/grid_full.py
"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
from mpc_amd import _lib  # noqa: E402
from mpc_amd.batch import pack_problems  # noqa: E402
from mpc_amd.core_solver import EgoConfig, SimConfig  # noqa: E402
from mpc_amd.grid_path_plan import grid_start_states  # noqa: E402

starts, total = grid_start_states(0.0, 0.0, 0.0)
sim = SimConfig()
sim.duration = 80
egos = []
for (x, y, th) in starts:
    e = EgoConfig()
    e.state_final = [0.0, 0.0, 0.0, 0, 0]
    e.state_start = [x, y, th, 0, 0]
    e.verbose = False
    egos.append(e)
t0 = time.perf_counter()
packed = pack_problems(egos, sim)
t1 = time.perf_counter()
out = _lib.solve_packed(packed)
t2 = time.perf_counter()
print(f"B={packed.B} N={packed.N}: packing {t1 - t0:.2f} s, solve {t2 - t1:.2f} s, converged "
      f"{(out['status'] == 0).sum()}, max qp_iter {out['qp_iter'].max()}, mean {out['qp_iter'].mean():.0f}")
