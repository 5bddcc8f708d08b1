"""Experiment: can a cheap pilot solve rank the problems of the reference's full grid sweep (3 520 poses,
N = 800) well enough to serve as the priority list (mpcx_batch.order) of the real solve?  The pilot is the
same grid at a coarse discretisation (same duration, N = 50 ... 200).  Results never depend on the list.
Usage on the GPU box:  python profiles/grid_pilot_exp.py
"""
import dataclasses
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
from mpc_amd import _lib  # noqa: E402
from mpc_amd.batch import longest_first, pack_problems  # noqa: E402
from mpc_amd.core_solver import EgoConfig, SimConfig  # noqa: E402
from mpc_amd.grid_path_plan import grid_start_states  # noqa: E402

starts, total = grid_start_states(0.0, 0.0, 0.0)
egos = []
for (x, y, th) in starts:
    e = EgoConfig()
    e.state_final = [0.0, 0.0, 0.0, 0, 0]
    e.state_start = [x, y, th, 0, 0]
    e.verbose = False
    egos.append(e)


def packed_at(n):
    sim = SimConfig()
    sim.duration = 80
    sim.dt = 80.0 / n
    p = pack_problems(egos, sim)
    assert p.N in (n, n - 1), (p.N, n)
    return p


def timed(p):
    t0 = time.perf_counter()
    out = _lib.solve_packed(p)
    return time.perf_counter() - t0, out


_lib.solve_packed(packed_at(50).slice(0, 64))      # module load
full = packed_at(800)
t_own, ref = timed(full)
q = ref["qp_iter"]
print(f"N={full.N} B={full.B}: library's own order {t_own:.2f} s; max qp_iter {q.max()}, mean {q.mean():.0f}", flush=True)
t_ideal, out = timed(dataclasses.replace(full, order=longest_first(q)))
same = all(np.array_equal(out[k], ref[k]) for k in ref)
print(f"priority list from the solve itself (the ideal): {t_ideal:.2f} s, same results: {same}", flush=True)
top = np.argsort(-q, kind="stable")[:20]
for n in (50, 100, 200):
    t_p, pilot = timed(packed_at(n))
    qp = pilot["qp_iter"]
    rank_of = np.empty(full.B, dtype=np.int64)
    rank_of[np.argsort(-qp, kind="stable")] = np.arange(full.B)
    # Spearman correlation of the two counts
    ra = np.argsort(np.argsort(q)).astype(float)
    rb = np.argsort(np.argsort(qp)).astype(float)
    rho = np.corrcoef(ra, rb)[0, 1]
    t_o, out = timed(dataclasses.replace(full, order=longest_first(qp)))
    same = all(np.array_equal(out[k], ref[k]) for k in ref)
    print(f"pilot N={n}: {t_p * 1e3:.0f} ms; rank correlation {rho:.2f}; the 20 longest problems rank "
          f"{sorted(rank_of[top].tolist())} in it; solve with its list {t_o:.2f} s (+ pilot = {t_o + t_p:.2f} s), "
          f"same results: {same}", flush=True)
