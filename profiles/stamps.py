"""Per-phase cycle shares of the tick loop (diagnostic build libmpcx_stamps.so).

Usage on the GPU box:  python profiles/stamps.py [batch]
Prints, per phase, the share of wavefront cycles and the mean number of lanes
that took part each time the phase ran.
"""
import ctypes as C
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from mpc_amd import _lib  # noqa: E402
from mpc_amd.synthetic import synthetic_batch  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 262144
ONLY = int(sys.argv[2]) if len(sys.argv) > 2 else -1        # solve just this problem of the batch
lib = _lib.load_library(os.path.join(ROOT, "mpc_amd", "csrc", "libmpcx_stamps.so"))
_lib._lib = lib
if os.environ.get("STAMPS_GRID"):
    # the reference's 32-pose grid sweep at N = 800 instead of the synthetic batch
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import cases
    from mpc_amd.batch import pack_problems
    packed = pack_problems(*cases.grid_problems())
elif os.environ.get("STAMPS_REACHABLE"):
    from mpc_amd.synthetic import reachable_batch
    packed = reachable_batch(B)
else:
    packed = synthetic_batch(B)
if ONLY >= 0:
    packed = packed.slice(ONLY, ONLY + 1)
t0 = time.time()
out = _lib.solve_packed(packed)
dt = time.time() - t0
NS = 64
buf = np.zeros((2048, NS), dtype=np.uint64)
lib.mpcx_debug_read_stamps.argtypes = [C.c_void_p, C.c_int]
assert lib.mpcx_debug_read_stamps(buf.ctypes.data, 1) == 0
names = ["A fetch+init", "B sweep L", "sweep B0", "sweep F (affine)", "sweep Bs (both rhs)", "sweep F (both steps)",
         "-", "-", "E write-out", "busy lanes", "ticks*64", "single-problem ticks / extra iterations", "  single B0", "  single F (affine)",
         "  single Bs", "  single F (both)"]
cyc = buf[:, 0::2].astype(np.float64)
cnt = buf[:, 1::2].astype(np.float64)
used = cyc.sum(1) > 0
print(f"B={B} wall {dt:.2f}s (incl. transfers); waves with work: {used.sum()}")
mark_slots = [20, 21, 22, 23]
marks_raw = cyc[:, mark_slots].copy()
cyc[:, mark_slots] = 0
# (time, age) pairs: first burst of a problem at least 1000 / 2000 / 4000 / 8000 iterations old; the end of a
# problem of 5000 iterations or more with the time of its fetch and its index
age_marks = {thr: (buf[:, w].astype(np.float64), buf[:, wa].astype(np.int64))
             for thr, w, wa in ((1000, 12, 13), (2000, 14, 15), (4000, 55, 57), (8000, 59, 61))}
long_end = (buf[:, 41].astype(np.float64), buf[:, 43].astype(np.int64), buf[:, 45].astype(np.float64), buf[:, 47].astype(np.int64))
cyc[:, [6, 7]] = 0
cnt[:, [6, 7]] = 0
cnt[:, [27, 28, 29, 30, 31]] = 0
cnt[:, [20, 21, 22, 23]] = 0
last_out = (buf[:, 49].astype(np.float64), buf[:, 51].astype(np.int64), buf[:, 53].astype(np.float64))
cnt[:, [24, 25, 26]] = 0
# parts of the single-problem sweeps (already contained in slots 12..15)
# bursts (N < 64): slot 12 holds whole bursts, lanes-sum / 64 = iterations + QP ends
sub_slots = {16: "    burst / block: B0 rows + reductions", 17: "    burst: Riccati recursion (block: reductions)", 18: "    block: Riccati pipeline", 19: "    block: stores of the factors", 24: "    burst: F affine", 25: "    burst: Bs", 26: "    burst: F both",
             27: "    burst: loads at the start / step selection", 28: "    sweep L, single-problem form",
             29: "    move to a region", 30: "    burst: stores at the end", 31: "    separate sweeps: B0"}
sub_raw = {k: cyc[used, k].sum() for k in sub_slots}
cyc[:, list(sub_slots)] = 0
tot = cyc[used].sum()
ticks = None
for i, n in enumerate(names):
    c = cyc[used, i].sum()
    print(f"{n:22s} share {100 * c / tot:5.1f}%   cycles {c:.4g}  lanes-sum {cnt[used, i].sum():.3g}")
for k, n in sub_slots.items():
    print(f"{n:26s} share {100 * sub_raw[k] / tot:5.1f}%   cycles {sub_raw[k]:.4g}")
for i in (12, 13, 14, 15):
    calls = cnt[used, i].sum() / 64
    if calls:
        print(f"{names[i]:22s} calls {calls:.4g}  cycles per call {cyc[used, i].sum() / calls:.0f}")
print("cycles per wave (mean) %.3g" % cyc[used].sum(1).mean())
w = cyc[used].sum(1)
print("cycles per wave: min %.3g median %.3g p90 %.3g max %.3g" % (w.min(), np.median(w), np.percentile(w, 90), w.max()))
print("qp_iter sum", int(out["qp_iter"].sum()), "sqp_iter sum", int(out["sqp_iter"].sum()))

# wall-clock marks (100 MHz counter)
t0 = marks_raw[used, 1].min()
for name, col in (("start", 1), ("queue seen empty", 0), ("first extra iteration of a problem older than 1000", 3), ("exit", 2)):
    v = marks_raw[used, col]
    v = v[v > 0]
    if v.size:
        r = (v - t0) / 1e8
        print(f"mark {name:52s} waves {v.size:5d}  min {r.min():.3f}s median {np.median(r):.3f}s "
              f"p90 {np.percentile(r, 90):.3f}s max {r.max():.3f}s")

# the life of every very long problem (5000 iterations or more), from its wavefront's marks
for w in np.nonzero(long_end[0] > 0)[0]:
    line = f"wave {w:4d}: problem {long_end[3][w]} fetched {(long_end[2][w] - t0) / 1e8:.3f}s" if long_end[2][w] > 0 else f"wave {w:4d}: problem {long_end[3][w]} (handed over)"
    for thr, (tm, ag) in age_marks.items():
        if tm[w] > 0:
            line += f", age {ag[w]} at {(tm[w] - t0) / 1e8:.3f}s"
    line += f", ended with {long_end[1][w]} iterations at {(long_end[0][w] - t0) / 1e8:.3f}s"
    print(line)

# the wavefronts that leave last: the last problem each wrote out
ex = marks_raw[:, 2].copy()
ex[~used] = 0
lo = last_out[0].copy()
lo[~used] = 0
for w in np.argsort(-(ex + 1e-3 * lo))[:12]:
    print(f"wave {w:4d} exit {(ex[w] - t0) / 1e8:.3f}s: last problem out at {(last_out[0][w] - t0) / 1e8:.3f}s, "
          f"{last_out[1][w]} iterations, " + (f"fetched {(last_out[2][w] - t0) / 1e8:.3f}s" if last_out[2][w] > 0 else "handed over"))
