"""Iteration-count tail of the benchmark batch (run on the GPU box)."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from mpc_amd import _lib
from mpc_amd.synthetic import synthetic_batch
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
p = synthetic_batch(B)
t0 = time.time(); out = _lib.solve_packed(p); print("solve wall", time.time() - t0)
qi, si, st = out["qp_iter"], out["sqp_iter"], out["status"]
order = np.argsort(qi)[::-1]
print("top 30 qp_iter:", qi[order[:30]])
print("their sqp_iter:", si[order[:30]])
print("their status  :", st[order[:30]])
print("their index   :", order[:30])
for thr in (500, 1000, 2000, 3000, 5000):
    print(f"problems with qp_iter > {thr}: {(qi > thr).sum()}")
print("status counts", {int(s): int((st == s).sum()) for s in np.unique(st)})
print("sum qp_iter", qi.sum())
