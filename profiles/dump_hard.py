import sys, os, numpy as np
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, ROOT)
import torch
from mpc_amd import _lib
from mpc_amd.synthetic import synthetic_batch, reachable_batch
out = {}
for name, mk in (("synthetic", synthetic_batch), ("reachable", reachable_batch)):
    for seed in (20250310, 20250311, 20250315):
        p = mk(1000000, seed=seed, device_guess="all")
        r = _lib.solve_packed(p)
        out[f"{name}_{seed}_qp"] = r["qp_iter"].astype(np.int32)
        out[f"{name}_{seed}_sqp"] = r["sqp_iter"].astype(np.int16)
        out[f"{name}_{seed}_st"] = r["status"].astype(np.int8)
        print(name, seed, "max qp", r["qp_iter"].max(), "status2", int((r["status"] == 2).sum()), flush=True)
np.savez_compressed(os.path.join(ROOT, "gpurun_out", "hard.npz"), **out)
