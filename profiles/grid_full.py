import sys, time, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
from mpc_amd import _lib
from mpc_amd.grid_path_plan import grid_start_states
from mpc_amd.batch import pack_arrays
starts, total = grid_start_states(0.0, 0.0, 0.0)      # reference function defaults: 21 x 21 x 8 - 8
S = np.zeros((len(starts), 5)); S[:, :3] = np.array(starts)
G = np.zeros((len(starts), 5))
packed = pack_arrays(S, G, 80.0)
print(packed.B, packed.N, flush=True)
t0 = time.perf_counter(); out = _lib.solve_packed(packed); dt = time.perf_counter() - t0
print(f"B={packed.B} N={packed.N}: {dt:.2f} s, converged {(out['status']==0).sum()}, max qp_iter {out['qp_iter'].max()}, mean {out['qp_iter'].mean():.0f}")
