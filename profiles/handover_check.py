import os, sys, time
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import numpy as np, torch
import oracle_lib, cases
from mpc_amd import _lib
from mpc_amd.synthetic import synthetic_batch, reachable_batch
fast = oracle_lib.load(oracle_lib.ORACLE_FAST_SO if os.path.exists(oracle_lib.ORACLE_FAST_SO) else oracle_lib.ORACLE_SO)
for name, packed in (("synthetic 20000", synthetic_batch(20000, seed=3)), ("reachable 6000", reachable_batch(6000, seed=4)), ("synthetic 70000", synthetic_batch(70000, seed=5))):
    t0 = time.time(); gpu = _lib.solve_packed(packed); t1 = time.time()
    ref = oracle_lib.solve_packed(packed, nthreads=16, lib=fast)
    cases.assert_same_solution(gpu, ref, name)
    print(name, "identical, device %.3f s" % (t1 - t0), flush=True)
