"""Device against CPU restatement on whole batches (test infrastructure: drives the oracle).

Usage on the GPU box:  python profiles/parity_campaign.py [problems per batch]
Solves each batch on the HIP device and on the oracle (all host threads) and compares statuses,
iteration counts, states and controls bit for bit; prints one line per batch.
"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: E402,F401  one HIP runtime per process
import oracle_lib  # noqa: E402
from mpc_amd import _lib  # noqa: E402
from mpc_amd.batch import longest_first  # noqa: E402
from mpc_amd.synthetic import reachable_batch, synthetic_batch  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
threads = max(1, min(len(os.sched_getaffinity(0)), int(os.environ.get("OMP_NUM_THREADS", "16") or 16)))
fast = oracle_lib.load(oracle_lib.ORACLE_FAST_SO if os.path.exists(oracle_lib.ORACLE_FAST_SO) else oracle_lib.ORACLE_SO)


def compare(name, packed, opts=None, oopts=None, with_order=True):
    t0 = time.time()
    gpu = _lib.solve_packed(packed, opts=opts)
    t1 = time.time()
    ref = oracle_lib.solve_packed(packed, opts=oopts, nthreads=threads, lib=fast)
    t2 = time.time()
    same = all(np.array_equal(gpu[k], ref[k]) for k in ("status", "sqp_iter", "qp_iter"))
    ok = ref["status"] != 1
    same = same and np.array_equal(gpu["X"][ok], ref["X"][ok]) and np.array_equal(gpu["U"][ok], ref["U"][ok])
    same = same and np.array_equal(gpu["cost"][ok], ref["cost"][ok])
    mix = np.bincount(ref["status"], minlength=5)
    print(f"{name:44s} B={packed.B:8d}  identical={same}  status mix {mix.tolist()}  "
          f"IP iterations {int(ref['qp_iter'].sum())}  device {t1 - t0:.1f} s  oracle {t2 - t1:.1f} s ({threads} threads)")
    sys.stdout.flush()
    if with_order:
        # once more with the caller's priority list (the batch ranked by these iteration counts)
        import dataclasses
        again = _lib.solve_packed(dataclasses.replace(packed, order=longest_first(ref["qp_iter"])), opts=opts)
        same2 = all(np.array_equal(again[k], gpu[k]) for k in ("status", "sqp_iter", "qp_iter", "X", "U", "cost", "res"))
        print(f"{'  the same with mpcx_batch.order':44s} B={packed.B:8d}  identical={same2}")
        sys.stdout.flush()
        same = same and same2
    return same


good = True
good &= compare("benchmark batch (seed 20250310)", synthetic_batch(B, device_guess="all"))
good &= compare("reachable goals", reachable_batch(B // 4, device_guess="all"))
good &= compare("one obstacle each", synthetic_batch(B // 4, obstacle=True, device_guess="all"))
o, oo = _lib.default_opts(), oracle_lib.default_opts()
o.globalization = oo.globalization = 1
good &= compare("benchmark batch, line search on", synthetic_batch(B // 4, seed=99, device_guess="all"), o, oo)
print("ALL IDENTICAL" if good else "MISMATCH")
sys.exit(0 if good else 1)
