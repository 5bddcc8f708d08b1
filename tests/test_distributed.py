"""Two-process (gloo) check of the sharded solve: each rank solves its slice
with the CPU oracle standing in for the device call, results are gathered and
must equal the single-process solve, in the same order."""
import os
import socket
import subprocess
import sys
import textwrap

import numpy as np

from mpc_amd.parallel import shard_range

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_shard_ranges_cover_batch():
    for B in (0, 1, 7, 64, 1000, 1_000_000):
        for G in (1, 2, 3, 8):
            r = [shard_range(B, g, G) for g in range(G)]
            assert r[0][0] == 0 and r[-1][1] == B
            assert all(a[1] == b[0] for a, b in zip(r, r[1:]))
            per = -(-B // G) if B else 0
            assert all(hi - lo <= per for lo, hi in r)


WORKER = textwrap.dedent("""
    import os, sys
    sys.path.insert(0, {root!r}); sys.path.insert(0, os.path.join({root!r}, "tests"))
    import numpy as np, torch.distributed as dist
    import oracle_lib
    from mpc_amd.parallel import solve_sharded
    from mpc_amd.synthetic import synthetic_batch
    dist.init_process_group("gloo")
    packed = synthetic_batch(101)
    full = solve_sharded(packed, lambda sub: oracle_lib.solve_packed(sub, nthreads=2))
    if dist.get_rank() == 0:
        np.savez({out!r}, **full)
    dist.destroy_process_group()
""")


def test_two_rank_gloo_matches_single_process(tmp_path, oracle):
    out = str(tmp_path / "gathered.npz")
    script = tmp_path / "worker.py"
    script.write_text(WORKER.format(root=ROOT, out=out))
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, OMP_NUM_THREADS="2")
    subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
                    "--master-addr", "127.0.0.1", "--master-port", str(port), str(script)],
                   check=True, env=env, timeout=600)
    from mpc_amd.synthetic import synthetic_batch
    ref = oracle.solve_packed(synthetic_batch(101))
    got = np.load(out)
    for k in ref:
        assert got[k].shape == ref[k].shape, k
        assert np.array_equal(got[k], ref[k], equal_nan=True), k
