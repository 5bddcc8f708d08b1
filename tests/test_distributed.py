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
    solve = lambda sub: oracle_lib.solve_packed(sub, nthreads=2)
    full = solve_sharded(packed, solve, everyone=False)            # gather to rank 0
    assert (full is None) == (dist.get_rank() != 0)
    if dist.get_rank() == 0:
        np.savez({out!r}, **full)
    everywhere = solve_sharded(packed, solve, everyone=True)       # all-gather
    np.savez({out!r} + f".rank{{dist.get_rank()}}.npz", **everywhere)
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
    for path in (out, out + ".rank0.npz", out + ".rank1.npz"):
        got = np.load(path)
        for k in ref:
            assert got[k].shape == ref[k].shape, (path, k)
            assert np.array_equal(got[k], ref[k], equal_nan=True), (path, k)


def test_bench_refuses_to_report_more_gpus_than_it_has():
    """`bench.py --gpus N` without a launcher starts N ranks itself; with fewer devices visible it
    fails loudly instead of printing a line (here: as many GPUs as this host has, plus two)."""
    import torch
    n = torch.cuda.device_count() + 2
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(n), "--steps", "1", "--warmup", "0",
                        "--batch", "64"], capture_output=True, text=True, timeout=300,
                       env={k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")})
    assert r.returncode != 0 and "GPU(s) visible" in r.stderr and r.stdout.strip() == ""
    # under a launcher whose world size disagrees with --gpus it is an error too
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "1", "--warmup", "0"],
                       capture_output=True, text=True, timeout=300,
                       env=dict(os.environ, WORLD_SIZE="2", RANK="0", LOCAL_RANK="0"))
    assert r.returncode != 0 and "WORLD_SIZE=2" in r.stderr and r.stdout.strip() == ""
