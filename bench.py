#!/usr/bin/env python3
"""Headline benchmark: converged OCPs/sec on the 1M random start/goal batch.

One step = one full solve of the rank's batch (N = 50 kinematic-car problems,
seed 20250310 + rank) by the fused HIP kernel, inputs and outputs resident in
HBM.  Prints ONE JSON line on rank 0.  Multi-GPU: one process per GPU, each
solves its own batch (no data-path collective); the only collective is the
reduction of the counters for the report.
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

FP64_VECTOR_PEAK_TFLOPS = 78.6      # MI355X vector FP64 (datasheet)
HBM_PEAK_GBS = 8000.0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--batch", type=int, default=1_000_000, help="problems per GPU")
    ap.add_argument("--seed", type=int, default=None, help="batch seed of rank 0 (default: the headline seed)")
    ap.add_argument("--same-batch", action="store_true",
                    help="every rank solves rank 0's batch (per-GPU work identical; default: seed + rank)")
    ap.add_argument("--obstacle", action="store_true",
                    help="second workload: one obstacle (r=1, margin 0.5) beside the middle of every path")
    ap.add_argument("--order", default=None, help="experiment: permute the batch before upload (align | align2 | align3 | counts:FILE | reverse; +deal)")
    ap.add_argument("--lib", default=None, help="another build of the library (diagnostic builds)")
    ap.add_argument("--cpu-sample", type=int, default=262144, help="problems timed on the CPU oracle (0 = skip)")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    from mpc_amd import _lib, flops
    from mpc_amd.synthetic import SEED, synthetic_batch

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device (there is no CPU fallback)")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)

    lib = _lib.load_library(args.lib)
    opts = _lib.default_opts(lib)
    opts.max_obs = 1 if args.obstacle else 0

    B = args.batch
    seed0 = SEED if args.seed is None else args.seed
    packed = synthetic_batch(B, seed=seed0 + (0 if args.same_batch else rank), obstacle=args.obstacle)
    N = packed.N
    if args.order:
        from mpc_amd.batch import F_X0, F_GOAL
        sx, sy, sth = packed.pdata[F_X0], packed.pdata[F_X0 + 1], packed.pdata[F_X0 + 2]
        gx, gy, gth = packed.pdata[F_GOAL], packed.pdata[F_GOAL + 1], packed.pdata[F_GOAL + 2]
        phi = np.arctan2(gy - sy, gx - sx)
        wrap = lambda a: (a + np.pi) % (2 * np.pi) - np.pi
        score = np.abs(wrap(phi - sth)) + np.abs(wrap(gth - phi))
        if args.order.split("+")[0] == "align":
            perm = np.argsort(score, kind="stable")
        elif args.order == "align2":      # the aligned fifth first, in the given order; then the others
            first = score <= np.quantile(score, 0.2)
            perm = np.concatenate([np.nonzero(first)[0], np.nonzero(~first)[0]])
        elif args.order == "align3":     # aligned AND short manoeuvres first
            dist = np.hypot(gx - sx, gy - sy)
            first = (score <= np.quantile(score, 0.17)) & (dist < 4.3)
            perm = np.concatenate([np.nonzero(first)[0], np.nonzero(~first)[0]])
        elif args.order.split("+")[0].startswith("counts:"):  # longest first, from the iteration counts a previous run saved
            perm = np.argsort(-np.load(args.order.split("+")[0][7:]).astype(np.int64), kind="stable")
        else:
            perm = np.arange(B)[::-1].copy()
        if args.order.endswith("+deal") or os.environ.get("MPCX_BENCH_DEAL"):
            # deal the head of the order out over the wavefronts (1024 x 64 lanes fetch at time 0)
            n = min(B, 65536) // 1024 * 1024
            perm = np.concatenate([perm[:n].reshape(-1, 1024).T.reshape(-1), perm[n:]])
        packed = type(packed)(B, N, np.ascontiguousarray(packed.pdata[:, perm]), np.ascontiguousarray(packed.n_obs[perm]),
                              np.ascontiguousarray(packed.xguess[:, :, perm]), packed.duration)

    # inputs resident in HBM before the timed region
    d_pd = torch.from_numpy(packed.pdata).to(dev)
    d_no = torch.from_numpy(packed.n_obs).to(dev)
    d_xg = torch.from_numpy(packed.xguess).to(dev)
    d_X = torch.empty((B, N + 1, 5), dtype=torch.float64, device=dev)
    d_U = torch.empty((B, N, 2), dtype=torch.float64, device=dev)
    d_st = torch.empty(B, dtype=torch.int32, device=dev)
    d_si = torch.empty(B, dtype=torch.int32, device=dev)
    d_qi = torch.empty(B, dtype=torch.int32, device=dev)
    d_co = torch.empty(B, dtype=torch.float64, device=dev)
    d_re = torch.empty((4, B), dtype=torch.float64, device=dev)
    ws_bytes = lib.mpcx_workspace_bytes(B, N)
    d_ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
    assert d_ws.data_ptr() % 256 == 0

    bs = _lib.MpcxBatch(B, N, d_pd.data_ptr(), d_no.data_ptr(), d_xg.data_ptr(), None)
    rs = _lib.MpcxResult(d_X.data_ptr(), d_U.data_ptr(), d_st.data_ptr(), d_si.data_ptr(),
                         d_qi.data_ptr(), d_co.data_ptr(), d_re.data_ptr())
    stream = torch.cuda.current_stream()

    def step():
        rc = lib.mpcx_solve_batch_device(C.byref(bs), C.byref(opts), C.byref(rs),
                                         d_ws.data_ptr(), ws_bytes, stream.cuda_stream)
        if rc != 0:
            raise RuntimeError(lib.mpcx_last_error().decode())

    def barrier():
        if world > 1:
            dist.barrier()

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    barrier()
    torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
          for _ in range(args.steps)]
    t0 = time.perf_counter()
    for e0, e1 in ev:
        e0.record(stream)
        step()
        e1.record(stream)
    torch.cuda.synchronize()
    barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    kernel_ms = float(np.mean([e0.elapsed_time(e1) for e0, e1 in ev])) if ev else 0.0

    # outcome counters reduced on the device (the per-problem arrays are only downloaded for the CPU sample)
    d_stats = torch.empty(8, dtype=torch.int64, device=dev)
    if lib.mpcx_batch_stats_device(d_st.data_ptr(), d_si.data_ptr(), d_qi.data_ptr(), B, d_stats.data_ptr(),
                                   C.c_void_p(stream.cuda_stream)) != 0:
        raise RuntimeError(lib.mpcx_last_error().decode())
    stats = d_stats.cpu().numpy()
    n_conv = int(stats[0])
    status = d_st.cpu().numpy()
    sqp_it = d_si.cpu().numpy()
    qp_it = d_qi.cpu().numpy()
    assert n_conv == int((status == 0).sum())
    if os.environ.get("MPCX_BENCH_SAVE_COUNTS"):
        np.save(os.environ["MPCX_BENCH_SAVE_COUNTS"], qp_it)
    fl = flops.flops_from_stats(stats, N)
    assert fl == flops.algorithmic_flops(status, sqp_it, qp_it, N)

    tot = torch.tensor([elapsed, float(n_conv), fl, float(B)], dtype=torch.float64, device=dev)
    if world > 1:
        mx = tot[:1].clone()
        dist.all_reduce(mx, op=dist.ReduceOp.MAX)
        sm = tot[1:].clone()
        dist.all_reduce(sm, op=dist.ReduceOp.SUM)
        elapsed = float(mx[0])
        n_conv_all, fl_all, B_all = (float(v) for v in sm)
    else:
        n_conv_all, fl_all, B_all = float(n_conv), fl, float(B)

    if rank != 0:
        if world > 1:
            dist.destroy_process_group()
        return

    steps = max(args.steps, 1)
    value = n_conv_all * steps / elapsed
    achieved_tf = fl / (kernel_ms * 1e-3) / 1e12 if kernel_ms > 0 else 0.0
    # Memory-side bytes of one launch from the committed counter passes (same batch only).  FETCH_SIZE
    # counts half of the bytes of this kernel's loads (64 lanes x 8 B in one 512-byte segment): measured
    # on the same access shape with a known byte count (profiles/micro/stream_rw, r01_pmc_extra.txt);
    # WRITE_SIZE is exact for it.
    traffic = None
    try:
        with open(os.path.join(ROOT, "profiles", "pmc_traffic.json")) as fh:
            pmc = json.load(fh)
        if int(pmc["batch"]) == B and not args.obstacle and args.seed is None:
            traffic = (float(pmc["read_factor"]) * float(pmc["fetch_kib"]) + float(pmc["write_kib"])) * 1024.0
    except (OSError, ValueError, KeyError):
        pass
    io_bytes = flops.algorithmic_io_bytes(N) * B
    line = {
        "metric": "converged OCPs/sec (N=50 kinematic car)",
        "value": value, "unit": "OCP/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": 1e3 * elapsed / steps,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f64", "data": "synthetic",
        "config": {"workload": f"{B} random start/goal pose OCPs per GPU, N=50, "
                               + ("one obstacle each" if args.obstacle else "no obstacles") + (f", seed {seed0} on every rank" if args.same_batch else f", seed {seed0}+rank"),
                   "problems_per_gpu": B, "N": N, "problems_total": int(B_all),
                   "converged_fraction": n_conv_all / B_all,
                   "solved_per_sec_all_statuses": B_all * steps / elapsed,
                   "parallelism": f"independent batch slice per GPU x{world}"},
        "roofline": {"bound": "valu_fp64", "achieved": achieved_tf, "peak": FP64_VECTOR_PEAK_TFLOPS,
                     "unit": "TFLOP/s", "frac": achieved_tf / FP64_VECTOR_PEAK_TFLOPS,
                     "traffic": traffic,
                     "traffic_GBs": traffic / (kernel_ms * 1e-3) / 1e9 if traffic and kernel_ms > 0 else None,
                     "traffic_frac_of_hbm_peak": traffic / (kernel_ms * 1e-3) / 1e9 / HBM_PEAK_GBS
                     if traffic and kernel_ms > 0 else None,
                     "kernel": "mpcx_sqp_kernel<1>" if args.obstacle else "mpcx_sqp_kernel<0>", "kernel_ms": kernel_ms,
                     "algorithmic_flops_per_launch": fl,
                     **({"flops_note": "per-iteration constants of the obstacle-free count (a lower bound here)"}
                        if args.obstacle else {}),
                     "batch_io_GBs": io_bytes / (kernel_ms * 1e-3) / 1e9 if kernel_ms > 0 else 0.0},
    }

    if args.cpu_sample > 0 and world == 1:
        import oracle_lib
        n = min(args.cpu_sample, B)
        cores = min(16, os.cpu_count() or 1)
        sample = packed.slice(0, n)
        t1 = time.perf_counter()
        ref = oracle_lib.solve_packed(sample, nthreads=cores)
        dt = time.perf_counter() - t1
        X = d_X[:n].cpu().numpy()
        U = d_U[:n].cpu().numpy()
        same = (np.array_equal(status[:n], ref["status"]) and np.array_equal(sqp_it[:n], ref["sqp_iter"])
                and np.array_equal(qp_it[:n], ref["qp_iter"]))
        okm = ref["status"] != 1
        same = same and np.array_equal(X[okm], ref["X"][okm]) and np.array_equal(U[okm], ref["U"][okm])
        line["cpu_baseline"] = {
            "value": float((ref["status"] == 0).sum()) / dt, "unit": "OCP/s", "cores": cores,
            "kind": "port",
            "sample": f"first {n} problems of the same batch, CPU restatement (oracle), OpenMP {cores} threads, "
                      f"{dt:.1f} s; all-status rate {n / dt:.0f}/s",
            "gpu_matches_cpu_on_sample": bool(same)}
    print(json.dumps(line))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
