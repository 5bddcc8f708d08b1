#!/usr/bin/env python3
"""Headline benchmark: converged OCPs/sec on the 1M random start/goal batch (N = 50).

One step = one full solve of a rank's problems by the fused HIP kernel, inputs and outputs
resident in HBM.  Rank 0 prints ONE JSON line.

  python bench.py --gpus N --steps K --warmup W

With N > 1 and no launcher around it (WORLD_SIZE unset) this process starts N ranks itself, one
per GPU, before anything touches a GPU, and passes rank 0's line through; under
``torch.distributed.run`` it is one of the ranks.  Either way ``n_gpus`` is the number of ranks
RCCL sees, and a mismatch with --gpus is an error, not a silent single-GPU run.

  --scaling weak    (default) every rank solves its own batch of --batch problems; no data-path
                    collective, only the reduction of the report counters.  Every rank gets the
                    SAME batch (seed 20250310) so that per-GPU work is identical;
                    --distinct-batches gives rank r seed + r (then the slowest rank's batch, i.e.
                    where its longest problem sits, sets the time: per-rank times are in the line).
  --scaling strong  ONE batch of --batch problems, contiguous slices per rank, results gathered to
                    rank 0 over RCCL (device tensors) inside the timed region.

Besides the headline the same line carries, by default:
  N = 1:  "reachable": a few steps of the workload whose goals are reachable by construction (about
          95 % converge; on the headline distribution 91.5 % of the first QPs are infeasible);
          "two_launches_in_flight": the headline batch with two solves in flight (a launch ends on its
          longest problem; the next launch's bulk runs beside that tail);
  N > 1 (weak): "strong": the SAME run repeated as ONE batch of --batch problems cut into contiguous
          slices, one per rank, with the RCCL gather to rank 0 inside the timed region.
"""
import argparse
import ctypes as C
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

FP64_VECTOR_PEAK_TFLOPS = 78.6      # MI355X vector FP64: 256 CUs x 4 SIMDs x 16 lanes x 2 flop x 2.4 GHz
HBM_ACHIEVABLE_MIX_GBS = 5270.0
HBM_PEAK_GBS = 8000.0


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--batch", type=int, default=1_000_000,
                    help="problems per GPU (weak scaling) or in total (strong scaling)")
    ap.add_argument("--scaling", choices=("weak", "strong"), default="weak")
    ap.add_argument("--seed", type=int, default=None, help="batch seed (default: the headline seed)")
    ap.add_argument("--distinct-batches", action="store_true",
                    help="weak scaling: rank r solves seed + r instead of every rank the same batch")
    ap.add_argument("--same-batch", action="store_true", help="(the default now; kept for old command lines)")
    ap.add_argument("--obstacle", action="store_true",
                    help="second workload: one obstacle (r=1, margin 0.5) beside the middle of every path")
    ap.add_argument("--reachable", action="store_true",
                    help="third workload: goals reached by a random admissible input sequence (about 95 %% converge)")
    ap.add_argument("--host-guess", action="store_true",
                    help="upload the host's initial guess instead of letting the kernel compute it")
    ap.add_argument("--order-threshold", type=int, default=0,
                    help="the callers_order block moves problems of this many interior-point iterations or more to the front "
                         "(0: the whole batch sorted by iteration count)")
    ap.add_argument("--order", default=None, help="experiment: permute the batch before upload (align | counts:FILE | reverse; +deal)")
    ap.add_argument("--lib", default=None, help="another build of the library (diagnostic builds)")
    ap.add_argument("--cpu-sample", type=int, default=262144, help="problems timed on the CPU oracle (0 = skip)")
    ap.add_argument("--no-host-to-host", action="store_true", help="skip the host-to-host timing of the batch")
    ap.add_argument("--no-extra", action="store_true",
                    help="headline only: neither the reachable-workload block (N = 1) nor the strong-scaling block (N > 1)")
    ap.add_argument("--extra-steps", type=int, default=3, help="timed steps of those blocks (one warm-up step each)")
    ap.add_argument("--rehearse-on-one-gpu", action="store_true",
                    help="control-flow rehearsal of the multi-rank paths where only one GPU is there: every rank "
                         "uses GPU 0 and the collectives run over gloo on host copies (RCCL refuses two ranks on "
                         "one device); the line says so and is no measurement")
    return ap.parse_args(argv)


def launch_ranks(args):
    """Start args.gpus ranks of this script, one per GPU, and wait for them.  Runs before this
    process has made any GPU call (counting devices does not initialise one)."""
    import torch
    have = torch.cuda.device_count()
    if args.rehearse_on_one_gpu:
        have = args.gpus if have >= 1 else 0
    if have < args.gpus:
        raise SystemExit(f"bench.py --gpus {args.gpus}: only {have} GPU(s) visible to this process; refusing to "
                         f"report a {args.gpus}-GPU number from fewer devices")
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for r in range(args.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.gpus),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env))
    codes = [p.wait() for p in procs]
    if any(codes):
        raise SystemExit(f"bench.py: rank exit codes {codes}")


def make_batch(args, seed, B):
    from mpc_amd.synthetic import reachable_batch, synthetic_batch
    dev_guess = False if args.host_guess else "all"
    if args.reachable:
        return reachable_batch(B, seed=seed, device_guess=dev_guess)
    return synthetic_batch(B, seed=seed, obstacle=args.obstacle, device_guess=dev_guess)


def reorder(args, packed):
    """--order experiments: permute the batch before upload."""
    import numpy as np
    from mpc_amd.batch import F_X0, F_GOAL
    B = packed.B
    sx, sy, sth = packed.pdata[F_X0], packed.pdata[F_X0 + 1], packed.pdata[F_X0 + 2]
    gx, gy, gth = packed.pdata[F_GOAL], packed.pdata[F_GOAL + 1], packed.pdata[F_GOAL + 2]
    phi = np.arctan2(gy - sy, gx - sx)
    wrap = lambda a: (a + np.pi) % (2 * np.pi) - np.pi
    score = np.abs(wrap(phi - sth)) + np.abs(wrap(gth - phi))
    head = args.order.split("+")[0]
    if head == "align":
        perm = np.argsort(score, kind="stable")
    elif head.startswith("counts:"):      # longest first, from the iteration counts a previous run saved
        perm = np.argsort(-np.load(head[7:]).astype(np.int64), kind="stable")
    else:
        perm = np.arange(B)[::-1].copy()
    if args.order.endswith("+deal"):
        n = min(B, 65536) // 1024 * 1024
        perm = np.concatenate([perm[:n].reshape(-1, 1024).T.reshape(-1), perm[n:]])
    return packed.take(perm)


def git_sha_of(path):
    try:
        return subprocess.run(["git", "log", "-n", "1", "--format=%h", "--", path], cwd=ROOT, capture_output=True,
                              text=True, timeout=10).stdout.strip() or None
    except Exception:
        return None


def main():
    args = parse_args()
    env_world = os.environ.get("WORLD_SIZE")
    if env_world is None and args.gpus > 1:
        launch_ranks(args)
        return

    import numpy as np
    import torch
    import torch.distributed as dist
    from mpc_amd import _lib, flops
    from mpc_amd.parallel import gather_to_root, shard_range
    from mpc_amd.synthetic import SEED

    rank = int(os.environ.get("RANK", "0"))
    world = int(env_world or "1")
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"bench.py --gpus {args.gpus} but WORLD_SIZE={world}: start it with --gpus {world} "
                         f"(or without a launcher, and it starts the ranks itself)")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device (there is no CPU fallback)")
    rehearsal = args.rehearse_on_one_gpu and world > 1
    if rehearsal:
        local_rank = 0
    if local_rank >= torch.cuda.device_count():
        raise SystemExit(f"rank {rank}: no GPU {local_rank} (visible: {torch.cuda.device_count()})")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if rehearsal:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        assert dist.get_world_size() == world
    # tensors that go through a collective: device tensors under RCCL, host copies in a rehearsal
    coll = (lambda t: t.cpu()) if rehearsal else (lambda t: t)

    lib = _lib.load_library(args.lib)
    opts = _lib.default_opts(lib)
    opts.max_obs = 1 if args.obstacle else 0

    seed0 = SEED if args.seed is None else args.seed
    strong = args.scaling == "strong"
    stream = torch.cuda.current_stream()

    def prepare(is_strong, reachable=False):
        """This rank's problems of a workload: (packed, B_total)."""
        if reachable:
            from mpc_amd.synthetic import reachable_batch
            return reachable_batch(args.batch, seed=seed0, device_guess=False if args.host_guess else "all"), args.batch * world
        if is_strong:
            whole = make_batch(args, seed0, args.batch)
            if args.order:
                whole = reorder(args, whole)
            lo, hi = shard_range(whole.B, rank, world)
            return whole.slice(lo, hi), whole.B
        packed = make_batch(args, seed0 + (rank if args.distinct_batches else 0), args.batch)
        if args.order:
            packed = reorder(args, packed)
        return packed, packed.B * world

    def barrier():
        if world > 1:
            dist.barrier()

    def measure_in_flight(packed, n_steps, n_warmup, depth=2):
        """The same batch solved n_steps times with `depth` launches in flight: each launch has its own stream,
        workspace and result arrays (a serving loop that double-buffers).  A launch ends with its longest
        problem on a handful of wavefronts; the wavefronts it has freed take up the next launch meanwhile.
        Returns seconds per solve, wall clock between two synchronisations."""
        B, N = packed.B, packed.N
        d_pd = torch.from_numpy(packed.pdata).to(dev)
        d_no = torch.from_numpy(packed.n_obs).to(dev)
        d_xg = torch.from_numpy(packed.xguess).to(dev) if packed.xguess is not None else None
        ws_bytes = lib.mpcx_workspace_bytes(max(B, 1), N)
        sets = []
        for _ in range(depth):
            o = {"X": torch.empty((B, N + 1, 5), dtype=torch.float64, device=dev),
                 "U": torch.empty((B, N, 2), dtype=torch.float64, device=dev),
                 "status": torch.full((B,), -1, dtype=torch.int32, device=dev),
                 "sqp_iter": torch.empty(B, dtype=torch.int32, device=dev),
                 "qp_iter": torch.empty(B, dtype=torch.int32, device=dev),
                 "cost": torch.empty(B, dtype=torch.float64, device=dev),
                 "res": torch.empty((4, B), dtype=torch.float64, device=dev)}
            ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
            rs = _lib.MpcxResult(*(o[k].data_ptr() for k in ("X", "U", "status", "sqp_iter", "qp_iter", "cost", "res")))
            sets.append((o, ws, rs, torch.cuda.Stream(device=dev)))
        bs = _lib.MpcxBatch(B, N, d_pd.data_ptr(), d_no.data_ptr(), d_xg.data_ptr() if d_xg is not None else None, None)

        def launch(i):
            o, ws, rs, st = sets[i % depth]
            rc = lib.mpcx_solve_batch_device(C.byref(bs), C.byref(opts), C.byref(rs), ws.data_ptr(), ws_bytes, st.cuda_stream)
            if rc != 0:
                raise RuntimeError(lib.mpcx_last_error().decode())
        torch.cuda.synchronize()
        for i in range(n_warmup):
            launch(i)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(n_steps):
            launch(i)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        n_conv = [int((o["status"] == 0).sum()) for o, _, _, _ in sets[:min(depth, n_steps)]]
        return dt / max(n_steps, 1), n_conv

    def measure(packed, B_total, is_strong, n_steps, n_warmup, order=None):
        """n_warmup untimed and n_steps timed solves of this rank's problems (plus, strong scaling, the
        gather of every rank's results to rank 0), bracketed by barrier + synchronize; the counters
        reduced over the ranks.  Returns everything the report needs."""
        B, N = packed.B, packed.N
        # inputs resident in HBM before the timed region
        d_pd = torch.from_numpy(packed.pdata).to(dev)
        d_no = torch.from_numpy(packed.n_obs).to(dev)
        d_xg = torch.from_numpy(packed.xguess).to(dev) if packed.xguess is not None else None
        out = {"X": torch.empty((B, N + 1, 5), dtype=torch.float64, device=dev),
               "U": torch.empty((B, N, 2), dtype=torch.float64, device=dev),
               "status": torch.empty(B, dtype=torch.int32, device=dev),
               "sqp_iter": torch.empty(B, dtype=torch.int32, device=dev),
               "qp_iter": torch.empty(B, dtype=torch.int32, device=dev),
               "cost": torch.empty(B, dtype=torch.float64, device=dev),
               "res": torch.empty((4, B), dtype=torch.float64, device=dev)}
        ws_bytes = lib.mpcx_workspace_bytes(max(B, 1), N)
        d_ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
        assert d_ws.data_ptr() % 256 == 0
        d_or = torch.from_numpy(order).to(dev) if order is not None else None      # mpcx_batch.order: a fetch order
        bs = _lib.MpcxBatch(B, N, d_pd.data_ptr(), d_no.data_ptr(), d_xg.data_ptr() if d_xg is not None else None, None,
                            d_or.data_ptr() if d_or is not None else None)
        rs = _lib.MpcxResult(*(out[k].data_ptr() for k in ("X", "U", "status", "sqp_iter", "qp_iter", "cost", "res")))
        gathered = [None]
        gather_ev = []

        def step(timed=False):
            if B > 0:
                rc = lib.mpcx_solve_batch_device(C.byref(bs), C.byref(opts), C.byref(rs),
                                                 d_ws.data_ptr(), ws_bytes, stream.cuda_stream)
                if rc != 0:
                    raise RuntimeError(lib.mpcx_last_error().decode())
            if is_strong and world > 1:
                # the one exchange of the path: every rank's slice of the results to rank 0, over RCCL
                g0, g1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                g0.record(stream)
                gathered[0] = gather_to_root({k: coll(v) for k, v in out.items()} if rehearsal else out, B_total, dst=0)
                g1.record(stream)
                if timed:
                    gather_ev.append((g0, g1))

        for _ in range(n_warmup):
            step()
        torch.cuda.synchronize()
        barrier()
        torch.cuda.synchronize()
        ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n_steps)]
        t0 = time.perf_counter()
        for e0, e1 in ev:
            e0.record(stream)       # the solve is queued on this same stream (passed to the library)
            step(timed=True)
            e1.record(stream)
        torch.cuda.synchronize()
        own_elapsed = time.perf_counter() - t0
        barrier()
        torch.cuda.synchronize()
        elapsed = time.perf_counter() - t0
        step_ms = [e0.elapsed_time(e1) for e0, e1 in ev]
        gather_ms = float(np.mean([a.elapsed_time(b) for a, b in gather_ev])) if gather_ev else 0.0
        kernel_ms = (float(np.mean(step_ms)) - gather_ms) if ev else 0.0

        # outcome counters reduced on the device
        d_stats = torch.zeros(8, dtype=torch.int64, device=dev)
        if B > 0 and lib.mpcx_batch_stats_device(out["status"].data_ptr(), out["sqp_iter"].data_ptr(), out["qp_iter"].data_ptr(),
                                                 B, d_stats.data_ptr(), C.c_void_p(stream.cuda_stream)) != 0:
            raise RuntimeError(lib.mpcx_last_error().decode())
        stats = d_stats.cpu().numpy()
        n_conv = int(stats[0])
        status = out["status"].cpu().numpy()
        sqp_it = out["sqp_iter"].cpu().numpy()
        qp_it = out["qp_iter"].cpu().numpy()
        assert n_conv == int((status == 0).sum())
        fl_dense = flops.flops_from_stats(stats, N)
        fl = flops.flops_from_stats(stats, N, sparse=True)
        assert fl_dense == flops.algorithmic_flops(status, sqp_it, qp_it, N)
        conv = status == 0
        fl_conv = flops.algorithmic_flops(status[conv], sqp_it[conv], qp_it[conv], N, sparse=True)

        tot = torch.tensor([elapsed, float(n_conv), fl, fl_dense, float(B), float(stats[6])], dtype=torch.float64, device=dev)
        per_rank = torch.tensor([own_elapsed * 1e3 / max(n_steps, 1), kernel_ms, float(qp_it.max()) if B else 0.0, float(B)],
                                dtype=torch.float64, device=dev)
        if world > 1:
            mx = coll(tot[:1].clone())
            dist.all_reduce(mx, op=dist.ReduceOp.MAX)
            sm = coll(tot[1:].clone())
            dist.all_reduce(sm, op=dist.ReduceOp.SUM)
            elapsed = float(mx[0])
            n_conv_all, fl_all, fl_dense_all, B_all, qp_all = (float(v) for v in sm)
            per_rank = coll(per_rank)
            pr = [torch.empty_like(per_rank) for _ in range(world)]
            dist.all_gather(pr, per_rank)
            per_rank_rows = [[float(v) for v in t] for t in pr]
        else:
            n_conv_all, fl_all, fl_dense_all, B_all, qp_all = float(n_conv), fl, fl_dense, float(B), float(stats[6])
            per_rank_rows = [[float(v) for v in per_rank]]

        if is_strong and world > 1 and rank == 0:
            # the gathered arrays are the whole batch in the caller's order
            g = gathered[0]
            assert g["status"].shape[0] == B_total
            assert int((g["status"] == 0).sum()) == int(n_conv_all)
            assert torch.equal(g["X"][:B], coll(out["X"]))
        return dict(B=B, N=N, B_total=B_total, packed=packed, out=out, elapsed=elapsed, kernel_ms=kernel_ms, gather_ms=gather_ms,
                    stats=stats, status=status, sqp_it=sqp_it, qp_it=qp_it, n_conv=n_conv, fl=fl, fl_dense=fl_dense,
                    fl_conv=fl_conv, n_conv_all=n_conv_all, fl_all=fl_all, fl_dense_all=fl_dense_all, B_all=B_all,
                    qp_all=qp_all, per_rank_rows=per_rank_rows, steps=max(n_steps, 1))

    def per_rank_list(R):
        return [{"rank": i, "ms_per_step": r[0], "kernel_ms": r[1], "longest_problem_ip_iterations": int(r[2]),
                 "problems": int(r[3])} for i, r in enumerate(R["per_rank_rows"])]

    packed, B_total = prepare(strong)
    R = measure(packed, B_total, strong, args.steps, args.warmup)
    B, N = R["B"], R["N"]
    out, status, sqp_it, qp_it, stats = R["out"], R["status"], R["sqp_it"], R["qp_it"], R["stats"]
    elapsed, kernel_ms, gather_ms = R["elapsed"], R["kernel_ms"], R["gather_ms"]
    n_conv, fl, fl_dense, fl_conv = R["n_conv"], R["fl"], R["fl_dense"], R["fl_conv"]
    n_conv_all, B_all, qp_all = R["n_conv_all"], R["B_all"], R["qp_all"]
    if os.environ.get("MPCX_BENCH_SAVE_COUNTS"):
        np.save(os.environ["MPCX_BENCH_SAVE_COUNTS"], qp_it)

    # the second block of the line (every rank takes part; rank 0 reports)
    headline_workload = not (args.obstacle or args.reachable or args.order or args.host_guess)
    extra = None
    if not args.no_extra and headline_workload and not strong and B > 0:
        if world > 1:
            # the north-star's configuration: ONE batch cut into contiguous slices, results gathered over RCCL
            p2, bt2 = prepare(True)
            S = measure(p2, bt2, True, args.extra_steps, 1)
            extra = ("strong", {
                "what": f"ONE batch of {bt2} problems (seed {seed0}), contiguous slices over {world} GPUs, RCCL gather of "
                        f"every slice's results to rank 0 inside the timed region",
                "value": S["n_conv_all"] * S["steps"] / S["elapsed"], "unit": "OCP/s",
                "ms_per_step": 1e3 * S["elapsed"] / S["steps"], "steps": S["steps"], "warmup": 1,
                "solved_per_sec_all_statuses": S["B_all"] * S["steps"] / S["elapsed"],
                "converged_fraction": S["n_conv_all"] / S["B_all"] if S["B_all"] else 0.0,
                "gather_ms_per_step": S["gather_ms"],
                "gather_bytes": int((bt2 - S["B"]) * flops.result_bytes(S["N"])),
                "per_rank": per_rank_list(S),
                "speedup_over_one_rank_solving_its_weak_batch": (1e3 * elapsed / max(args.steps, 1)) / (1e3 * S["elapsed"] / S["steps"]),
                "note": "a rank's time is bounded below by its longest problem (interior-point iterations x the "
                        "single-problem iteration time), which does not shrink with the slice"})
            del S, p2
        else:
            p2, bt2 = prepare(False, reachable=True)
            S = measure(p2, bt2, False, args.extra_steps, 1)
            secs2 = S["kernel_ms"] * 1e-3
            extra = ("reachable", {
                "what": f"{bt2} problems whose goals are reachable by construction (mpc_amd.synthetic.reachable_batch, "
                        f"seed {seed0}), N={S['N']}: the workload on which `converged per second` means what it says",
                "value": S["n_conv_all"] * S["steps"] / S["elapsed"], "unit": "OCP/s",
                "ms_per_step": 1e3 * S["elapsed"] / S["steps"], "steps": S["steps"], "warmup": 1,
                "converged_fraction": S["n_conv_all"] / S["B_all"] if S["B_all"] else 0.0,
                "solved_per_sec_all_statuses": S["B_all"] * S["steps"] / S["elapsed"],
                "ip_iterations_total": int(S["qp_all"]), "kernel_ms": S["kernel_ms"],
                "frac": (S["fl"] / secs2 / 1e12 / FP64_VECTOR_PEAK_TFLOPS) if secs2 > 0 else 0.0,
                "frac_dense_count": (S["fl_dense"] / secs2 / 1e12 / FP64_VECTOR_PEAK_TFLOPS) if secs2 > 0 else 0.0})
            del S, p2
        torch.cuda.empty_cache()
    in_flight = None
    if not args.no_extra and headline_workload and not strong and world == 1 and B > 0:
        sec, conv2 = measure_in_flight(packed, max(4, 2 * args.extra_steps), 2)
        in_flight = {"what": "the headline batch solved repeatedly with TWO launches in flight (own stream, workspace and result "
                             "arrays each): what a caller who double-buffers batches gets; not the headline value",
                     "ms_per_solve": 1e3 * sec, "converged_per_sec": n_conv / sec,
                     "over_one_launch_at_a_time": (elapsed / max(args.steps, 1)) / sec,
                     "same_converged_count": bool(all(c == n_conv for c in conv2))}
        torch.cuda.empty_cache()

    ordered = None
    if not args.no_extra and not (args.order or args.host_guess) and not strong and world == 1 and B > 0:
        from mpc_amd.batch import longest_first
        order = longest_first(qp_it, threshold=args.order_threshold)
        S = measure(packed, B_total, False, args.extra_steps, 1, order=order)
        ordered = {"what": "the batch above once more with mpcx_batch.order = the problems sorted by the interior-point iteration "
                           "counts of the solve above, longest first (mpc_amd.batch.longest_first"
                           + (f", only those of {args.order_threshold} or more moved" if args.order_threshold > 0 else "")
                           + "): what a caller who has solved the same or a similar batch before gets; not the headline value",
                   "moved_to_the_front": int((qp_it >= args.order_threshold).sum()),
                   "ms_per_step": 1e3 * S["elapsed"] / S["steps"], "kernel_ms": S["kernel_ms"], "steps": S["steps"], "warmup": 1,
                   "converged_per_sec": S["n_conv_all"] * S["steps"] / S["elapsed"],
                   "same_results": bool(np.array_equal(S["status"], status) and np.array_equal(S["qp_it"], qp_it)
                                        and torch.equal(S["out"]["X"], out["X"]) and torch.equal(S["out"]["U"], out["U"]))}
        del S
        torch.cuda.empty_cache()

    if rank != 0:
        if world > 1:
            dist.destroy_process_group()
        return

    steps = max(args.steps, 1)
    value = n_conv_all * steps / elapsed
    secs = kernel_ms * 1e-3
    achieved_tf = fl / secs / 1e12 if secs > 0 else 0.0
    # Memory-side bytes of ONE launch of this batch on one GPU, from the committed counter passes (not
    # measured in this run: counters need rocprofv3 around the process).
    traffic = traffic_src = None
    try:
        with open(os.path.join(ROOT, "profiles", "pmc_traffic.json")) as fh:
            pmc = json.load(fh)
        if (int(pmc["batch"]) == B and not args.obstacle and not args.reachable and args.seed is None
                and not args.order and not (args.distinct_batches and world > 1) and not strong):
            traffic = (float(pmc["read_factor"]) * float(pmc["fetch_kib"]) + float(pmc["write_kib"])) * 1024.0
            traffic_src = {"file": "profiles/pmc_traffic.json", "git": pmc.get("git") or git_sha_of("profiles/pmc_traffic.json"),
                           "passes": pmc.get("source"), "measured_in_this_run": False}
    except (OSError, ValueError, KeyError):
        pass
    io_bytes = flops.algorithmic_io_bytes(N, host_guess=args.host_guess) * B
    workload = ("goals reachable by construction" if args.reachable else
                "one obstacle each" if args.obstacle else "no obstacles")
    if strong:
        what = f"{B_total} random start/goal pose OCPs in total, N={N}, {workload}, seed {seed0}, contiguous slices over {world} GPU(s)"
    else:
        what = (f"{B} random start/goal pose OCPs per GPU, N={N}, {workload}, "
                + (f"seed {seed0}+rank" if args.distinct_batches else f"seed {seed0} on every rank"))
    line = {
        "metric": "converged OCPs/sec (N=50 kinematic car)",
        "value": value, "unit": "OCP/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": 1e3 * elapsed / steps,
        "higher_is_better": True, "scaling": args.scaling, "vs_baseline": None,
        "dtype": "f64", "data": "synthetic",
        **({"rehearsal": "all ranks on one GPU, collectives over gloo: control flow only, not a measurement"} if rehearsal else {}),
        "config": {"workload": what,
                   "problems_per_gpu": B if not strong else None, "N": N, "problems_total": int(B_all),
                   "converged_fraction": n_conv_all / B_all if B_all else 0.0,
                   "solved_per_sec_all_statuses": B_all * steps / elapsed,
                   "ip_iterations_total": int(qp_all),
                   "initial_guess": "host array uploaded" if args.host_guess else "computed in the solve kernel",
                   "parallelism": (f"one batch, contiguous slices x{world}, RCCL gather to rank 0" if strong
                                   else f"independent batch per GPU x{world}, no data-path collective"),
                   "per_rank": per_rank_list(R),
                   **({"gather_ms_per_step": gather_ms,
                       "gather_bytes": int((B_total - B) * flops.result_bytes(N))} if strong and world > 1 else {})},
        "roofline": {
            # what limits the launch: the per-problem working set (67 KB at N = 50) lives in an HBM
            # workspace, so the kernel is bound by the bytes it moves, not by the FP64 units
            "bound": "hbm",
            "bound_detail": "hbm_workspace" if traffic else "hbm_workspace (no committed counter pass for this batch)",
            # the FP64 fraction the metric asks for, on the structure-exploiting operation count (estimated
            # closed forms; frac_dense_count below is on the count taken from an operation-counting build)
            "achieved": achieved_tf, "peak": FP64_VECTOR_PEAK_TFLOPS, "unit": "TFLOP/s",
            "frac": achieved_tf / FP64_VECTOR_PEAK_TFLOPS,
            "flops_per_launch": fl, "flops_constants": flops.constants(sparse=True),
            "flops_per_launch_dense_count": fl_dense, "flops_constants_dense_count": flops.constants(sparse=False),
            "frac_dense_count": (fl_dense / secs / 1e12 / FP64_VECTOR_PEAK_TFLOPS) if secs > 0 else 0.0,
            # the same fraction counting only the work of the problems that converged (rank 0's batch)
            "converged_only_flops_frac": (fl_conv / secs / 1e12 / FP64_VECTOR_PEAK_TFLOPS) if secs > 0 else 0.0,
            "traffic": traffic, "traffic_source": traffic_src,
            "traffic_GBs": traffic / secs / 1e9 if traffic and secs > 0 else None,
            "hbm_peak_GBs": HBM_PEAK_GBS,
            "frac_of_hbm_peak": traffic / secs / 1e9 / HBM_PEAK_GBS if traffic and secs > 0 else None,
            # against what this load/store mix (96 loads + 35 stores of 512 B per stage visit) reaches with no
            # arithmetic at all: profiles/micro/stream_rw.hip, profiles/r01_micro_stream_rw.txt
            "achievable_GBs_this_mix": HBM_ACHIEVABLE_MIX_GBS,
            "frac_of_achievable": traffic / secs / 1e9 / HBM_ACHIEVABLE_MIX_GBS if traffic and secs > 0 else None,
            "algorithmic_bytes_per_launch": io_bytes,
            "traffic_over_algorithmic_bytes": traffic / io_bytes if traffic and io_bytes else None,
            "kernel": "mpcx_sqp_kernel<1>" if args.obstacle else "mpcx_sqp_kernel<0>", "kernel_ms": kernel_ms,
            "kernel_ms_source": "HIP events on the launch stream around each step, mean over the timed steps (rank 0)",
            "batch_io_GBs": io_bytes / secs / 1e9 if secs > 0 else 0.0},
    }

    if extra:
        line[extra[0]] = extra[1]
    if in_flight:
        line["two_launches_in_flight"] = in_flight
    if ordered:
        line["callers_order"] = ordered

    if world == 1 and not args.no_host_to_host and B > 0:
        # the whole call as a host caller sees it: host tables in, results and counters in host memory
        # (page-locked arrays of the handle), buffers kept from the previous call
        with _lib.Context(local_rank) as ctx:
            ctx.solve(packed.slice(0, min(B, 1024)), opts)          # first call: creates stream, small buffers
            t1 = time.perf_counter()
            res = ctx.solve(packed, opts)
            first_ms = 1e3 * (time.perf_counter() - t1)
            t1 = time.perf_counter()
            res = ctx.solve(packed, opts)
            h2h_ms = 1e3 * (time.perf_counter() - t1)
            same = bool(np.array_equal(res["status"], status) and np.array_equal(res["qp_iter"], qp_it)
                        and int(ctx.stats[0]) == n_conv)
        line["host_to_host"] = {"ms": h2h_ms, "ms_first_call_with_allocation": first_ms,
                                "converged_per_sec": n_conv / (h2h_ms * 1e-3),
                                "over_kernel": h2h_ms / kernel_ms if kernel_ms > 0 else None,
                                "what": "mpcx_ctx_solve: host problem table in (pageable), X/U/status/counters in "
                                        "page-locked host memory; not the headline value",
                                "same_results": same}

    if args.cpu_sample > 0 and world == 1 and B > 0:
        import oracle_lib
        n = min(args.cpu_sample, B)
        avail = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
        # the box's share of its host is 16 threads (OMP_NUM_THREADS there); never more than that
        cores = max(1, min(avail, int(os.environ.get("OMP_NUM_THREADS", "16") or 16)))
        sample = packed.slice(0, n)
        fast = oracle_lib.ORACLE_FAST_SO if os.path.exists(oracle_lib.ORACLE_FAST_SO) else oracle_lib.ORACLE_SO
        t1 = time.perf_counter()
        ref = oracle_lib.solve_packed(sample, nthreads=cores, lib=oracle_lib.load(fast))
        dt = time.perf_counter() - t1
        X = out["X"][:n].cpu().numpy()
        U = out["U"][:n].cpu().numpy()
        same = (np.array_equal(status[:n], ref["status"]) and np.array_equal(sqp_it[:n], ref["sqp_iter"])
                and np.array_equal(qp_it[:n], ref["qp_iter"]))
        okm = ref["status"] != 1
        same = same and np.array_equal(X[okm], ref["X"][okm]) and np.array_equal(U[okm], ref["U"][okm])
        line["cpu_baseline"] = {
            "value": float((ref["status"] == 0).sum()) / dt, "unit": "OCP/s", "cores": cores,
            "cores_available": avail, "cores_used": cores,
            "kind": "port",
            "build": os.path.basename(fast) + " (gcc -O3 -mavx2 -mfma, FP contraction off: still the bit-exact checker)",
            "sample": f"first {n} problems of the same batch, CPU restatement (oracle), OpenMP {cores} threads, "
                      f"{dt:.1f} s; all-status rate {n / dt:.0f}/s",
            "gpu_matches_cpu_on_sample": bool(same)}
    print(json.dumps(line))
    sys.stdout.flush()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
