"""GPU parity: the HIP path through the C ABI against the CPU oracle.

Bar: integer outputs (status, SQP and QP iteration counts) equal for every
problem; states, controls, cost and residuals bit-identical (the kernels and
the oracle perform the same IEEE operations in the same order).  North-star
tolerance would be 1e-6 absolute on states/controls and 1e-8 relative on
cost; bit equality is stricter and is what is asserted.
"""
import ctypes as C

import numpy as np
import pytest
import torch  # noqa: F401  first, so that the process has ONE HIP runtime (torch ships its own; the
              # receding-horizon test below shares device memory between torch and libmpcx)

from mpc_amd import _lib
from mpc_amd.batch import pack_problems
from mpc_amd.core_solver import solve_groups
from mpc_amd.synthetic import reachable_batch, synthetic_batch

import cases

pytestmark = pytest.mark.gpu


def test_library_sees_device():
    lib = _lib.load_library()
    assert lib.mpcx_device_count() >= 1


def test_sincos_bits(oracle):
    lib = _lib.load_library()
    rng = np.random.default_rng(7)
    x = np.concatenate([rng.uniform(-10, 10, 20000), rng.uniform(-1e3, 1e3, 2000),
                        np.array([0.0, -0.0, np.pi, -np.pi, np.pi / 2, np.pi / 4, 1e6, 2e6, np.inf, np.nan])])
    s = np.empty_like(x)
    c = np.empty_like(x)
    rc = lib.mpcx_probe_sincos(x.ctypes.data, x.size, s.ctypes.data, c.ctypes.data, 0)
    assert rc == 0, lib.mpcx_last_error()
    olib = oracle.load()
    so, co = C.c_double(), C.c_double()
    for i in range(x.size):
        olib.mpcx_oracle_sincos(float(x[i]), C.byref(so), C.byref(co))
        assert (s[i] == so.value or (np.isnan(s[i]) and np.isnan(so.value))), (x[i], s[i], so.value)
        assert (c[i] == co.value or (np.isnan(c[i]) and np.isnan(co.value))), (x[i], c[i], co.value)


@pytest.mark.parametrize("B", [1, 63, 64, 65, 1000, 20000])
def test_synthetic_batch_matches_oracle(oracle, B):
    packed = synthetic_batch(B)
    gpu = _lib.solve_packed(packed)
    ref = oracle.solve_packed(packed)
    cases.assert_same_solution(gpu, ref, f"synthetic B={B}")
    assert (ref["status"] == 0).any() or B < 20


@pytest.mark.parametrize("B", [0, 1, 64, 100003])
def test_device_batch_stats(B):
    """Outcome counters reduced on the device against numpy on the same arrays (any int32 values
    the solver can write: statuses 0..4, iteration counts up to 200 and 10000)."""
    import ctypes as C
    import torch
    from mpc_amd import flops
    lib = _lib.load_library()
    rng = np.random.default_rng(B + 11)
    status = rng.integers(0, 5, B).astype(np.int32)
    sqp = rng.integers(0, 201, B).astype(np.int32)
    qp = rng.integers(0, 10001, B).astype(np.int32)
    dev = torch.device("cuda", 0)
    d_st, d_sq, d_qp = (torch.from_numpy(a).to(dev) for a in (status, sqp, qp))
    d_out = torch.full((8,), -1, dtype=torch.int64, device=dev)
    rc = lib.mpcx_batch_stats_device(d_st.data_ptr(), d_sq.data_ptr(), d_qp.data_ptr(), B, d_out.data_ptr(),
                                     C.c_void_p(torch.cuda.current_stream().cuda_stream))
    assert rc == 0, lib.mpcx_last_error()
    got = d_out.cpu().numpy()
    want = [int((status == k).sum()) for k in range(5)] + [int(sqp.sum(dtype=np.int64)), int(qp.sum(dtype=np.int64)),
                                                           int(sqp.sum(dtype=np.int64) + (status != 4).sum())]
    assert got.tolist() == want
    assert flops.flops_from_stats(got, 50) == flops.algorithmic_flops(status, sqp, qp, 50)


def test_full_size_properties():
    """Properties that need no oracle, on a batch too large for it (300 000 problems, N = 50):
    every problem's result is the same whatever else is in the batch and wherever it sits (a
    permuted batch gives the permuted results, bit for bit, and a re-run the same bits); every
    converged trajectory starts at the start state, closes the RK4 step of every interval and
    ends in the terminal box to the solver's tolerance, inside the input bounds."""
    from mpc_amd.batch import F_X0, F_GOAL, F_LBU, F_UBU, F_H, F_L
    from mpc_amd.rti import plant_step
    B = 300_000
    packed = synthetic_batch(B)
    a = _lib.solve_packed(packed)
    rng = np.random.default_rng(7)
    perm = rng.permutation(B)
    shuffled = type(packed)(B, packed.N, np.ascontiguousarray(packed.pdata[:, perm]),
                            np.ascontiguousarray(packed.n_obs[perm]),
                            np.ascontiguousarray(packed.xguess[:, :, perm]), packed.duration)
    b = _lib.solve_packed(shuffled)
    again = _lib.solve_packed(packed)
    cases.assert_same_solution(again, a, "second run")
    b_back = {k: (v[:, np.argsort(perm)] if k == "res" else v[np.argsort(perm)]) for k, v in b.items()}
    cases.assert_same_solution(b_back, a, "permuted batch")
    assert set(np.unique(a["status"])) <= {0, 2, 4}
    ok = np.nonzero(a["status"] == 0)[0]
    assert ok.size > 0.05 * B
    X, U = a["X"][ok], a["U"][ok]
    pd = packed.pdata[:, ok]
    tol = 1e-3
    assert np.array_equal(X[:, 0, :], pd[F_X0:F_X0 + 5].T)
    gap = 0.0
    for k in range(packed.N):
        xn = plant_step(np, X[:, k, :], U[:, k, :], pd[F_H], pd[F_L])
        gap = max(gap, float(np.abs(xn - X[:, k + 1, :]).max()))
    assert gap < tol, gap
    eps = np.array([0.1, 0.1, 0.01, 0.01, 0.01])
    assert (np.abs(X[:, -1, :] - pd[F_GOAL:F_GOAL + 5].T) <= eps + tol).all()
    assert (U >= pd[F_LBU:F_LBU + 2].T[:, None, :] - tol).all() and (U <= pd[F_UBU:F_UBU + 2].T[:, None, :] + tol).all()
    assert (a["res"][:, ok] < tol).all()


def test_fetch_order_leaves_no_trace(oracle, monkeypatch):
    """One full block of 65536 problems plus a ragged one: the dealt fetch order (default) and the
    caller's order (MPCX_DEAL=0) give the oracle's results at every problem's own index."""
    B = 65536 + 1237
    packed = synthetic_batch(B)
    dealt = _lib.solve_packed(packed)
    monkeypatch.setenv("MPCX_DEAL", "0")
    plain = _lib.solve_packed(packed)
    monkeypatch.delenv("MPCX_DEAL")
    cases.assert_same_solution(dealt, plain, "dealt fetch vs caller's order")
    ref = oracle.solve_packed(packed)
    cases.assert_same_solution(dealt, ref, f"synthetic B={B}")


@pytest.mark.parametrize("lanes", [None, "64"])
def test_callers_order_leaves_no_trace(oracle, monkeypatch, lanes):
    """mpcx_batch.order: the problems taken back to front, in a random order and longest first (from the
    iteration counts of the first solve) give the results of the library's own order, the oracle's, at
    every problem's own index; through the solver handle as well.  With one problem at a time per wavefront
    (the list taken as it stands) and with 64 (the list dealt over the wavefronts, 32 entries left over)."""
    import dataclasses
    from mpc_amd.batch import longest_first
    if lanes is not None:
        monkeypatch.setenv("MPCX_LANES", lanes)
    B = 20000
    packed = synthetic_batch(B)
    base = _lib.solve_packed(packed)
    rng = np.random.default_rng(3)
    orders = {"back to front": np.arange(B - 1, -1, -1, dtype=np.int32),
              "random": np.ascontiguousarray(rng.permutation(B), dtype=np.int32),
              "whole batch sorted": longest_first(base["qp_iter"]),
              "longest first": longest_first(base["qp_iter"], threshold=300)}
    assert (np.diff(base["qp_iter"][orders["whole batch sorted"]]) <= 0).all()
    lf = orders["longest first"]
    assert sorted(lf.tolist()) == list(range(B)) and base["qp_iter"][lf[0]] == base["qp_iter"].max()
    for name, order in orders.items():
        other = _lib.solve_packed(dataclasses.replace(packed, order=order))
        cases.assert_same_solution(base, other, f"order {name} vs the library's")
    with _lib.Context() as ctx:
        other = ctx.solve(dataclasses.replace(packed, order=orders["random"]))
        cases.assert_same_solution(base, other, "order random through the handle")
    ref = oracle.solve_packed(packed)
    cases.assert_same_solution(base, ref, f"synthetic B={B}")


def test_callers_order_is_checked():
    """An order that is no permutation is refused by the host entry points; the device entry point cannot
    check and skips entries that name no problem: the problems named are solved, the others' rows untouched."""
    import dataclasses
    import torch
    B = 300
    packed = synthetic_batch(B)
    twice = np.arange(B, dtype=np.int32); twice[7] = 8
    with pytest.raises(_lib.MpcxError, match="twice"):
        _lib.solve_packed(dataclasses.replace(packed, order=twice))
    outside = np.arange(B, dtype=np.int32); outside[0] = B
    with pytest.raises(_lib.MpcxError, match="outside"):
        _lib.solve_packed(dataclasses.replace(packed, order=outside))
    with _lib.Context() as ctx, pytest.raises(_lib.MpcxError, match="outside"):
        ctx.solve(dataclasses.replace(packed, order=outside))

    base = _lib.solve_packed(packed)
    lib = _lib.load_library()
    dev = torch.device("cuda:0")
    order = np.arange(B, dtype=np.int32)[::-1].copy()
    skipped = [5, 64, 299]
    for i, b in enumerate(skipped):
        order[B - 1 - b] = (-1, B, 2 ** 31 - 1)[i]
    d_pd, d_no, d_or = (torch.from_numpy(a).to(dev) for a in (packed.pdata, packed.n_obs, order))
    N = packed.N
    out = {"X": torch.full((B, N + 1, 5), 7.0, dtype=torch.float64, device=dev),
           "U": torch.full((B, N, 2), 7.0, dtype=torch.float64, device=dev),
           "status": torch.full((B,), -5, dtype=torch.int32, device=dev),
           "sqp_iter": torch.zeros(B, dtype=torch.int32, device=dev), "qp_iter": torch.zeros(B, dtype=torch.int32, device=dev),
           "cost": torch.zeros(B, dtype=torch.float64, device=dev), "res": torch.zeros((4, B), dtype=torch.float64, device=dev)}
    ws_bytes = lib.mpcx_workspace_bytes(B, N)
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
    d_xg = None if packed.xguess is None else torch.from_numpy(packed.xguess).to(dev)
    bs = _lib.MpcxBatch(B, N, d_pd.data_ptr(), d_no.data_ptr(), None if d_xg is None else d_xg.data_ptr(), None,
                        d_or.data_ptr())
    rs = _lib.MpcxResult(*(out[k].data_ptr() for k in ("X", "U", "status", "sqp_iter", "qp_iter", "cost", "res")))
    opts = _lib.default_opts(lib)
    opts.max_obs = 0
    rc = lib.mpcx_solve_batch_device(C.byref(bs), C.byref(opts), C.byref(rs), ws.data_ptr(), ws_bytes, None)
    assert rc == 0, lib.mpcx_last_error()
    torch.cuda.synchronize()
    status = out["status"].cpu().numpy()
    named = np.setdiff1d(np.arange(B), skipped)
    assert (status[skipped] == -5).all() and (out["X"][skipped] == 7.0).all()
    assert np.array_equal(status[named], base["status"][named])
    assert np.array_equal(out["X"].cpu().numpy()[named], base["X"][named])
    assert np.array_equal(out["U"].cpu().numpy()[named], base["U"][named])


@pytest.mark.parametrize("hold", ["0,0", "0.5,8", "1e-3,1000"])
def test_hold_policy_leaves_no_trace(oracle, monkeypatch, hold):
    """Which problems are aged into single-problem mode is scheduling only: no ageing hold (0,0), a hold
    that keeps almost every QP in the one-lane-per-problem sweeps for 8 more iterations, and one that never
    releases a collapsed QP give the default's results, which are the oracle's."""
    B = 40000
    packed = synthetic_batch(B)
    base = _lib.solve_packed(packed)
    monkeypatch.setenv("MPCX_HOLD", hold)
    other = _lib.solve_packed(packed)
    monkeypatch.delenv("MPCX_HOLD")
    cases.assert_same_solution(base, other, f"hold policy {hold} vs default")
    if hold == "0,0":
        ref = oracle.solve_packed(packed)
        cases.assert_same_solution(base, ref, f"synthetic B={B}")


@pytest.mark.parametrize("B", [64, 3000])
def test_synthetic_obstacle_batch_matches_oracle(oracle, B):
    """The benchmark's second workload: one obstacle per problem beside the middle of the path."""
    packed = synthetic_batch(B, obstacle=True)
    assert (packed.n_obs == 1).all()
    gpu = _lib.solve_packed(packed)
    ref = oracle.solve_packed(packed)
    cases.assert_same_solution(gpu, ref, f"synthetic with obstacle B={B}")
    if B >= 1000:
        assert (ref["status"] == 0).any()


def test_reachable_batch_matches_oracle(oracle):
    """The benchmark's third workload: goals reached by an admissible input sequence, so most
    problems converge and need tens of interior-point iterations (ageing, bursts, regions)."""
    from mpc_amd.synthetic import reachable_batch
    packed = reachable_batch(3000)
    gpu = _lib.solve_packed(packed)
    ref = oracle.solve_packed(packed)
    cases.assert_same_solution(gpu, ref, "reachable B=3000")
    assert (ref["status"] == 0).mean() > 0.9


def test_scenarios_match_oracle(oracle):
    names = list(cases.SCENARIOS)
    egos, sims = zip(*(cases.make_scenario(n) for n in names))
    st_g, si_g, qi_g, res_g = solve_groups(list(egos), list(sims))
    st_o, si_o, qi_o, res_o = solve_groups(list(egos), list(sims), solver=oracle.solve_packed)
    assert (st_g == st_o).all() and (si_g == si_o).all() and (qi_g == qi_o).all()
    assert (st_g == 0).all(), dict(zip(names, st_g))
    for n, a, b in zip(names, res_g, res_o):
        assert np.array_equal(a["x"], b["x"]), n
        assert np.array_equal(a["u"], b["u"]), n


def test_grid_sweep_matches_oracle(oracle):
    egos, sim = cases.grid_problems()
    packed = pack_problems(egos, sim)
    gpu = _lib.solve_packed(packed)
    ref = oracle.solve_packed(packed)
    cases.assert_same_solution(gpu, ref, "grid 32 x N=800")
    assert (gpu["status"] == 0).sum() == 32     # the reference's published outcome


def test_failures_match_oracle(oracle):
    """Unreachable goals: statuses and iteration counts must still agree."""
    rng = np.random.default_rng(11)
    B = 128
    start = np.zeros((B, 5))
    goal = np.zeros((B, 5))
    start[:, 2] = rng.uniform(-3, 3, B)
    goal[:, 0] = rng.uniform(-60, 60, B)      # far beyond what 5 s allow
    goal[:, 1] = rng.uniform(-60, 60, B)
    goal[:, 2] = rng.uniform(-3, 3, B)
    from mpc_amd.batch import pack_arrays
    packed = pack_arrays(start, goal, 5.0)
    gpu = _lib.solve_packed(packed)
    ref = oracle.solve_packed(packed)
    cases.assert_same_solution(gpu, ref, "infeasible batch")
    assert (ref["status"] != 0).all()


@pytest.mark.parametrize("duration", [0.2, 5.0, 6.35, 6.4, 6.5, 13.0])
@pytest.mark.parametrize("lanes", [1, 3, 64])
def test_sweep_modes_match_oracle(oracle, monkeypatch, duration, lanes):
    """The same problems through the one-lane-per-problem sweeps (64 per wavefront) and
    through the single-problem sweeps (1 or 3 per wavefront), horizons around the 64-stage
    block size (N = 2, 50, 63, 64, 65, 130), obstacle rows included."""
    egos, sim = cases.obstacle_problems(96, duration)
    packed = pack_problems(egos, sim)
    assert packed.N == int(duration / 0.1)
    monkeypatch.setenv("MPCX_LANES", str(lanes))
    gpu = _lib.solve_packed(packed)
    ref = oracle.solve_packed(packed)
    cases.assert_same_solution(gpu, ref, f"N={packed.N} lanes={lanes}")


@pytest.mark.parametrize("lanes", [None, 1, 3, 20, 64])
def test_line_search_matches_oracle(oracle, monkeypatch, lanes):
    """mpcx_opts.globalization = 1 (backtracking on the exact penalty function; the reference takes full
    steps, so this has no counterpart there): device = oracle bit for bit through every scheduling path,
    on batches where most steps are taken in full and on scenarios where the search shortens steps."""
    if lanes is not None:
        monkeypatch.setenv("MPCX_LANES", str(lanes))
    opts = _lib.default_opts()
    opts.globalization = 1
    oopts = _lib.default_opts()
    oopts.globalization = 1
    for packed in (synthetic_batch(1500, seed=5), reachable_batch(600, seed=6), synthetic_batch(300, seed=7, obstacle=True)):
        gpu = _lib.solve_packed(packed, opts=opts)
        ref = oracle.solve_packed(packed, opts=oopts)
        cases.assert_same_solution(gpu, ref, f"line search, lanes={lanes}")
    if lanes in (None, 64):
        changed = 0
        for name in ("line", "B_to_C", "corner_turn", "u_turn"):      # N = 150 ... 400: blocks of 64 stages
            ego, sim = cases.make_scenario(name)
            p = pack_problems([ego], sim)
            g = _lib.solve_packed(p, opts=opts)
            r = oracle.solve_packed(p, opts=oopts)
            cases.assert_same_solution(g, r, f"line search, {name}")
            changed += int(r["sqp_iter"][0] != oracle.solve_packed(p)["sqp_iter"][0])
        assert changed >= 1          # the search did shorten steps somewhere


@pytest.mark.parametrize("burst", ["0", "1", "3", "8"])
def test_burst_lengths_leave_no_trace(oracle, monkeypatch, burst):
    """Single-problem iterations run back to back with the stages kept in registers (bursts of
    1, 3, 8 iterations) or as four sweeps each (0): the same bits as the oracle every way, on a
    batch small enough to start in single-problem mode and on one large enough for its old
    problems to be aged into it."""
    monkeypatch.setenv("MPCX_BURST", burst)
    for B in (48, 3000):
        packed = synthetic_batch(B, seed=77 + B)
        gpu = _lib.solve_packed(packed)
        ref = oracle.solve_packed(packed)
        cases.assert_same_solution(gpu, ref, f"burst={burst} B={B}")


@pytest.mark.parametrize("spread", ["0", "1"])
def test_spread_recursion_leaves_no_trace(oracle, monkeypatch, spread):
    """The Riccati recursion of a burst as two lane pipelines (0) or with the algebra of one stage spread
    over the lanes (1, the default; its staging area holds 25 / 23 / 18 / 13 / 25 stages at a time with 0..4
    obstacle rows): the same bits as the oracle both ways, on batches small enough to be solved in bursts
    throughout, with horizons below one staging chunk, of two chunks and at the last size a burst takes."""
    monkeypatch.setenv("MPCX_SPREAD", spread)
    batches = [("synthetic", synthetic_batch(48, seed=31)), ("reachable", reachable_batch(600, seed=32))]
    for k in (1, 2, 3, 4):
        batches.append((f"up to {k} obstacles", pack_problems(*cases.obstacle_problems(64, 5.0, seed=40 + k, max_obs=k))))
    for duration in (0.2, 2.0, 6.3):
        batches.append((f"N={int(duration / 0.1)}", pack_problems(*cases.obstacle_problems(48, duration, seed=50, max_obs=1))))
    for name, packed in batches:
        gpu = _lib.solve_packed(packed)
        ref = oracle.solve_packed(packed)
        cases.assert_same_solution(gpu, ref, f"spread={spread} {name}")


@pytest.mark.parametrize("workers", ["0", "6,64", "20,1", "12,2,16"])
def test_handover_leaves_no_trace(oracle, monkeypatch, workers):
    """Old problems handed over to worker wavefronts (on by default for grids of 256 wavefronts and more; here
    forced on a grid of 47 with 6 workers, and with 20 workers and a queue limit of one so that wavefronts also
    keep problems for themselves, and with kept problems offered again from 16 iterations on instead of 128)
    or given their bursts by the wavefront that fetched them (0): the same bits as
    the oracle, on the benchmark distribution, on a batch where nearly every problem grows old, and with
    obstacle rows (up to 4, 2 and 1 per problem) at horizons of 50, 20 and 62 intervals."""
    monkeypatch.setenv("MPCX_LANES", "64")
    monkeypatch.setenv("MPCX_HANDOVER", workers)
    batches = [("synthetic", synthetic_batch(3000, seed=91)), ("reachable", reachable_batch(2500, seed=92))]
    for duration, k in ((5.0, 4), (2.0, 2), (6.3, 1)):
        packed = pack_problems(*cases.obstacle_problems(1500, duration, seed=60 + k, max_obs=k))
        batches.append((f"up to {k} obstacles, N={packed.N}", packed))
    for name, packed in batches:
        gpu = _lib.solve_packed(packed)
        ref = oracle.solve_packed(packed)
        cases.assert_same_solution(gpu, ref, f"handover={workers} {name}")


@pytest.mark.parametrize("lanes", [None, 3, 20, 64])
def test_rti_server_matches_oracle(oracle, monkeypatch, lanes):
    """Three receding-horizon ticks of a fleet: what the device solves each tick equals the
    oracle's one-iteration (rti) solve of the same tick inputs, bit for bit; the shift and the
    plant step move the fleet towards the goal.  Through every scheduling path: one vehicle per
    wavefront (per-problem regions), a few (single-problem sweeps, regions), 20 and 64 (one lane
    per vehicle)."""
    from mpc_amd.rti import RtiServer
    if lanes is not None:
        monkeypatch.setenv("MPCX_LANES", str(lanes))
    packed = synthetic_batch(4096)
    full = _lib.solve_packed(packed)
    ok = np.nonzero(full["status"] == 0)[0][:300]
    assert ok.size == 300
    sub = packed.slice(0, packed.B)
    pick = lambda a, ax: np.ascontiguousarray(np.take(a, ok, axis=ax))
    fleet = type(packed)(ok.size, packed.N, pick(packed.pdata, 1), pick(packed.n_obs, 0),
                         pick(packed.xguess, 2), packed.duration)
    srv = RtiServer(fleet, X=full["X"][ok], U=full["U"][ok])
    oopts = _lib.default_opts()
    oopts.rti = 1
    d0 = None
    for tick in range(3):
        inp = srv.host_inputs()
        ref = oracle.solve_packed(inp, opts=oopts)
        srv.solve()
        torch.cuda.synchronize()
        gpu = {"X": srv.X.cpu().numpy(), "U": srv.U.cpu().numpy(), "status": srv.status.cpu().numpy(),
               "sqp_iter": srv.sqp_iter.cpu().numpy(), "qp_iter": srv.qp_iter.cpu().numpy(),
               "cost": srv.cost.cpu().numpy(), "res": srv.res.cpu().numpy()}
        cases.assert_same_solution(gpu, ref, f"rti tick {tick}")
        assert (gpu["sqp_iter"] == 1).all() and (gpu["status"] == 0).all()
        u0 = srv.advance().cpu().numpy()
        assert np.array_equal(u0, gpu["U"][:, 0, :])
        x0 = srv.pdata[0:5].t().cpu().numpy()
        d = np.hypot(x0[:, 0] - fleet.pdata[5], x0[:, 1] - fleet.pdata[6])
        if d0 is not None:
            assert d.mean() < d0.mean()
        d0 = d


def test_edge_shapes_match_oracle(oracle, monkeypatch):
    """Shortest horizon (N = 1), the maximum of four obstacles per problem, an empty batch,
    and tight iteration limits (statuses 2 and QP iteration caps must agree too)."""
    from mpc_amd.core_solver import EgoConfig, SimConfig
    # N = 1
    egos, sim = cases.obstacle_problems(40, 0.1)
    packed = pack_problems(egos, sim)
    assert packed.N == 1
    for lanes in (1, 64):
        monkeypatch.setenv("MPCX_LANES", str(lanes))
        cases.assert_same_solution(_lib.solve_packed(packed), oracle.solve_packed(packed), f"N=1 lanes={lanes}")
    # four obstacles
    rng = np.random.default_rng(2)
    egos = []
    for b in range(48):
        e = EgoConfig()
        e.state_start = [0.0, 0.0, 0.0, 1.0, 0.0]
        e.state_final = [12.0, rng.uniform(-1, 1), 0.0, 1.0, 0.0]
        e.corridor_width = 6.0
        e.obstacles = [[3.0 * (j + 1) - 1.0, (1 if (j + b) % 2 else -1) * rng.uniform(1.2, 3.0), 0.5, 0.3]
                       for j in range(4)]
        e.verbose = False
        egos.append(e)
    sim = SimConfig()
    sim.duration = 8.0
    packed = pack_problems(egos, sim)
    assert packed.n_obs.max() == 4
    ref = oracle.solve_packed(packed)
    assert (ref["status"] == 0).sum() >= 10
    for lanes in (2, 64):
        monkeypatch.setenv("MPCX_LANES", str(lanes))
        cases.assert_same_solution(_lib.solve_packed(packed), ref, f"4 obstacles lanes={lanes}")
    monkeypatch.delenv("MPCX_LANES")
    # tight limits
    packed = synthetic_batch(500)
    opts = _lib.default_opts()
    opts.max_sqp_iter = 3
    opts.max_qp_iter = 7
    gpu = _lib.solve_packed(packed, opts=opts)
    ref = oracle.solve_packed(packed, opts=opts)
    cases.assert_same_solution(gpu, ref, "tight limits")
    assert (ref["status"] == 2).any()
    # empty batch
    out = _lib.solve_packed(packed.slice(0, 0))
    assert out["X"].shape == (0, packed.N + 1, 5) and out["status"].shape == (0,)


def test_device_initial_guess_is_the_host_loop():
    """The guess computed on the device from the problem table equals the host restatement of
    reference core_solver.py:477-499 bit for bit (obstacle-free problems)."""
    from mpc_amd.batch import pack_arrays
    rng = np.random.default_rng(4)
    B = 5000
    start = rng.uniform(-30, 30, (B, 5)); goal = rng.uniform(-30, 30, (B, 5))
    start[:, 2] = rng.uniform(-4, 4, B); goal[:, 2] = rng.uniform(-4, 4, B)
    for duration, width in ((5.0, 15.0), (12.3, 2.0)):
        packed = pack_arrays(start, goal, duration, corridor_width=width)
        got = _lib.initial_guess_on_device(packed.pdata, np.full(B, width), packed.N)
        assert got.shape == packed.xguess.shape
        assert np.array_equal(got, packed.xguess)


def test_obstacle_batch_all_scheduling_paths(oracle, monkeypatch):
    """2500 problems with one or two obstacles, 64 per wavefront with refill: the one-lane-per-
    problem sweeps for the young problems, ageing into per-problem regions, the end-of-batch move,
    all in the kernel variants with obstacle rows, against the oracle."""
    egos, sim = cases.obstacle_problems(2500, 5.0, seed=9)
    packed = pack_problems(egos, sim)
    ref = oracle.solve_packed(packed)
    assert (ref["status"] == 0).sum() > 50 and ref["qp_iter"].max() > 100
    for lanes in ("64", "16"):
        monkeypatch.setenv("MPCX_LANES", lanes)
        cases.assert_same_solution(_lib.solve_packed(packed), ref, f"obstacle batch, {lanes} per wavefront")
    monkeypatch.setenv("MPCX_BOOST", "0,0,0")          # no ageing: everything in the wavefront's ticks
    cases.assert_same_solution(_lib.solve_packed(packed), ref, "obstacle batch, no ageing")
