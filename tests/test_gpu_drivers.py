"""GPU: the product entry points with their DEFAULT solver (the HIP library) against the same
call with the CPU oracle as solver.

What is compared is what a user of the reference sees: the result dicts of ``find_path``
(reference src/core_solver.py:527-562), the list ``grid_path_planning`` returns
(src/grid_path_plan.py:59-86), the stitched plan of ``run_sequential_planning``
(src/scenarios.py:96-149) and the files ``scenarios.main`` writes (:363-381).  Bar: same entries
in the same order, arrays bit-identical, files byte-identical.
"""
import numpy as np
import pytest
import torch  # noqa: F401  first: one HIP runtime per process

from mpc_amd import _lib
from mpc_amd import scenarios as sc
from mpc_amd.batch import pack_problems
from mpc_amd.core_solver import EgoConfig, SimConfig, find_path, find_paths, solve_groups
from mpc_amd.grid_path_plan import grid_path_planning, grid_start_states
from mpc_amd.synthetic import synthetic_batch

import cases

pytestmark = pytest.mark.gpu


def _same_result(a, b, what):
    assert (a is None) == (b is None), what
    if a is None:
        return
    assert set(a) == set(b) == {"t", "x", "u", "status"}, what
    assert a["status"] == b["status"] == 0, what
    for k in ("t", "x", "u"):
        assert a[k].shape == b[k].shape and np.array_equal(a[k], b[k]), f"{what}: {k}"


def test_find_path_default_solver(oracle, capfd):
    """verbose False: the HIP call runs inside the fd-level silencer; success and failure."""
    for name in ("line", "path_finder", "corner_turn"):
        ego, sim = cases.make_scenario(name)
        assert ego.verbose is False
        got = find_path(ego, sim)
        want = find_path(ego, sim, solver=oracle.solve_packed)
        assert got is not None
        _same_result(got, want, name)
    assert capfd.readouterr().out == ""
    far = EgoConfig()
    far.state_start = [0, 0, 0, 0, 0]
    far.state_final = [90, 90, 0, 0, 0]
    far.verbose = False
    sim = SimConfig()
    sim.duration = 5.0
    assert find_path(far, sim) is None and find_path(far, sim, solver=oracle.solve_packed) is None
    # a verbose problem prints the reference's lines around the HIP call
    ego, sim = cases.make_scenario("line")
    ego.verbose = True
    assert find_path(ego, sim) is not None
    assert "✓ Path planning successful" in capfd.readouterr().out
    # a batch with a problem that cannot be set up: the others are solved on the device
    same = EgoConfig()
    same.state_start = [1, 1, 0, 0, 0]
    same.state_final = [1, 1, 0, 0, 0]
    same.verbose = False
    ego.verbose = False
    res = find_paths([ego, same, ego], sim)
    assert res[1] is None
    _same_result(res[0], find_path(ego, sim, solver=oracle.solve_packed), "beside a bad problem")
    _same_result(res[2], res[0], "beside a bad problem")


def test_grid_path_planning_default_solver(oracle):
    """The reference's CLI-default sweep (src/grid_path_plan.py:278-283): 32 poses, N = 800."""
    got = grid_path_planning(0, 0, 0, 40, 20, 4, verbose=False)
    want = grid_path_planning(0, 0, 0, 40, 20, 4, verbose=False, solver=oracle.solve_packed)
    assert len(got) == len(want) == 32                      # the reference's published outcome
    for g, w in zip(got, want):
        assert set(g) == {"start", "results", "ego"} and g["start"] == w["start"]
        _same_result(g["results"], w["results"], f"grid start {g['start']}")
        assert g["ego"].state_start == w["ego"].state_start and g["ego"].verbose is False


def test_grid_slice_of_function_defaults(oracle):
    """Every 55th pose of the 3520-pose sweep of the function defaults (21 x 21 x 8, N = 800)."""
    starts, total = grid_start_states(0.0, 0.0, 0.0)
    assert len(starts) == 3520
    sim = SimConfig()
    sim.duration = 80
    egos = []
    for (x, y, th) in starts[::55]:
        e = EgoConfig()
        e.state_final = [0.0, 0.0, 0.0, 0, 0]
        e.state_start = [x, y, th, 0, 0]
        e.verbose = False
        egos.append(e)
    assert len(egos) == 64
    packed = pack_problems(egos, sim)
    gpu = _lib.solve_packed(packed)
    ref = oracle.solve_packed(packed)
    cases.assert_same_solution(gpu, ref, "slice of the 3520-pose grid")
    assert (gpu["status"] == 0).sum() >= 32


def test_sequential_planning_default_solver(oracle):
    got, ok_g = sc.run_sequential_planning(sc.sequential_segments)
    want, ok_w = sc.run_sequential_planning(sc.sequential_segments, solver=oracle.solve_packed)
    assert ok_g and ok_w
    assert got["x"].shape == (1191, 5) and got["u"].shape == (1186, 2) and got["t"].shape == (1191,)
    for k in ("t", "x", "u"):
        assert np.array_equal(got[k], want[k]), k
    assert got["status"] == want["status"] == 0
    for a, b in zip(got["segments"], want["segments"]):
        assert (a["segment"], a["name"]) == (b["segment"], b["name"])
        _same_result(a["results"], b["results"], a["name"])


def test_scenarios_main_files_default_solver(oracle, tmp_path):
    """Both drivers of the reference's __main__ (src/scenarios.py:289-383): the files written with
    the HIP solver are byte for byte those written with the oracle as solver."""
    a = sc.main(docs_dir=sc.create_docs_dir(str(tmp_path / "hip")), run_individual_scenarios=True)
    b = sc.main(docs_dir=sc.create_docs_dir(str(tmp_path / "cpu")), run_individual_scenarios=True,
                solver=oracle.solve_packed)
    assert set(a) == set(b) == {"all_results", "sequential_path_results", "sequential_all_results"}
    for k in a:
        assert open(a[k], "rb").read() == open(b[k], "rb").read(), k


def _config3_cases():
    """Corner turn / U-turn / lane change (reference src/scenarios.py:206-229, commented out there)
    WITH obstacle constraints.  The reference never gives these scenarios obstacles; as the
    benchmark configuration words it, the pair of src/path_finder.py:37-41 is transplanted
    (synthetic: far from these paths, so the rows are present and inactive), and a second,
    equally synthetic pair is placed on each path so that the rows bind."""
    pair = [[-7, 12, 4.0, 1.0], [7, -12, 4.0, 1.0]]
    # (u_turn: the first obstacle touches the obstacle-free arc and the solve converges on its
    # row; corner_turn and line_change: a QP fails)
    placed = {"u_turn": [[5.0, 8.2, 1.0, 0.5], [5.0, -2.5, 0.8, 0.3]],
              "corner_turn": [[1.2, 6.1, 0.5, 0.3], [5.0, 8.2, 0.5, 0.3]],
              "line_change": [[13.8, -0.5, 1.0, 0.5], [24.0, -6.3, 1.0, 0.5]]}
    out = []
    for name in ("u_turn", "corner_turn", "line_change"):
        for tag, obs in (("transplanted", pair), ("placed", placed[name])):
            ego, sim = cases.make_scenario(name)
            ego.obstacles = [list(o) for o in obs]
            out.append((f"{name}/{tag}", ego, sim))
    return out


def test_config3_scenarios_with_obstacles(oracle):
    trio = _config3_cases()
    egos, sims = [t[1] for t in trio], [t[2] for t in trio]
    st_g, si_g, qi_g, res_g = solve_groups(egos, sims)
    st_o, si_o, qi_o, res_o = solve_groups(egos, sims, solver=oracle.solve_packed)
    names = [t[0] for t in trio]
    assert (st_g == st_o).all() and (si_g == si_o).all() and (qi_g == qi_o).all(), dict(zip(names, zip(st_g, st_o)))
    for n, a, b in zip(names, res_g, res_o):
        assert np.array_equal(a["x"], b["x"], equal_nan=True) and np.array_equal(a["u"], b["u"], equal_nan=True), n
    # far-away obstacles leave the obstacle-free outcome: all three converge
    for n, s in zip(names, st_g):
        if n.endswith("transplanted"):
            assert s == 0, (n, s)
    # where a placed pair leaves a converged path, it clears the obstacles
    for (n, ego, sim), s, r in zip(trio, st_g, res_g):
        if s == 0:
            for ox, oy, rad, margin in ego.obstacles:
                d = np.hypot(r["x"][1:-1, 0] - ox, r["x"][1:-1, 1] - oy)
                assert d.min() >= rad + margin - 1e-3, n
    assert dict(zip(names, st_g.tolist()))["u_turn/placed"] == 0
    assert {0, 4} <= set(st_g.tolist())          # converged and failed outcomes both occur (and agree, above)


def test_guess_computed_by_the_library(oracle):
    """xguess NULL: the solve computes the guess of core_solver.py:477-499 itself, node per lane;
    same results as with the host's array, bit for bit; with obstacles the library's guess follows
    the reference's rules (batch.library_guess is its restatement)."""
    import ctypes as C
    packed = synthetic_batch(3000)
    bare = type(packed)(packed.B, packed.N, packed.pdata, packed.n_obs, None, packed.duration)
    a = _lib.solve_packed(packed)
    b = _lib.solve_packed(bare)
    cases.assert_same_solution(b, a, "library guess vs host guess")
    for lanes in ("1", "64"):
        import os
        os.environ["MPCX_LANES"] = lanes
        try:
            cases.assert_same_solution(_lib.solve_packed(bare.slice(0, 200)), oracle.solve_packed(packed.slice(0, 200)),
                                       f"library guess, {lanes} per wavefront")
        finally:
            del os.environ["MPCX_LANES"]
    # other horizons and widths
    egos, sims = zip(*(cases.make_scenario(n) for n in ("line", "A_to_B", "C_to_D", "u_turn")))
    st_g, _, _, res_g = solve_groups(list(egos), list(sims))                      # default: guess left to the library
    st_o, _, _, res_o = solve_groups(list(egos), list(sims), solver=oracle.solve_packed)
    assert (st_g == st_o).all()
    for a_, b_ in zip(res_g, res_o):
        assert np.array_equal(a_["x"], b_["x"]) and np.array_equal(a_["u"], b_["u"])
    # batches with obstacles and no guess: the library moves its guess off the obstacles by the
    # reference's rules; the numpy restatement of that (PackedBatch.guess) is the same array bit for bit,
    # and the solve from it matches the oracle's
    obs = synthetic_batch(2000, obstacle=True, device_guess="all")
    assert obs.xguess is None and int(obs.n_obs.min()) == 1
    dev_guess = _lib.initial_guess_on_device(obs.pdata, None, obs.N, n_obs=obs.n_obs)
    assert np.array_equal(dev_guess, obs.guess())
    assert np.abs(dev_guess - synthetic_batch(2000, obstacle=True).xguess).max() < 1e-12
    cases.assert_same_solution(_lib.solve_packed(obs), oracle.solve_packed(obs), "library guess, one obstacle each")
    pairs = [cases.make_scenario("path_finder")] + [(e, s) for _, e, s in _config3_cases()[:4]]
    for ego, sim in pairs:
        p = pack_problems([ego], sim, device_guess="all")
        assert p.xguess is None
        assert np.array_equal(_lib.initial_guess_on_device(p.pdata, None, p.N, n_obs=p.n_obs), p.guess())
        cases.assert_same_solution(_lib.solve_packed(p), oracle.solve_packed(p), "library guess, scenario with obstacles")


def test_context_reuses_its_buffers(oracle):
    """mpcx_ctx: the handle's results equal those of the one-shot entry point, on a batch, then on a
    smaller one with another horizon (buffers kept), then on a larger one (buffers grown)."""
    with _lib.Context(0) as ctx:
        for packed in (synthetic_batch(5000), pack_problems(*cases.obstacle_problems(96, 6.5)), synthetic_batch(20000),
                       synthetic_batch(0)):
            got = ctx.solve(packed)
            want = _lib.solve_packed(packed)
            cases.assert_same_solution(got, want, f"context, B={packed.B} N={packed.N}")
            st = ctx.stats
            assert [int(v) for v in st[:5]] == [int((want["status"] == k).sum()) for k in range(5)]
            assert int(st[5]) == int(want["sqp_iter"].sum()) and int(st[6]) == int(want["qp_iter"].sum())
        bare = synthetic_batch(777)
        bare = type(bare)(bare.B, bare.N, bare.pdata, bare.n_obs, None, bare.duration)
        cases.assert_same_solution(ctx.solve(bare), oracle.solve_packed(bare), "context, library guess")


def test_rti_failed_vehicle_keeps_its_plan(oracle):
    """A vehicle whose tick fails (QP failure) keeps its previous plan shifted on, is reported in
    ``failed`` and can be reset; the caller's options object is not modified."""
    from mpc_amd.rti import RtiServer
    packed = synthetic_batch(2048)
    full = _lib.solve_packed(packed)
    ok = np.nonzero(full["status"] == 0)[0][:64]
    fleet = packed.take(ok)
    X, U = full["X"][ok].copy(), full["U"][ok].copy()
    fleet.pdata[5:7, 3] += 500.0            # vehicle 3: a goal no QP can reach
    fleet.pdata[36:38, 3] += 500.0
    fleet.pdata[41:43, 3] += 500.0
    opts = _lib.default_opts()
    srv = RtiServer(fleet, X=X, U=U, opts=opts)
    assert opts.rti == 0 and srv.opts.rti == 1
    inp = srv.host_inputs()
    srv.solve()
    torch.cuda.synchronize()
    status = srv.status.cpu().numpy()
    oopts = _lib.default_opts()
    oopts.rti = 1
    ref = oracle.solve_packed(inp, opts=oopts)
    assert np.array_equal(status, ref["status"]) and status[3] != 0 and (np.delete(status, 3) == 0).all()
    u0 = srv.advance().cpu().numpy()
    assert srv.failed.cpu().numpy().tolist() == [i == 3 for i in range(64)]
    assert np.array_equal(u0[3], inp.uguess[0, :, 3])                    # the previous plan's input
    assert np.array_equal(srv.xguess[1, :, 3].cpu().numpy(), inp.xguess[2, :, 3])
    good = np.delete(np.arange(64), 3)
    assert np.array_equal(u0[good], ref["U"][good, 0, :])
    assert srv.reset() == 1 and not srv.failed.any()
    g = srv.xguess[:, :, 3].cpu().numpy()
    assert np.array_equal(g[0, :3], srv.pdata[0:3, 3].cpu().numpy()) and g[0, 3] == 1.0 and (srv.uguess[:, :, 3] == 0).all()


def test_start_inside_constraints_matches_oracle(oracle):
    """Start poses that the obstacle or corridor rows exclude: the device ends them exactly as the oracle
    does (status 4 after one QP, same iteration counts), and find_path returns None."""
    pairs = cases.infeasible_start_problems()
    egos, sims = [p[0] for p in pairs], [p[1] for p in pairs]
    got = solve_groups(egos, sims)
    want = solve_groups(egos, sims, solver=oracle.solve_packed)
    for a, b in zip(got[:3], want[:3]):
        assert np.array_equal(a, b)
    assert list(got[0]) == [4, 4, 4]
    for ego, sim in pairs:
        assert find_path(ego, sim) is None


def _bench_line(*flags):
    import json, os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), *flags], capture_output=True, text=True,
                       timeout=600, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    return json.loads(lines[0])


def test_bench_line_carries_both_stories():
    """One run, one JSON line.  N = 1: the headline plus the workload whose goals are reachable by
    construction.  N = 2 (rehearsed on this one GPU, collectives over gloo): the weak number plus the
    north-star's strong-scaling block -- one batch, contiguous slices, gather to rank 0 in the timed region."""
    small = ("--batch", "4096", "--steps", "1", "--warmup", "0", "--extra-steps", "1")
    one = _bench_line(*small, "--cpu-sample", "512")
    assert one["n_gpus"] == 1 and one["scaling"] == "weak" and one["dtype"] == "f64" and one["unit"] == "OCP/s"
    assert {"roofline", "cpu_baseline", "host_to_host", "reachable", "two_launches_in_flight"} <= set(one)
    assert one["two_launches_in_flight"]["same_converged_count"] is True
    assert one["cpu_baseline"]["gpu_matches_cpu_on_sample"] is True
    rb = one["reachable"]
    assert rb["converged_fraction"] > 0.85 > one["config"]["converged_fraction"] and rb["value"] > 0 and rb["frac"] > 0
    two = _bench_line(*small, "--gpus", "2", "--rehearse-on-one-gpu")
    assert two["n_gpus"] == 2 and two["scaling"] == "weak" and "rehearsal" in two and "reachable" not in two
    assert two["config"]["problems_total"] == 2 * 4096
    sb = two["strong"]
    assert [p["problems"] for p in sb["per_rank"]] == [2048, 2048] and sb["gather_bytes"] > 0
    assert sb["value"] > 0 and sb["ms_per_step"] > 0 and sb["gather_ms_per_step"] >= 0
    assert abs(sb["converged_fraction"] - one["config"]["converged_fraction"]) < 1e-12      # the same 4096 problems
