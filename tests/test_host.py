"""Host-side logic: configuration objects, corridor, initial guess, packing,
grouping and ordering.  No GPU."""
import json
import os

import numpy as np
import pytest

from mpc_amd import EgoConfig, SimConfig, corridor_cal, verify_point_in_corridor
from mpc_amd import batch as bt
from mpc_amd.core_solver import find_path, find_paths, generate_controls, solve_groups
from mpc_amd.grid_path_plan import grid_start_states
from mpc_amd.synthetic import random_start_goal, synthetic_batch

import cases

HERE = os.path.dirname(os.path.abspath(__file__))
FIX = json.load(open(os.path.join(HERE, "golden", "corridor_guess.json")))


def test_ego_defaults_and_setters():
    e = EgoConfig()
    assert e.L == 2.7 and e.corridor_width == 15.0 and e.obstacles is None and e.verbose is True
    assert (e.velocity_min, e.velocity_max, e.acceleration_min, e.acceleration_max) == (-3.0, 3.0, -2.0, 2.0)
    assert (e.steering_min, e.steering_max, e.steering_rate_min, e.steering_rate_max) == (-0.5, 0.5, -0.4, 0.4)
    assert e.state_start == [-40, 0, 0, 0, 0] and e.state_final == [40, 0, 0, 0, 0]
    assert (e.weight_acceleration, e.weight_steering_rate) == (100.0, 100.0)
    assert (e.weight_terminal_position_x, e.weight_terminal_heading, e.weight_terminal_velocity) == (100.0, 100.0, 10.0)
    src = [1.0, 2.0, np.pi, 0.5, 0.0]
    e.state_start = src
    assert e.state_start is not src and src[2] == np.pi          # copied, input untouched
    assert e.state_start[2] == -np.pi                            # +pi wraps to -pi
    e.state_final = [0, 0, 3 * np.pi / 2, 0, 0]
    assert abs(e.state_final[2] + np.pi / 2) < 1e-15
    with pytest.raises(ValueError):
        e.state_start = [0, 0, 0]
    s = SimConfig()
    assert (s.duration, s.dt) == (30.0, 0.1)


def test_horizon_truncation():
    s = SimConfig()
    for duration, n in ((30.0, 300), (40, 400), (80, 800), (15.0, 150), (22.0, 220), (5.0, 50), (0.3, 2)):
        s.duration = duration
        assert bt.horizon_length(s) == n


def test_corridor_fixtures():
    for c in FIX["corridor"]:
        got = corridor_cal(tuple(c["a"]), tuple(c["b"]), c["distance"])
        assert [float(v) for v in got] == c["coeffs"]
    # hand-checked: horizontal line y = 5, half-width 2
    c1, a1, b1, c2, a2, b2 = corridor_cal((0, 5), (10, 5), 2.0)
    assert (c1, b1, b2) == (1.0, -3.0, -7.0) and a1 == 0
    assert verify_point_in_corridor((5, 5), c1, a1, b1, c2, a2, b2)
    assert not verify_point_in_corridor((5, 8), c1, a1, b1, c2, a2, b2)
    with pytest.raises(ValueError):
        corridor_cal((1, 1), (1, 1), 2.0)


def test_initial_guess_fixtures():
    ego, sim = cases.make_scenario("path_finder")
    g = bt.initial_guess(ego, 400)
    for i, row in FIX["guess"]["path_finder"]["rows"].items():
        assert g[int(i)].tolist() == row
    assert g[400].tolist() == [float(v) for v in ego.state_final]     # last node is the goal itself
    assert g[0, 3] == 1.0                                             # speed guess ignores the start speed
    g = bt.initial_guess(EgoConfig(), 300)
    for i, row in FIX["guess"]["default"]["rows"].items():
        assert g[int(i)].tolist() == row
    # hand check, obstacle branch: the path line is shifted sideways by R - dist + 0.1
    r = 4.0 + 1.0 + 0.05
    dist = np.hypot(-9.5 + 7, 9.5 - 12)
    shift = r - dist + 0.1
    g0 = bt.initial_guess(ego, 400)[0]
    assert abs(np.hypot(g0[0] + 20, g0[1] - 20) - np.hypot(shift, 0.6066017177982133)) < 1e-9


def test_vectorised_packing_equals_scalar():
    start, goal = random_start_goal(300, seed=5)
    start[:, 2] += np.linspace(-7, 7, 300)        # exercise the heading wrap
    packed_v = bt.pack_arrays(start, goal, 5.0)
    egos = []
    for s, g in zip(start, goal):
        e = EgoConfig()
        e.state_start = [float(v) for v in s]
        e.state_final = [float(v) for v in g]
        egos.append(e)
    sim = SimConfig()
    sim.duration = 5.0
    packed_s = bt.pack_problems(egos, sim)
    assert packed_v.N == packed_s.N == 50
    assert np.array_equal(packed_v.pdata, packed_s.pdata)
    assert np.array_equal(packed_v.xguess, packed_s.xguess)
    assert np.array_equal(packed_v.n_obs, packed_s.n_obs)


def test_vectorised_obstacle_guess_equals_scalar():
    """One obstacle per problem: the vectorised guess against the scalar loop through EgoConfig.
    The vectorised routine squares by multiplication where the scalar code calls pow(), so the
    bound is 1e-12 rather than equality; table fields must be equal."""
    B = 200
    start, goal, obs = random_start_goal(B, seed=5, with_obstacle=True)
    # both branches of the adjustment occur: line shifted (offset < 1.55) and not
    off = np.hypot(obs[:, 0] - (start[:, 0] + goal[:, 0]) / 2, obs[:, 1] - (start[:, 1] + goal[:, 1]) / 2)
    assert (off < 1.55).any() and (off > 1.55).any() and (off >= 0.5 - 1e-9).all() and (off <= 2 + 1e-9).all()
    egos = []
    for b in range(B):
        e = EgoConfig()
        e.state_start = list(start[b])
        e.state_final = list(goal[b])
        e.obstacles = [list(obs[b])]
        egos.append(e)
    sim = SimConfig()
    sim.duration = 5.0
    ref = bt.pack_problems(egos, sim)
    vec = synthetic_batch(B, seed=5, obstacle=True)
    assert np.array_equal(ref.pdata, vec.pdata) and np.array_equal(ref.n_obs, vec.n_obs)
    assert np.abs(ref.xguess - vec.xguess).max() <= 1e-12
    # the starts and goals are those of the obstacle-free batch of the same seed
    free = synthetic_batch(B, seed=5)
    assert np.array_equal(free.pdata[bt.F_X0:bt.F_X0 + 5], vec.pdata[bt.F_X0:bt.F_X0 + 5])


def test_pack_fields():
    ego, sim = cases.make_scenario("path_finder")
    p = bt.pack_problems([ego], sim)
    col = p.pdata[:, 0]
    assert p.N == 400 and p.n_obs[0] == 2 and col[bt.F_H] == 0.1
    assert col[bt.F_LBX + 3] == 0.0 and col[bt.F_UBX + 2] == np.pi and col[bt.F_LBX] == -100.0
    assert col[bt.F_LG] == -1e9 and col[bt.F_UH] == 1e9 and col[bt.F_LH] == 0.0
    assert col[bt.F_OBS + 2] == 25.0 and col[bt.F_OBS + 3] == 7.0
    np.testing.assert_allclose(col[bt.F_UBXE:bt.F_UBXE + 5] - col[bt.F_LBXE:bt.F_LBXE + 5],
                               [0.2, 0.2, 0.02, 0.02, 0.02], atol=1e-12)
    # corridor rows: -a1 x - c1 y <= b1 ; a2 x + c2 y <= -b2, half-width = corridor_width
    c1, a1, b1, c2, a2, b2 = corridor_cal((-20, 20), (20, -20), 10.0)
    assert col[bt.F_CG:bt.F_CG + 4].tolist() == [-a1, -c1, a2, c2] and col[bt.F_UG] == b1 and col[bt.F_UG + 1] == -b2
    e = EgoConfig()
    e.obstacles = [[0, 0, 1, 0]] * 5
    with pytest.raises(ValueError):
        bt.pack_problems([e], sim)


def test_grid_order_and_skip():
    starts, total = grid_start_states(0.0, 0.0, 0.0, 40, 20, 4)
    assert total == 36 and len(starts) == 32
    assert starts[0] == (-20.0, -20.0, 0.0) and starts[1][2] == np.pi / 2
    assert all(not (x == 0 and y == 0) for x, y, _ in starts)
    starts, total = grid_start_states(0.0, 0.0, 0.0)            # function defaults
    assert total == 21 * 21 * 8 and len(starts) == 3520
    # exactly 1 m away is NOT skipped (strict <)
    assert (1.0, 0.0, 0.0) in starts


def test_grouping_and_order_with_mixed_horizons(oracle):
    names = ["E_to_F", "line", "C_to_D", "lanechg" if False else "line_change"]
    pairs = [cases.make_scenario(n) for n in names]
    egos, sims = [p[0] for p in pairs], [p[1] for p in pairs]
    st, si, qi, res = solve_groups(egos, sims, solver=oracle.solve_packed)
    assert [r["x"].shape[0] for r in res] == [151, 151, 221, 221]
    for (ego, sim), r in zip(pairs, res):
        single = generate_controls(ego, sim, solver=oracle.solve_packed)
        assert np.array_equal(single["x"], r["x"]) and np.array_equal(single["u"], r["u"])
        assert single["t"][-1] == sim.duration and single["status"] == 0


def test_failure_contract(oracle, capsys):
    e = EgoConfig()
    e.state_start = [0, 0, 0, 0, 0]
    e.state_final = [90, 90, 0, 0, 0]          # unreachable in 5 s
    e.verbose = False
    s = SimConfig()
    s.duration = 5.0
    assert find_path(e, s, solver=oracle.solve_packed) is None
    assert generate_controls(e, s, solver=oracle.solve_packed) is None
    assert "Solver failed with status" in capsys.readouterr().out
    e2 = EgoConfig()
    e2.state_start = [0, 0, 0, 0, 0]
    e2.state_final = [0, 0, 0, 0, 0]           # start == goal: corridor_cal raises, find_path swallows it
    e2.verbose = False
    assert find_paths([e2], s, solver=oracle.solve_packed) == [None]


def test_find_paths_isolates_bad_problems(oracle, capsys):
    """One problem that cannot be set up fails alone (reference: one find_path call per problem,
    core_solver.py:540-562; the grid loop keeps going, grid_path_plan.py:78-89)."""
    good, s = cases.make_scenario("line")
    same = EgoConfig()                       # start == goal: corridor_cal raises
    same.state_start = [1, 1, 0, 0, 0]
    same.state_final = [1, 1, 0, 0, 0]
    many = EgoConfig()                       # more obstacles than a problem can carry
    many.state_start = list(good.state_start)
    many.state_final = list(good.state_final)
    many.obstacles = [[5.0, 9.0, 0.5, 0.1]] * 5
    short = EgoConfig()                      # a state of the wrong length smuggled past the setter
    short._state_start = [0, 0, 0]
    for e in (same, many, short):
        e.verbose = False
    res = find_paths([good, same, good, many, short, good], s, solver=oracle.solve_packed)
    assert [r is not None for r in res] == [True, False, True, False, False, True]
    alone = find_path(good, s, solver=oracle.solve_packed)
    for i in (0, 2, 5):
        assert np.array_equal(res[i]["x"], alone["x"]) and np.array_equal(res[i]["u"], alone["u"])
    # per-problem horizons: a bad SimConfig fails its own problem only
    bad_sim = SimConfig()
    bad_sim.dt = 0.0
    res = find_paths([good, good], [s, bad_sim], solver=oracle.solve_packed)
    assert res[0] is not None and res[1] is None
    # verbose problems report their own error
    same.verbose = True
    capsys.readouterr()
    find_paths([same], s, solver=oracle.solve_packed)
    assert "Points A and B cannot be the same" in capsys.readouterr().out
    # ... and a problem dropped for the obstacle limit (which the reference does not have) says so
    many.verbose = True
    find_paths([many], s, solver=oracle.solve_packed)
    said = capsys.readouterr().out
    assert "at most 4 obstacles" in said and "the reference has none" in said
    # generate_controls_batch without an error dict still raises, like one reference call would
    from mpc_amd.core_solver import generate_controls_batch
    with pytest.raises(ValueError):
        generate_controls_batch([good, same], s, solver=oracle.solve_packed)


def test_guess_left_to_the_library_is_the_host_loop():
    """A batch packed without a guess (xguess None: the library computes it on the device) yields,
    through PackedBatch.guess(), exactly the host loop's array; with obstacles the guess is packed."""
    start, goal = random_start_goal(100, seed=3)
    a = bt.pack_arrays(start, goal, 5.0)
    b = bt.pack_arrays(start, goal, 5.0, device_guess=True)
    assert b.xguess is None and np.array_equal(a.pdata, b.pdata) and np.array_equal(b.guess(), a.xguess)
    assert np.array_equal(a.pdata[bt.F_CW], np.full(100, 15.0))
    assert b.slice(10, 20).xguess is None and b.slice(10, 20).B == 10
    ego, sim = cases.make_scenario("path_finder")
    p = bt.pack_problems([ego], sim, device_guess=True)
    assert p.xguess is not None and p.pdata[bt.F_CW, 0] == 10.0
    q = bt.pack_problems([ego], sim, device_guess="all")
    assert q.xguess is None and np.array_equal(q.pdata, p.pdata)
    ego, sim = cases.make_scenario("line")
    p = bt.pack_problems([ego], sim, device_guess=True)
    assert p.xguess is None and np.array_equal(p.guess()[:, :, 0], bt.initial_guess(ego, p.N))


def test_priority_list_helpers():
    """longest_first ranks by a previous solve's iteration counts (ties and the short problems in the
    caller's order); a slice of a batch keeps the list restricted to its problems, renumbered."""
    import dataclasses
    q = np.array([5, 400, 7, 9000, 3, 400, 3, 3, 500, 2])
    assert bt.longest_first(q).tolist() == [3, 8, 1, 5, 2, 0, 4, 6, 7, 9]
    assert bt.longest_first(q, threshold=300).tolist() == [3, 8, 1, 5, 0, 2, 4, 6, 7, 9]
    assert bt.longest_first(np.zeros(4)).tolist() == [0, 1, 2, 3] and bt.longest_first(np.zeros(0)).shape == (0,)
    start, goal = random_start_goal(10, seed=3)
    a = dataclasses.replace(bt.pack_arrays(start, goal, 5.0), order=bt.longest_first(q))
    assert a.slice(2, 7).order.tolist() == [1, 3, 0, 2, 4] and a.slice(2, 7).order.dtype == np.int32   # problems 3, 5, 2, 4, 6
    assert a.take(np.array([9, 3, 8])).order.tolist() == [1, 2, 0]
    assert bt.pack_arrays(start, goal, 5.0).slice(2, 7).order is None


def test_library_guess_with_obstacles_follows_the_reference_rules():
    """What the library computes when a batch with obstacles brings no guess (restated in numpy,
    batch.library_guess): the reference's rules -- line shifted past a straddling obstacle, points
    inside an obstacle pushed out radially, several obstacles, sticky side -- to within rounding of the
    scalar routine that uses the reference's own operations (pow and arctan2 / cos / sin)."""
    from mpc_amd.core_solver import EgoConfig, SimConfig
    rng = np.random.default_rng(11)
    egos = []
    sim = SimConfig()
    sim.duration = 6.0
    for k in range(60):
        e = EgoConfig()
        e.state_start = [rng.uniform(-5, 5), rng.uniform(-5, 5), rng.uniform(-1, 1), 0.5, 0]
        ang = rng.uniform(-np.pi, np.pi)
        d = rng.uniform(6, 15)
        e.state_final = [e.state_start[0] + d * np.cos(ang), e.state_start[1] + d * np.sin(ang), ang, 0, 0]
        n = int(rng.integers(1, 5))
        obs = []
        for _ in range(n):
            s_ = rng.uniform(0.1, 0.9)
            off = rng.uniform(-3.0, 3.0)
            obs.append([e.state_start[0] + s_ * d * np.cos(ang) - off * np.sin(ang),
                        e.state_start[1] + s_ * d * np.sin(ang) + off * np.cos(ang),
                        rng.uniform(0.3, 1.5), rng.uniform(0.1, 0.6)])
        e.obstacles = obs
        e.verbose = False
        egos.append(e)
    host = bt.pack_problems(egos, sim)                          # the scalar loop, the reference's operations
    lib_ = bt.pack_problems(egos, sim, device_guess="all")
    g = lib_.guess()
    assert g.shape == host.xguess.shape
    assert np.abs(g - host.xguess).max() < 1e-12
    moved = np.abs(g[:, :2] - bt.library_guess(lib_.pdata, None, lib_.N)[:, :2]).max(axis=(0, 1)) > 1e-9
    assert moved.sum() > 20                                     # the obstacle branch did something in many problems
    # one obstacle each, vectorised host path: same shift branch bit for bit, push branch to rounding
    start, goal, obs = random_start_goal(500, seed=5, with_obstacle=True)
    a = bt.pack_arrays(start, goal, 5.0, obstacle=obs)
    b = bt.pack_arrays(start, goal, 5.0, obstacle=obs, device_guess="all")
    assert b.xguess is None and np.abs(b.guess() - a.xguess).max() < 1e-12


def test_no_silent_cpu_fallback():
    """Without a HIP device the product path must raise, not compute."""
    from mpc_amd import _lib
    lib = _lib.load_library()
    if lib.mpcx_device_count() > 0:
        pytest.skip("a HIP device is present")
    with pytest.raises(_lib.MpcxError):
        _lib.solve_packed(synthetic_batch(4))
    e, s = cases.make_scenario("line")
    with pytest.raises(_lib.MpcxError):
        generate_controls(e, s)
    assert find_path(e, s) is None             # find_path never raises (reference :559-562)


def test_results_carry_what_the_reference_plots_read(oracle):
    """Consumer side: everything reference plots.py reads from a result dict and its EgoConfig
    (:20-22 t/x/u, :31-32 start/goal, :45-56 obstacles, :72 corridor_cal on the two points and the
    width, :130-164 the bound attributes, :212/:248-249 grid entries) is there with the type and
    shape those lines index."""
    e, s = cases.make_scenario("line")
    e.obstacles = None
    r = find_path(e, s, solver=oracle.solve_packed)
    N = int(s.duration / s.dt)
    assert set(r) >= {"t", "x", "u", "status"}
    assert r["t"].shape == (N + 1,) and r["x"].shape == (N + 1, 5) and r["u"].shape == (N, 2)
    assert r["x"].dtype == np.float64 and r["x"][:, 0:5].shape == (N + 1, 5) and r["u"][:, 1].shape == (N,)
    assert len(e.state_start) == 5 and len(e.state_final) == 5
    assert hasattr(e, "obstacles") and e.obstacles is None
    assert len(corridor_cal((e.state_start[0], e.state_start[1]), (e.state_final[0], e.state_final[1]),
                            e.corridor_width)) == 6
    for name in ("velocity", "steering", "acceleration", "steering_rate"):
        lo, hi = getattr(e, name + "_min"), getattr(e, name + "_max")
        assert float(lo) < float(hi)
    np.rad2deg(e.state_final[2]), np.rad2deg(e.steering_max)
    # an obstacle is (x, y, radius, margin), as the circles are drawn (:46-54)
    eo, so = cases.make_scenario("line")
    eo.obstacles = [[5.0, 3.0, 1.0, 0.5]]
    assert all(len(o) == 4 for o in eo.obstacles)
    # grid entries (:212, :248-249)
    from mpc_amd.grid_path_plan import grid_path_planning
    got = grid_path_planning(0.0, 0.0, 0.0, grid_size=40, grid_step=40, num_angles=1,
                             solver=oracle.solve_packed, verbose=False)
    assert len(got) >= 1
    for g in got:
        assert set(g) == {"start", "results", "ego"} and len(g["start"]) == 3
        assert g["results"]["x"].shape[1] == 5 and g["ego"].state_final[:2] == [0.0, 0.0]


def test_result_files_load_like_the_reference(oracle, tmp_path):
    """Files written by scenarios.main load the way reference create_animation.py:10-14 loads
    them (own files, so unpickling is safe) and embed EgoConfig under module name core_solver."""
    import pickletools
    from mpc_amd import scenarios as sc
    written = sc.main(docs_dir=str(tmp_path), solver=oracle.solve_packed)
    assert set(written) == {"sequential_path_results", "sequential_all_results"}
    raw = open(written["sequential_path_results"], "rb").read()
    assert b"core_solver" in raw and b"mpc_amd" not in raw
    with sc._PickledAs("core_solver"):
        combined = np.load(written["sequential_path_results"], allow_pickle=True).item()
    assert combined["x"].shape == (1191, 5) and combined["u"].shape == (1186, 2)
    assert combined["t"].shape == (1191,) and combined["status"] == 0
    assert [s["name"] for s in combined["segments"]] == [s["name"] for s in sc.sequential_segments]
    assert combined["segments"][0]["ego"].corridor_width == sc.sequential_segments[0].get("corridor_width", 2.0)


def test_rti_plan_ok_rejects_non_finite_iterates():
    """A real-time iteration reports status 0 with the residuals from before its one step; the server
    therefore checks the new iterate itself and keeps a vehicle's previous plan when any number of it
    is not finite (poisoned guess, overflowing step)."""
    import torch
    from mpc_amd.rti import plan_ok, QP_TOL_PRESETS
    B, N = 6, 4
    X = np.zeros((B, N + 1, 5)); U = np.zeros((B, N, 2))
    status = np.array([0, 0, 4, 0, 0, 1], dtype=np.int32)
    X[1, 2, 3] = np.nan
    U[3, 0, 1] = np.inf
    X[4, N, 0] = -np.inf
    want = np.array([True, False, False, False, False, False])
    assert np.array_equal(plan_ok(np, status, X, U), want)
    got = plan_ok(torch, torch.from_numpy(status), torch.from_numpy(X), torch.from_numpy(U))
    assert np.array_equal(got.numpy(), want)
    assert QP_TOL_PRESETS["reference"] == (1e-6, 1e-8, 1e-8, 1e-6) and set(QP_TOL_PRESETS["loose"]) == {1e-3}
