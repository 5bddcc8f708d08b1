"""Problem sets shared by the CPU and GPU tests.

The scenario tables restate reference src/scenarios.py:205-287 (the live
`line` scenario, the three commented-out ones, the five sequential segments)
and src/path_finder.py:24-41 (the obstacle demo).
"""
import numpy as np

from mpc_amd import EgoConfig, SimConfig

PI = np.pi

SCENARIOS = {
    # name: (start, goal, corridor_width, duration, velocity_min, obstacles)
    "line": ([0, 0, 0, 1.0, 0], [30, 0, 0, 2.0, 0], 3.0, 15.0, 0.0, None),
    "A_to_B": ([30, 35, PI, 2.0, 0], [0, 30, PI, 1.0, 0], 7.0, 22.0, 0.0, None),
    "B_to_C": ([0, 30, PI, 1, 0], [0, 20, 0, 1, 0], 7.0, 40.0, 0.0, None),
    "C_to_D": ([0, 20, 0, 1.0, 0], [30, 15, 0, 1.0, 0], 7.0, 22.0, 0.0, None),
    "D_to_E": ([30, 15, 0, 1.0, 0], [37, 8, -PI / 2, 1.0, 0], 3.0, 20.0, 0.0, None),
    "E_to_F": ([37, 8, -PI / 2, 1.0, 0], [37, -22, -PI / 2, 3.0, 0], 3.0, 15.0, 0.0, None),
    "u_turn": ([0, 0, PI / 2, 1.0, 0], [10, 0, -PI / 2, 1.0, 0], 7.0, 40.0, 0.0, None),
    "corner_turn": ([0, 0, PI / 2, 1.0, 0], [7, 7, 0, 1.0, 0], 3.0, 20.0, 0.0, None),
    "line_change": ([0, 0, 0, 2.0, 0], [30, -5, 0, 1.0, 0], 7.0, 22.0, 0.0, None),
    "path_finder": ([-20, 20, -PI / 4, 0.0, 0], [20, -20, -PI / 4, 0.0, 0], 10.0, 40, 0.0,
                    [[-7, 12, 4.0, 1.0], [7, -12, 4.0, 1.0]]),
}


def make_scenario(name):
    start, goal, width, duration, vmin, obstacles = SCENARIOS[name]
    ego = EgoConfig()
    sim = SimConfig()
    ego.corridor_width = width
    sim.duration = duration
    ego.state_start = list(start)
    ego.state_final = list(goal)
    ego.velocity_min = vmin
    ego.obstacles = obstacles
    ego.verbose = False
    return ego, sim


def grid_problems(goal=(0.0, 0.0, 0.0), grid_size=40, grid_step=20, num_angles=4, duration=80):
    """The start-pose sweep of reference src/grid_path_plan.py:28-76."""
    gx, gy, gth = goal
    half = grid_size / 2
    xs = np.arange(gx - half, gx + half + grid_step, grid_step)
    ys = np.arange(gy - half, gy + half + grid_step, grid_step)
    angles = np.linspace(0, 2 * np.pi, num_angles, endpoint=False)
    sim = SimConfig()
    sim.duration = duration
    egos = []
    for x in xs:
        for y in ys:
            for th in angles:
                if np.sqrt((x - gx)**2 + (y - gy)**2) < 1.0:
                    continue
                e = EgoConfig()
                e.state_final = [gx, gy, gth, 0, 0]
                e.state_start = [x, y, th, 0, 0]
                e.verbose = False
                egos.append(e)
    return egos, sim


def assert_same_solution(gpu, ref, what=""):
    """GPU output dict against the oracle's: integers exact, floats bit for bit
    (-0.0 == +0.0 and NaN == NaN allowed)."""
    for key in ("status", "sqp_iter", "qp_iter"):
        bad = np.nonzero(gpu[key] != ref[key])[0]
        assert bad.size == 0, f"{what}: {key} differs at {bad[:8]}: {gpu[key][bad[:8]]} vs {ref[key][bad[:8]]}"
    ok = ref["status"] != 1
    for key in ("X", "U", "cost"):
        a, b = gpu[key][ok], ref[key][ok]
        same = (a == b) | (np.isnan(a) & np.isnan(b))
        if not same.all():
            diff = np.nanmax(np.abs(a - b))
            raise AssertionError(f"{what}: {key} not bit-identical, {np.count_nonzero(~same)} "
                                 f"elements differ, max abs diff {diff:.3e}")
    a, b = gpu["res"][:, ok], ref["res"][:, ok]
    assert ((a == b) | (np.isnan(a) & np.isnan(b))).all(), f"{what}: residuals differ"


def obstacle_problems(B, duration, seed=5, max_obs=2):
    """Random start/goal pairs with one to `max_obs` obstacles near the straight path (a synthetic
    mix: the reference's only obstacle case is the path_finder demo above)."""
    rng = np.random.default_rng(seed)
    sim = SimConfig()
    sim.duration = duration
    egos = []
    for b in range(B):
        e = EgoConfig()
        s = np.array([rng.uniform(-20, 20), rng.uniform(-20, 20), rng.uniform(-1.5, 1.5), rng.uniform(0, 1), 0.0])
        d = rng.uniform(0.4, 0.9) * 2.0 * duration
        phi = s[2] + rng.uniform(-0.5, 0.5)
        g = np.array([s[0] + d * np.cos(phi), s[1] + d * np.sin(phi), phi + rng.uniform(-0.4, 0.4),
                      rng.uniform(0, 1), 0.0])
        e.state_start = list(s)
        e.state_final = list(g)
        obs = []
        for j in range(1 + b % max_obs):
            t = rng.uniform(0.3, 0.7)
            off = rng.choice([-1.0, 1.0]) * rng.uniform(0.5, 2.0)
            obs.append([s[0] + t * (g[0] - s[0]) - off * np.sin(phi), s[1] + t * (g[1] - s[1]) + off * np.cos(phi),
                        1.0, 0.5])
        e.obstacles = obs if b % 3 else None
        e.verbose = False
        egos.append(e)
    return egos, sim


def infeasible_start_problems():
    """Start poses the constraints exclude: inside an inflated obstacle, on its margin, and outside the
    corridor.  The rows are imposed from stage 1 on (x0 is fixed), so the first linearised QP is
    infeasible and the solve ends with status 4; the reference holds no fixture for such a start."""
    from mpc_amd.core_solver import EgoConfig, SimConfig
    out = []
    for start, obstacles in (([0, 0, 0, 1, 0], [[0.5, 0.3, 1.0, 0.5]]),
                             ([0, 0, 0, 1, 0], [[1.2, 0.0, 1.0, 0.5]]),
                             ([0, 9, 0, 1, 0], None)):
        e, s = EgoConfig(), SimConfig()
        s.duration = 10.0
        e.state_start = start
        e.state_final = [15, 0, 0, 0, 0]
        e.obstacles = obstacles
        e.verbose = False
        out.append((e, s))
    return out
