"""Synthetic random start/goal batch of the headline benchmark.

There is no counterpart in the reference (its loops are the grid sweep and
the scenario table); the distribution below is the one the benchmark is
defined on: seed 20250310, duration 5 s, dt 0.1 s (N = 50), every other
EgoConfig field at its default (reference core_solver.py:14-53).
"""
import numpy as np

from .batch import pack_arrays

SEED = 20250310


def random_start_goal(B, seed=SEED, with_obstacle=False):
    """start, goal as float64 [B, 5], drawn in a fixed order from one generator.

    with_obstacle: also return [B, 4] rows (x, y, radius 1, margin 0.5): one
    obstacle per problem at the middle of the start-goal segment, moved
    sideways by 0.5 .. 2 m to either side (drawn after everything else, so
    the starts and goals are those of the obstacle-free batch).
    """
    rng = np.random.default_rng(seed)
    start = np.zeros((B, 5))
    goal = np.zeros((B, 5))
    start[:, 0] = rng.uniform(-20.0, 20.0, B)
    start[:, 1] = rng.uniform(-20.0, 20.0, B)
    start[:, 2] = rng.uniform(-np.pi / 2, np.pi / 2, B)
    start[:, 3] = rng.uniform(0.0, 1.0, B)
    d = rng.uniform(3.0, 8.0, B)
    phi = start[:, 2] + rng.uniform(-np.pi / 4, np.pi / 4, B)
    goal[:, 0] = start[:, 0] + d * np.cos(phi)
    goal[:, 1] = start[:, 1] + d * np.sin(phi)
    goal[:, 2] = np.clip(phi + rng.uniform(-np.pi / 6, np.pi / 6, B),
                         -np.pi + 0.05, np.pi - 0.05)
    goal[:, 3] = rng.uniform(0.0, 1.0, B)
    if not with_obstacle:
        return start, goal
    offset = rng.uniform(0.5, 2.0, B) * np.where(rng.uniform(-1.0, 1.0, B) < 0.0, -1.0, 1.0)
    px = goal[:, 0] - start[:, 0]
    py = goal[:, 1] - start[:, 1]
    plen = np.sqrt(px * px + py * py)
    obstacle = np.empty((B, 4))
    obstacle[:, 0] = (start[:, 0] + goal[:, 0]) / 2 + offset * (-py / plen)
    obstacle[:, 1] = (start[:, 1] + goal[:, 1]) / 2 + offset * (px / plen)
    obstacle[:, 2] = 1.0
    obstacle[:, 3] = 0.5
    return start, goal, obstacle


def synthetic_batch(B, seed=SEED, duration=5.0, dt=0.1, obstacle=False, device_guess=False):
    """PackedBatch of B random problems (N = int(duration/dt)), without obstacles
    or with one obstacle each.  device_guess: as in pack_arrays (True: the guess of an
    obstacle-free batch is left to the library; "all": that of the obstacle batch too)."""
    if obstacle:
        start, goal, obs = random_start_goal(B, seed, with_obstacle=True)
        return pack_arrays(start, goal, duration, dt, obstacle=obs, device_guess=device_guess)
    start, goal = random_start_goal(B, seed)
    return pack_arrays(start, goal, duration, dt, device_guess=device_guess)


def reachable_start_goal(B, seed=SEED, duration=5.0, dt=0.1, L=2.7):
    """start, goal [B, 5] where every goal IS reachable: the start is drawn as in
    ``random_start_goal``; the goal is the state a random admissible input sequence leads to in
    ``duration`` seconds (five one-second pieces of constant acceleration in [-0.6, 0.6] m/s^2 and
    steering rate in [-0.15, 0.15] rad/s, cut back where speed would leave [0, 1.8] m/s or
    steering [-0.45, 0.45] rad; gentle enough that about 95 % of the problems also converge from
    the reference's straight-line guess -- a feasible problem whose first linearised QP is
    infeasible still fails, the solver takes full steps), integrated with one RK4 step of the model per interval
    (reference core_solver.py:148-155) -- the discretisation the solver itself uses, so the
    sequence is a feasible point of the problem up to the terminal box."""
    rng = np.random.default_rng(seed)
    N = int(duration / dt)
    h = duration / N
    start = np.zeros((B, 5))
    start[:, 0] = rng.uniform(-20.0, 20.0, B)
    start[:, 1] = rng.uniform(-20.0, 20.0, B)
    start[:, 2] = rng.uniform(-np.pi / 2, np.pi / 2, B)
    start[:, 3] = rng.uniform(0.0, 1.0, B)
    pieces = 5
    acc = rng.uniform(-0.6, 0.6, (pieces, B))
    rate = rng.uniform(-0.15, 0.15, (pieces, B))

    def f(s, u):
        return np.stack([s[:, 3] * np.cos(s[:, 2]), s[:, 3] * np.sin(s[:, 2]),
                         s[:, 3] * np.tan(s[:, 4]) / L, u[:, 0], u[:, 1]], 1)
    x = start.copy()
    for k in range(N):
        j = min(pieces - 1, (k * pieces) // N)
        a = np.clip(acc[j], (0.0 - x[:, 3]) / h, (1.8 - x[:, 3]) / h)
        r = np.clip(rate[j], (-0.45 - x[:, 4]) / h, (0.45 - x[:, 4]) / h)
        u = np.stack([a, r], 1)
        k1 = f(x, u)
        k2 = f(x + 0.5 * h * k1, u)
        k3 = f(x + 0.5 * h * k2, u)
        k4 = f(x + h * k3, u)
        x = x + (h / 6.0) * (k1 + 2.0 * k2 + 2.0 * k3 + k4)
    goal = x
    goal[:, 2] = np.clip(goal[:, 2], -np.pi + 0.05, np.pi - 0.05)
    return start, goal


def reachable_batch(B, seed=SEED, duration=5.0, dt=0.1, device_guess=False):
    """PackedBatch of B problems whose goals are reachable by construction (see
    ``reachable_start_goal``); every other field as in ``synthetic_batch``."""
    start, goal = reachable_start_goal(B, seed, duration, dt)
    return pack_arrays(start, goal, duration, dt, device_guess=device_guess)
