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


def synthetic_batch(B, seed=SEED, duration=5.0, dt=0.1, obstacle=False):
    """PackedBatch of B random problems (N = int(duration/dt)), without obstacles
    or with one obstacle each."""
    if obstacle:
        start, goal, obs = random_start_goal(B, seed, with_obstacle=True)
        return pack_arrays(start, goal, duration, dt, obstacle=obs)
    start, goal = random_start_goal(B, seed)
    return pack_arrays(start, goal, duration, dt)
