"""Start-pose grid sweep, batched (reference src/grid_path_plan.py:12-92).

The reference runs one ``find_path`` per grid point in a triple loop; here the
same points, in the same x -> y -> heading order, form ONE batch.  The plotting
and animation half of the reference file (:94-273) is presentation and is not
rebuilt; the returned list has the reference's structure so those functions
can consume it unchanged.
"""
import numpy as np

from .core_solver import EgoConfig, SimConfig, find_paths


def grid_start_states(goal_x, goal_y, goal_theta, grid_size=20, grid_step=1, num_angles=8):
    """The (x, y, theta) tuples the reference loop visits, skips applied, in its order."""
    half_size = grid_size / 2
    x_points = np.arange(goal_x - half_size, goal_x + half_size + grid_step, grid_step)
    y_points = np.arange(goal_y - half_size, goal_y + half_size + grid_step, grid_step)
    angles = np.linspace(0, 2 * np.pi, num_angles, endpoint=False)
    starts = []
    for x in x_points:
        for y in y_points:
            for theta in angles:
                # reference :67-70: strictly closer than 1 m is skipped
                if np.sqrt((x - goal_x)**2 + (y - goal_y)**2) < 1.0:
                    continue
                starts.append((x, y, theta))
    total = len(x_points) * len(y_points) * len(angles)
    return starts, total


def grid_path_planning(goal_x, goal_y, goal_theta, grid_size=20, grid_step=1, num_angles=8,
                       opts=None, device=0, solver=None, verbose=True):
    """Plan from every grid start to the goal; returns the successful results.

    Each entry is ``{'start': (x, y, theta), 'results': dict, 'ego': EgoConfig}``
    with the un-wrapped start heading, successes only, in loop order
    (reference :80-86).  ``duration`` is 80 s as in the reference (:53).
    """
    starts, total = grid_start_states(goal_x, goal_y, goal_theta, grid_size, grid_step, num_angles)
    if verbose:
        print(f"Grid search: {len(starts)} start poses ({total} combinations, "
              f"{total - len(starts)} skipped within 1 m of the goal)")
    sim_cfg = SimConfig()
    sim_cfg.duration = 80
    state_final = [goal_x, goal_y, goal_theta, 0, 0]
    egos = []
    for (x, y, theta) in starts:
        ego = EgoConfig()
        ego.state_final = state_final.copy()
        ego.state_start = [x, y, theta, 0, 0]
        ego.verbose = False
        egos.append(ego)
    results = find_paths(egos, sim_cfg, opts=opts, device=device, solver=solver)
    successful = [{"start": s, "results": r, "ego": e}
                  for s, r, e in zip(starts, results, egos) if r is not None]
    if verbose:
        print(f"Completed grid search. Successful paths: {len(successful)}/{total}")
    return successful
