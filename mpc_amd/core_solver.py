"""Drop-in front end for reference src/core_solver.py.

Same names and semantics as the reference module — ``EgoConfig`` (:11-105),
``SimConfig`` (:107-110), ``generate_controls`` (:112-524), ``find_path``
(:527-562) — with the acados solve (:354-513) replaced by the batched HIP
library, plus the batched forms ``generate_controls_batch`` / ``find_paths``
that the grid and scenario loops use.  Nothing is written to the working
directory (the reference drops acados_ocp_kinematic_car.json and
c_generated_code/ there, :354).
"""
import os

import numpy as np

from . import _lib
from .batch import horizon_length, pack_problems

_STATE_ERR = "State must have 5 elements: [x, y, theta, velocity, steering]"


class EgoConfig:
    """Vehicle, bounds, weights, obstacles, start and goal of one problem.

    Attribute names and defaults are the reference's (core_solver.py:14-56).
    ``state_start`` / ``state_final`` are properties whose setters copy the
    given list and wrap the heading with ``((a + pi) % 2pi) - pi`` (:59-105),
    so +pi is stored as -pi.
    """

    _DEFAULTS = (
        ("L", 2.7),
        ("velocity_max", 3.0), ("velocity_min", -3.0),
        ("acceleration_max", 2.0), ("acceleration_min", -2.0),
        ("steering_max", 0.5), ("steering_min", -0.5),
        ("steering_rate_max", 0.4), ("steering_rate_min", -0.4),
        ("corridor_width", 15.0),
        ("obstacles", None),
        ("weight_acceleration", 100.0), ("weight_steering_rate", 100.0),
        ("weight_position_x", 0.0), ("weight_position_y", 0.0),
        ("weight_heading", 0.0), ("weight_velocity", 0.0), ("weight_steering", 0.0),
        ("weight_terminal_position_x", 100.0), ("weight_terminal_position_y", 100.0),
        ("weight_terminal_heading", 100.0), ("weight_terminal_velocity", 10.0),
        ("weight_terminal_steering", 10.0),
        ("verbose", True),
    )

    def __init__(self):
        for name, value in self._DEFAULTS:
            setattr(self, name, value)
        # constructor defaults bypass the setters, as in the reference (:30-31)
        self._state_start = [-40, 0, 0, 0, 0]
        self._state_final = [+40, 0, 0, 0, 0]

    @staticmethod
    def normalize_angle(angle):
        return ((angle + np.pi) % (2 * np.pi)) - np.pi

    def _wrapped(self, state):
        if len(state) != 5:
            raise ValueError(_STATE_ERR)
        out = state.copy()
        out[2] = self.normalize_angle(state[2])
        return out

    @property
    def state_start(self):
        return self._state_start

    @state_start.setter
    def state_start(self, state):
        self._state_start = self._wrapped(state)

    @property
    def state_final(self):
        return self._state_final

    @state_final.setter
    def state_final(self, state):
        self._state_final = self._wrapped(state)


class SimConfig:
    def __init__(self):
        self.duration = 30.0
        self.dt = 0.1


class _Silence:
    """fd-level stdout/stderr silencer (role of reference utils/io_utils.py:6-37)."""

    def __enter__(self):
        self._null = [os.open(os.devnull, os.O_RDWR) for _ in range(2)]
        self._saved = [os.dup(1), os.dup(2)]
        os.dup2(self._null[0], 1)
        os.dup2(self._null[1], 2)
        return self

    def __exit__(self, *exc):
        os.dup2(self._saved[0], 1)
        os.dup2(self._saved[1], 2)
        for fd in self._null + self._saved:
            os.close(fd)
        return False


def _result_dict(out, b, N, duration):
    return {
        "t": np.linspace(0, duration, N + 1),
        "x": out["X"][b].copy(),
        "u": out["U"][b].copy(),
        "status": int(out["status"][b]),
    }


def solve_groups(egos, sim_cfgs, opts=None, device=0, solver=None, errors=None):
    """Solve many problems; returns ``(status, sqp_iter, qp_iter, results)``.

    Problems are grouped by horizon length N (one library call per distinct N,
    since a call has one N), solved, and scattered back to their input order.
    ``results[i]`` is the reference's result dict whatever the status.
    ``solver`` replaces the HIP call with a callable of the same signature
    as ``_lib.solve_packed`` (tests use it to run the CPU oracle).
    ``errors``: None = a problem that cannot be set up raises (as one call of the
    reference's generate_controls would); a dict = such a problem is left out,
    ``errors[i]`` receives its exception, ``results[i]`` stays None and
    ``status[i]`` is -1, and every other problem is solved.
    """
    device_guess = solver is None       # the HIP library computes the guess of obstacle-free batches itself
    solver = solver or _lib.solve_packed
    n = len(egos)
    if not isinstance(sim_cfgs, (list, tuple)):
        sim_cfgs = [sim_cfgs] * n
    groups = {}
    for i, sc in enumerate(sim_cfgs):
        try:
            key = (horizon_length(sc), float(sc.duration))
            if key[0] < 1:
                raise ValueError("duration / dt must give at least one shooting interval")
        except Exception as e:
            if errors is None:
                raise
            errors[i] = e
            continue
        groups.setdefault(key, []).append(i)
    status = np.zeros(n, dtype=np.int32)
    sqp_iter = np.zeros(n, dtype=np.int32)
    qp_iter = np.zeros(n, dtype=np.int32)
    results = [None] * n
    for (N, duration), idx in groups.items():
        bad = None if errors is None else {}
        packed = pack_problems([egos[i] for i in idx], sim_cfgs[idx[0]], errors=bad, device_guess=device_guess)
        if bad:
            for j, e in bad.items():
                errors[idx[j]] = e
            idx = [i for j, i in enumerate(idx) if j not in bad]
        out = solver(packed, opts, device)
        for j, i in enumerate(idx):
            status[i] = out["status"][j]
            sqp_iter[i] = out["sqp_iter"][j]
            qp_iter[i] = out["qp_iter"][j]
            results[i] = _result_dict(out, j, N, duration)
    if errors:
        for i in errors:
            status[i] = -1
    return status, sqp_iter, qp_iter, results


def generate_controls_batch(egos, sim_cfgs, opts=None, device=0, solver=None, errors=None):
    """Batched ``generate_controls``: list of result dicts, ``None`` where the
    solver status is non-zero (with the reference's printed line, :505-507).
    ``errors`` as in ``solve_groups`` (no line is printed for a problem listed there)."""
    status, _, _, results = solve_groups(egos, sim_cfgs, opts, device, solver, errors)
    out = []
    for i, (s, r) in enumerate(zip(status, results)):
        if errors and i in errors:
            out.append(None)
        elif s != 0:
            print(f"Solver failed with status {int(s)}")
            out.append(None)
        else:
            out.append(r)
    return out


def generate_controls(ego, sim_cfg, opts=None, device=0, solver=None):
    """Plan one trajectory: ``{'t','x','u','status'}`` or ``None`` on failure."""
    return generate_controls_batch([ego], [sim_cfg], opts, device, solver)[0]


def find_paths(egos, sim_cfgs, opts=None, device=0, solver=None):
    """Batched ``find_path``: never raises; ``None`` for every failed problem.

    A problem that cannot be set up fails alone, as in the reference's loop of
    ``find_path`` calls (core_solver.py:540-562, grid_path_plan.py:78-89): the
    others are still solved.  Only a failure of the solve call itself (no
    device, library missing) fails the whole batch.  Output of the solve is
    silenced unless some ``ego.verbose`` is true; the per-problem success /
    failure lines follow each problem's own flag.
    """
    n = len(egos)
    errors = {}
    try:
        if any(e.verbose for e in egos):
            res = generate_controls_batch(egos, sim_cfgs, opts, device, solver, errors)
        else:
            with _Silence():
                res = generate_controls_batch(egos, sim_cfgs, opts, device, solver, errors)
    except Exception as e:                              # reference :559-562
        for ego in egos:
            if ego.verbose:
                print(f"✗ Error during path planning: {str(e)}")
        return [None] * n
    for i, (ego, r) in enumerate(zip(egos, res)):
        if ego.verbose:
            if i in errors:
                print(f"✗ Error during path planning: {str(errors[i])}")
            else:
                print("✓ Path planning successful" if r is not None
                      else "✗ Error during path planning: 'NoneType' object is not subscriptable")
    return res


def find_path(ego, sim_cfg, opts=None, device=0, solver=None):
    """``generate_controls`` that never raises and honours ``ego.verbose``."""
    return find_paths([ego], [sim_cfg], opts, device, solver)[0]
