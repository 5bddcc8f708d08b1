"""mpc_amd — batched kinematic-car OCP solver for AMD MI355X.

Drop-in for the solve path of reference src/core_solver.py: the Python
surface (EgoConfig, SimConfig, generate_controls, find_path) is kept, the
acados solve behind it is replaced by hand-written HIP kernels reached
through the C ABI of include/mpcx.h.
"""
from .core_solver import (EgoConfig, SimConfig, find_path, find_paths,
                          generate_controls, generate_controls_batch, solve_groups)
from .corridor import corridor_cal, verify_point_in_corridor

__all__ = ["EgoConfig", "SimConfig", "find_path", "find_paths", "generate_controls",
           "generate_controls_batch", "solve_groups", "corridor_cal",
           "verify_point_in_corridor"]
