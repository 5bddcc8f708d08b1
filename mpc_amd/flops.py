"""Algorithmic FP64 operation counts of the solve, per problem.

Two sets of per-stage constants (FMA = 2, + - * / = 1, sin and cos = 1 each):

* structure-exploiting (the set the roofline fraction is quoted on; ESTIMATED: rounded closed forms, not
  counted on any executable -- the counted figure is the dense one below): the closed forms for
  nx = 5, nu = 2 -- interior-point iteration 1.5e3 = Riccati factorisation
  7/3 nx^3 + 4 nx^2 nu + 2 nx nu^2 + nu^3/3 = 534, two substitutions (predictor, corrector)
  2 (8 nx^2 + 8 nx nu + 2 nu^2) = 576, barrier Hessian / right-hand sides / step lengths /
  residuals over the 18 row sides of a stage ~ 400; linearisation 1.1e3 = four RK4 stages with
  sparse forward sensitivities, cost gradient and row values; NLP residuals and cold start 2e2
  per QP solve;
* dense count: what the CPU restatement executes, frozen by running it built with an
  operation-counting scalar type on the synthetic benchmark batch (it multiplies by the
  structural zeros and ones of the sensitivities and always substitutes three right-hand
  sides); an upper bound of the algorithm's work, reported beside the other.
"""
import numpy as np

F_LIN, F_INIT, F_IPM = 1100.0, 200.0, 1500.0                       # structure-exploiting closed forms
F_LIN_DENSE, F_INIT_DENSE, F_IPM_DENSE = 1032.0, 345.0, 3444.0    # counted on the dense restatement


def constants(sparse=True):
    """The per-stage constants as a dict (for the benchmark line)."""
    if sparse:
        return {"per_stage_linearisation": F_LIN, "per_stage_qp_solve": F_INIT, "per_stage_ip_iteration": F_IPM,
                "kind": "closed forms for nx=5, nu=2 exploiting the sparsity of the sensitivities",
                "estimated": True}
    return {"per_stage_linearisation": F_LIN_DENSE, "per_stage_qp_solve": F_INIT_DENSE,
            "per_stage_ip_iteration": F_IPM_DENSE, "kind": "counted on the dense CPU restatement", "estimated": False}


def algorithmic_flops(status, sqp_iter, qp_iter, N, sparse=False):
    """Total flops over the problems of a batch, from their own counters."""
    status = np.asarray(status)
    sqp = np.asarray(sqp_iter, dtype=np.float64)
    qp = np.asarray(qp_iter, dtype=np.float64)
    # a QP failure (status 4) stops right after the QP; every other exit follows a linearisation
    n_lin = np.where(status == 4, sqp, sqp + 1.0)
    lin, init, ipm = (F_LIN, F_INIT, F_IPM) if sparse else (F_LIN_DENSE, F_INIT_DENSE, F_IPM_DENSE)
    return float(N * (lin * n_lin.sum() + init * sqp.sum() + ipm * qp.sum()))


def flops_from_stats(stats, N, sparse=False):
    """The same total from the device-side counters of mpcx_batch_stats_device
    (stats[5] = QP solves, [6] = interior-point iterations, [7] = linearisations)."""
    lin, init, ipm = (F_LIN, F_INIT, F_IPM) if sparse else (F_LIN_DENSE, F_INIT_DENSE, F_IPM_DENSE)
    return float(N * (lin * float(stats[7]) + init * float(stats[5]) + ipm * float(stats[6])))


def result_bytes(N):
    """Bytes of one problem's results: X, U, status and two counters, cost, four residuals."""
    return 8 * ((N + 1) * 5 + N * 2) + 3 * 4 + 8 * 5


def algorithmic_io_bytes(N, host_guess=False):
    """Bytes the caller hands over and gets back per problem (N shooting intervals): one column of
    the problem table (71 doubles) and the obstacle count, the state guess if the host supplies
    it, and the results."""
    return 8 * 71 + 4 + (8 * (N + 1) * 5 if host_guess else 0) + result_bytes(N)
