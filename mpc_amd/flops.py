"""Algorithmic FP64 operation counts of the solve, per problem.

The constants were frozen by running the CPU restatement built with an
operation-counting scalar type (FMA = 2, + - * / = 1, sin and cos = 1 each)
on the synthetic benchmark batch (no obstacles, default options):
  F_LIN   per stage and linearisation: RK4 + sensitivities, cost and row
          values, NLP residuals, iterate update, objective
  F_INIT  per stage and QP solve: interior-point cold start + first residuals
  F_IPM   per stage and interior-point iteration (mean over the batch: the
          conditional third Riccati solve runs on part of the iterations)
A structure-exploiting kernel executes fewer operations than this; the
figure is the algorithm's, as the roofline convention asks.
"""
import numpy as np

F_LIN = 1032.0
F_INIT = 345.0
F_IPM = 3325.0


def algorithmic_flops(status, sqp_iter, qp_iter, N):
    """Total flops over the problems of a batch, from their own counters."""
    status = np.asarray(status)
    sqp = np.asarray(sqp_iter, dtype=np.float64)
    qp = np.asarray(qp_iter, dtype=np.float64)
    # a QP failure (status 4) stops right after the QP; every other exit follows a linearisation
    n_lin = np.where(status == 4, sqp, sqp + 1.0)
    return float(N * (F_LIN * n_lin.sum() + F_INIT * sqp.sum() + F_IPM * qp.sum()))


def flops_from_stats(stats, N):
    """The same total from the device-side counters of mpcx_batch_stats_device
    (stats[5] = QP solves, [6] = interior-point iterations, [7] = linearisations)."""
    return float(N * (F_LIN * float(stats[7]) + F_INIT * float(stats[5]) + F_IPM * float(stats[6])))


# bytes the caller hands over and gets back per problem (N shooting intervals)
def algorithmic_io_bytes(N):
    return 8 * (70 + (N + 1) * 5) + 4 + 8 * ((N + 1) * 5 + N * 2) + 3 * 4 + 8 * 5
