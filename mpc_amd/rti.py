"""Receding-horizon real-time iteration (RTI) over a fleet of vehicles.

The reference solves every problem from scratch (src/core_solver.py:354-502);
it has no receding-horizon loop.  This module defines one on top of the same
C ABI: every vehicle keeps its last solution on the device, each tick performs
exactly ONE linearisation and ONE QP solve per vehicle (``mpcx_opts.rti = 1``)
from the previous solution shifted by one stage, applies the first input to a
simulated plant (one RK4 step of the model of core_solver.py:148-155) and makes
the resulting state the next tick's initial state.

Tick semantics, for vehicle b with horizon N and step h:
  1. solve:  (X, U) <- one SQP step from (xguess, uguess) with x0 = pdata[X0]
  2. apply:  u0 = U[0];  x+ = RK4_h(x0, u0)
  3. shift:  xguess[k] = X[k+1] (k < N), xguess[N] = X[N];  xguess[0] = x+
             uguess[k] = U[k+1] (k < N-1), uguess[N-1] = U[N-1]
             pdata[X0] = x+
A vehicle whose tick did not end with status 0 (QP failure, non-finite residual), or whose new
iterate holds a non-finite number (the status of a real-time iteration reports the residuals from
BEFORE its one step, so the iterate is checked here: ``plan_ok``), keeps its PREVIOUS plan: in steps
2 and 3 its (X, U) are the tick's input guess instead of the failed iterate, so the plan it already
had is carried on one more stage; ``failed`` holds the mask of such vehicles after ``advance`` and
``reset`` restarts chosen vehicles from the straight-line guess of core_solver.py:477-499 at their
current state.

QP tolerances.  The library's defaults (1e-6 / 1e-8 / 1e-8 / 1e-6) are the ones that reproduce the
reference's recorded trajectories; a tick then takes about 7 interior-point iterations per vehicle.
A server that prefers shorter ticks to the last digits can pass ``qp_tol="loose"`` (all four at the
NLP tolerance 1e-3, about 4.5 iterations per tick): see ``QP_TOL_PRESETS``.
The goal, bounds and corridor stay as packed.  PyTorch is used for device
memory and for the elementwise shift / plant step only; the solve is the HIP
library's.  There is no CPU fallback.

Import this module (or torch) before the first call into libmpcx.so in a process:
torch brings its own HIP runtime, and the runtime loaded first is the one both use.
"""
import ctypes as C

import numpy as np
import torch

from . import _lib
from .batch import F_X0, F_L, F_H


# (stationarity, equalities, inequalities, complementarity) of the QP solver, by name
QP_TOL_PRESETS = {
    "reference": (1e-6, 1e-8, 1e-8, 1e-6),      # the library's defaults
    "loose": (1e-3, 1e-3, 1e-3, 1e-3),          # QP stopped at the NLP tolerance of core_solver.py:252
}


def plan_ok(xp, status, X, U):
    """[B] mask of vehicles whose tick produced a usable plan: status 0 and every number of the new
    iterate finite.  xp = numpy or torch module."""
    finite = xp.isfinite(X).reshape(X.shape[0], -1).all(1) & xp.isfinite(U).reshape(U.shape[0], -1).all(1)
    return (status == 0) & finite


def plant_step(xp, x, u, h, wheelbase):
    """One RK4 step of the kinematic bicycle; x [B,5], u [B,2]; xp = numpy or torch module."""
    def f(s):
        return xp.stack([s[:, 3] * xp.cos(s[:, 2]), s[:, 3] * xp.sin(s[:, 2]),
                         s[:, 3] * xp.tan(s[:, 4]) / wheelbase, u[:, 0], u[:, 1]], 1)
    k1 = f(x)
    k2 = f(x + 0.5 * h[:, None] * k1)
    k3 = f(x + 0.5 * h[:, None] * k2)
    k4 = f(x + h[:, None] * k3)
    return x + (h[:, None] / 6.0) * (k1 + 2.0 * k2 + 2.0 * k3 + k4)


class RtiServer:
    def __init__(self, packed, X=None, U=None, opts=None, device=0, qp_tol=None):
        """packed: PackedBatch (problem table + initial guess).  X [B,N+1,5], U [B,N,2]:
        a previous solution to start from (else the packed guess and zero inputs).
        qp_tol: None = as in ``opts``; a key of QP_TOL_PRESETS or four numbers."""
        self.torch = torch
        self.lib = _lib.load_library()
        if self.lib.mpcx_device_count() <= 0:
            raise _lib.MpcxError("no HIP device available (there is no CPU fallback)")
        torch.cuda.set_device(device)
        self.dev = torch.device("cuda", device)
        self.B, self.N = packed.B, packed.N
        B, N = self.B, self.N
        # a copy: the caller's options object is not switched to real-time iteration behind its back
        self.opts = _lib.MpcxOpts.from_buffer_copy(opts) if opts is not None else _lib.default_opts(self.lib)
        self.opts.rti = 1
        self.opts.max_obs = int(packed.n_obs.max()) if B else 0
        if qp_tol is not None:
            tol = QP_TOL_PRESETS[qp_tol] if isinstance(qp_tol, str) else tuple(float(v) for v in qp_tol)
            (self.opts.qp_tol_stat, self.opts.qp_tol_eq, self.opts.qp_tol_ineq, self.opts.qp_tol_comp) = tol
        self.failed = torch.zeros(B, dtype=torch.bool, device=self.dev)
        to = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(self.dev)
        self.pdata = to(packed.pdata)
        self.n_obs = to(packed.n_obs)
        self.xguess = to(packed.guess() if X is None else np.transpose(X, (1, 2, 0)))
        self.uguess = to(np.zeros((N, 2, B)) if U is None else np.transpose(U, (1, 2, 0)))
        f64 = dict(dtype=torch.float64, device=self.dev)
        i32 = dict(dtype=torch.int32, device=self.dev)
        self.X = torch.empty((B, N + 1, 5), **f64)
        self.U = torch.empty((B, N, 2), **f64)
        self.status = torch.empty(B, **i32)
        self.sqp_iter = torch.empty(B, **i32)
        self.qp_iter = torch.empty(B, **i32)
        self.cost = torch.empty(B, **f64)
        self.res = torch.empty((4, B), **f64)
        self.ws_bytes = self.lib.mpcx_workspace_bytes(B, N)
        self.ws = torch.empty(self.ws_bytes, dtype=torch.uint8, device=self.dev)
        self._bs = _lib.MpcxBatch(B, N, self.pdata.data_ptr(), self.n_obs.data_ptr(),
                                  self.xguess.data_ptr(), self.uguess.data_ptr())
        self._rs = _lib.MpcxResult(self.X.data_ptr(), self.U.data_ptr(), self.status.data_ptr(),
                                   self.sqp_iter.data_ptr(), self.qp_iter.data_ptr(),
                                   self.cost.data_ptr(), self.res.data_ptr())
        self.ticks = 0
        self._warm()

    def _warm(self):
        """Everything a tick does for the first time, done once here on copies: the first use of each
        elementwise kernel loads its code and the first temporaries come from the driver, not from
        torch's cache -- hundreds of milliseconds that would otherwise land in the first tick."""
        if self.B == 0:
            return
        t = self.torch
        self.status.zero_()
        self.X.copy_(self.xguess.permute(2, 0, 1))
        self.U.copy_(self.uguess.permute(2, 0, 1))
        saved = (self.xguess.clone(), self.uguess.clone(), self.pdata[F_X0:F_X0 + 5].clone(), self.ticks)
        self.advance()
        self.xguess.copy_(saved[0]); self.uguess.copy_(saved[1]); self.pdata[F_X0:F_X0 + 5] = saved[2]
        self.ticks = saved[3]
        self.failed.zero_()
        t.cuda.synchronize(self.dev)

    def host_inputs(self):
        """The inputs of the NEXT tick as a PackedBatch on the host (for checking)."""
        from .batch import PackedBatch
        return PackedBatch(self.B, self.N, self.pdata.cpu().numpy(), self.n_obs.cpu().numpy(),
                           self.xguess.cpu().numpy(), 0.0, self.uguess.cpu().numpy())

    def solve(self):
        """Step 1 of a tick, asynchronous on torch's current stream."""
        stream = self.torch.cuda.current_stream()
        rc = self.lib.mpcx_solve_batch_device(C.byref(self._bs), C.byref(self.opts), C.byref(self._rs),
                                              self.ws.data_ptr(), self.ws_bytes, stream.cuda_stream)
        if rc != 0:
            raise _lib.MpcxError(self.lib.mpcx_last_error().decode())

    def advance(self):
        """Steps 2 and 3 of a tick on the device; returns the applied inputs [B,2]."""
        t = self.torch
        ok = plan_ok(t, self.status, self.X, self.U)[:, None, None]
        self.failed = ~ok[:, 0, 0]
        # failed vehicles carry their previous plan (this tick's input guess) on
        X = t.where(ok, self.X, self.xguess.permute(2, 0, 1))
        U = t.where(ok, self.U, self.uguess.permute(2, 0, 1))
        u0 = U[:, 0, :].clone()
        x0 = self.pdata[F_X0:F_X0 + 5].t()
        xn = plant_step(t, x0, u0, self.pdata[F_H], self.pdata[F_L])
        self.xguess[:-1] = X[:, 1:, :].permute(1, 2, 0)
        self.xguess[-1] = X[:, -1, :].t()
        self.xguess[0] = xn.t()
        if self.N > 1:
            self.uguess[:-1] = U[:, 1:, :].permute(1, 2, 0)
        self.uguess[-1] = U[:, -1, :].t()
        self.pdata[F_X0:F_X0 + 5] = xn.t()
        self.ticks += 1
        return u0

    def reset(self, vehicles=None):
        """Restart vehicles (index tensor / array, default: the failed ones) from the straight-line
        guess at their current state and zero inputs.  The guess is the library's device routine
        (mpcx_initial_guess_device): the reference's loop bit for bit without obstacles, and with up to
        MPCX_MAX_OBS obstacles its rules with last-bit differences in pow and arctan2 (include/mpcx.h)."""
        t = self.torch
        if vehicles is None:
            vehicles = t.nonzero(self.failed)[:, 0]
        vehicles = t.as_tensor(vehicles, device=self.dev, dtype=t.long)
        if vehicles.numel() == 0:
            return 0
        fresh = t.empty_like(self.xguess)
        stream = t.cuda.current_stream()
        rc = self.lib.mpcx_initial_guess_device(self.pdata.data_ptr(), self.n_obs.data_ptr(), None, self.B, self.N,
                                                fresh.data_ptr(), stream.cuda_stream)
        if rc != 0:
            raise _lib.MpcxError(self.lib.mpcx_last_error().decode())
        self.xguess[:, :, vehicles] = fresh[:, :, vehicles]
        self.uguess[:, :, vehicles] = 0.0
        self.failed[vehicles] = False
        return int(vehicles.numel())

    def tick(self):
        self.solve()
        return self.advance()
