"""Test-side loader of the CPU oracle (oracle/libmpc_oracle.so).

Only tests, __graft_entry__.smoke() and bench.py's cpu_baseline leg use this.
"""
import ctypes as C
import os

import numpy as np

from mpc_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_SO = os.path.join(ROOT, "oracle", "libmpc_oracle.so")
ORACLE_COUNT_SO = os.path.join(ROOT, "oracle", "libmpc_oracle_count.so")
ORACLE_FAST_SO = os.path.join(ROOT, "oracle", "libmpc_oracle_fast.so")      # -O3 build: the timed CPU baseline

_cache = {}


def load(path=ORACLE_SO):
    if path in _cache:
        return _cache[path]
    if not os.path.exists(path):
        raise RuntimeError(f"{path} missing: run `make -C oracle`")
    lib = C.CDLL(path)
    lib.mpcx_oracle_opts_default.argtypes = [C.POINTER(_lib.MpcxOpts)]
    lib.mpcx_oracle_opts_default.restype = None
    lib.mpcx_oracle_solve_batch.argtypes = [C.POINTER(_lib.MpcxBatch), C.POINTER(_lib.MpcxOpts),
                                            C.POINTER(_lib.MpcxResult), C.c_int, C.c_int, C.c_int]
    lib.mpcx_oracle_solve_batch.restype = C.c_int
    lib.mpcx_oracle_sincos.argtypes = [C.c_double, C.POINTER(C.c_double), C.POINTER(C.c_double)]
    lib.mpcx_oracle_sincos.restype = None
    lib.mpcx_oracle_rk4.argtypes = [C.c_double, C.c_double] + [C.POINTER(C.c_double)] * 5
    lib.mpcx_oracle_rk4.restype = None
    lib.mpcx_oracle_qp_stride.argtypes = []
    lib.mpcx_oracle_qp_stride.restype = C.c_int
    lib.mpcx_oracle_first_qp.argtypes = [C.POINTER(_lib.MpcxBatch), C.POINTER(_lib.MpcxOpts), C.c_int,
                                         C.POINTER(C.c_double)]
    lib.mpcx_oracle_first_qp.restype = C.c_int
    _cache[path] = lib
    return lib


def default_opts():
    o = _lib.MpcxOpts()
    load().mpcx_oracle_opts_default(C.byref(o))
    return o


def solve_packed(packed, opts=None, device=0, nthreads=None, lib=None):
    """Same signature and output dict as mpc_amd._lib.solve_packed, on the CPU oracle."""
    lib = lib or load()
    opts = opts or default_opts()
    out = _lib.alloc_result(packed.B, packed.N)
    if packed.B == 0:
        return out
    if packed.xguess is None:      # the oracle has no guess of its own: the host loop's (reference :477-499)
        import dataclasses
        packed = dataclasses.replace(packed, xguess=packed.guess())
    b = _lib.batch_struct(packed)
    r = _lib.result_struct(out)
    if nthreads is None:
        nthreads = min(8, os.cpu_count() or 1)
    rc = lib.mpcx_oracle_solve_batch(C.byref(b), C.byref(opts), C.byref(r), 0, packed.B, nthreads)
    if rc != 0:
        raise RuntimeError(f"oracle failed: {rc}")
    return out


def first_qp(packed, b, opts=None):
    """Constraint set of problem b's first QP (linearisation at the initial guess) as the oracle poses it:
    array [N+1, stride], layout in oracle/mpc_oracle.c (mpcx_oracle_first_qp)."""
    lib = load()
    opts = opts or default_opts()
    if packed.xguess is None:
        import dataclasses
        packed = dataclasses.replace(packed, xguess=packed.guess())
    bs = _lib.batch_struct(packed)
    q = np.zeros((packed.N + 1, lib.mpcx_oracle_qp_stride()))
    rc = lib.mpcx_oracle_first_qp(C.byref(bs), C.byref(opts), int(b), q.ctypes.data_as(C.POINTER(C.c_double)))
    if rc != 0:
        raise RuntimeError(f"mpcx_oracle_first_qp failed: {rc}")
    return q
