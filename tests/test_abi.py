"""The C-ABI library loads and exports every symbol include/mpcx.h declares;
argument checking works without a GPU."""
import ctypes as C
import os
import re

import numpy as np

from mpc_amd import _lib
from mpc_amd.synthetic import synthetic_batch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_functions():
    text = open(os.path.join(ROOT, "include", "mpcx.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(mpcx_[a-z_0-9]+)\s*\(", text)))


def test_exports_match_header():
    lib = _lib.load_library()
    names = declared_functions()
    assert {"mpcx_version", "mpcx_last_error", "mpcx_opts_default", "mpcx_device_count",
            "mpcx_solve_batch", "mpcx_workspace_bytes", "mpcx_solve_batch_device",
            "mpcx_probe_sincos", "mpcx_ctx_create", "mpcx_ctx_solve", "mpcx_ctx_destroy",
            "mpcx_max_horizon"} <= set(names)
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/mpcx.h but not exported"
    assert lib.mpcx_version() == 108


def test_struct_layout_and_defaults():
    o = _lib.default_opts()
    assert C.sizeof(_lib.MpcxOpts) == 176 and C.sizeof(_lib.MpcxBatch) == 48 and C.sizeof(_lib.MpcxResult) == 56
    assert (o.max_sqp_iter, o.max_qp_iter, o.tol_stat, o.lm) == (200, 50, 1e-3, 1e-2)   # core_solver.py:246-252
    assert (o.qp_tol_stat, o.qp_tol_eq, o.qp_tol_ineq, o.qp_tol_comp, o.lm_scale_by_h) == (1e-6, 1e-8, 1e-8, 1e-6, 1)
    assert (o.mu0, o.thr0, o.cond_pred_corr, o.rti, o.max_obs) == (10.0, 0.1, 1, 0, -1)


def test_oracle_and_library_share_defaults(oracle):
    a, b = _lib.default_opts(), oracle.default_opts()
    assert bytes(a) == bytes(b)


def test_bad_arguments_and_no_device():
    lib = _lib.load_library()
    p = synthetic_batch(8)
    out = _lib.alloc_result(p.B, p.N)
    o = _lib.default_opts()
    b, r = _lib.batch_struct(p), _lib.result_struct(out)
    assert lib.mpcx_solve_batch(None, C.byref(o), C.byref(r), 0) == -1
    b.N = 0
    assert lib.mpcx_solve_batch(C.byref(b), C.byref(o), C.byref(r), 0) == -1
    assert b"N >= 1" in lib.mpcx_last_error()
    b.N = p.N
    o.max_obs = 9
    assert lib.mpcx_solve_batch(C.byref(b), C.byref(o), C.byref(r), 0) == -1
    o.max_obs = -1
    assert lib.mpcx_workspace_bytes(0, 50) == 0 and lib.mpcx_workspace_bytes(64, 50) > 0
    # horizons whose wavefront workspace would pass the 4 GiB that 32-bit offsets reach are refused
    nmax = lib.mpcx_max_horizon()
    assert 30_000 < nmax < 60_000
    b.N = nmax + 1
    assert lib.mpcx_solve_batch(C.byref(b), C.byref(o), C.byref(r), 0) == -1
    assert b"too large" in lib.mpcx_last_error()
    b.N = p.N
    # limits of the entry point that allocates nothing, computed by the library from its record layout
    ns = lib.mpcx_static_max_horizon()
    assert 500 < ns < 2000 and lib.mpcx_static_max_wavefronts(ns) >= 1 and lib.mpcx_static_max_wavefronts(ns + 1) == 0
    assert 8 <= lib.mpcx_static_max_wavefronts(50) <= 64 and lib.mpcx_static_max_wavefronts(0) == 0
    h = C.c_void_p()
    assert lib.mpcx_ctx_create(None, 0) == -1
    lib.mpcx_ctx_destroy(None)
    if lib.mpcx_device_count() == 0:
        assert lib.mpcx_solve_batch(C.byref(b), C.byref(o), C.byref(r), 0) == -2      # MPCX_E_NODEVICE
        x = np.zeros(4)
        assert lib.mpcx_probe_sincos(x.ctypes.data, 4, x.ctypes.data, x.ctypes.data, 0) == -2


def test_workspace_sizes(monkeypatch):
    """mpcx_workspace_bytes: at least one wavefront's share, grows with the spread of the batch,
    stays under the target (default 64 GiB, MPCX_WORKSPACE_GIB overrides), no device needed."""
    lib = _lib.load_library()
    one = lib.mpcx_workspace_bytes(1, 50)
    assert 1_000_000 < one < 20_000_000
    assert lib.mpcx_workspace_bytes(0, 50) == 0 and lib.mpcx_workspace_bytes(10, 0) == 0
    sizes = [lib.mpcx_workspace_bytes(B, 50) for B in (1, 2, 100, 1024, 5000, 10**6, 10**8)]
    assert sizes[:4] == sorted(sizes[:4]) and max(sizes) == sizes[3] == sizes[-1]    # 1024 wavefronts at most
    assert sizes[-1] < 1024 * one + 4096
    big = lib.mpcx_workspace_bytes(3520, 800)
    assert big <= (64 << 30) + 4096
    monkeypatch.setenv("MPCX_WORKSPACE_GIB", "4")
    assert lib.mpcx_workspace_bytes(3520, 800) <= (4 << 30) + 4096
    assert lib.mpcx_workspace_bytes(1, 50) == one
