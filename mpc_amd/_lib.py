"""ctypes binding of libmpcx.so (include/mpcx.h).

The library is built in-tree (mpc_amd/csrc/libmpcx.so) by
``__graft_entry__.build()``.  There is no CPU fallback: if the library is
missing or no HIP device is usable, solving raises.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "csrc", "libmpcx.so")

c_double_p = C.POINTER(C.c_double)
c_int32_p = C.POINTER(C.c_int32)


class MpcxBatch(C.Structure):
    _fields_ = [("B", C.c_int32), ("N", C.c_int32),
                ("pdata", C.c_void_p), ("n_obs", C.c_void_p), ("xguess", C.c_void_p),
                ("uguess", C.c_void_p), ("order", C.c_void_p)]


class MpcxOpts(C.Structure):
    _fields_ = [("max_sqp_iter", C.c_int32), ("max_qp_iter", C.c_int32),
                ("tol_stat", C.c_double), ("tol_eq", C.c_double),
                ("tol_ineq", C.c_double), ("tol_comp", C.c_double),
                ("qp_tol_stat", C.c_double), ("qp_tol_eq", C.c_double),
                ("qp_tol_ineq", C.c_double), ("qp_tol_comp", C.c_double),
                ("lm", C.c_double), ("lm_scale_by_h", C.c_int32),
                ("cond_pred_corr", C.c_int32),
                ("mu0", C.c_double), ("thr0", C.c_double), ("alpha_min", C.c_double),
                ("lam_min", C.c_double), ("t_min", C.c_double),
                ("rti", C.c_int32), ("max_obs", C.c_int32),
                ("globalization", C.c_int32), ("reserved_", C.c_int32),
                ("ls_shrink", C.c_double), ("ls_alpha_min", C.c_double),
                ("ls_c1", C.c_double), ("ls_rho_factor", C.c_double)]


class MpcxResult(C.Structure):
    _fields_ = [("X", C.c_void_p), ("U", C.c_void_p), ("status", C.c_void_p),
                ("sqp_iter", C.c_void_p), ("qp_iter", C.c_void_p),
                ("cost", C.c_void_p), ("res", C.c_void_p)]


class MpcxError(RuntimeError):
    pass


MPCX_NFIELDS = 71      # rows of the problem table (include/mpcx.h)
MPCX_NSTATS = 8


_lib = None


def load_library(path=None):
    """dlopen libmpcx.so and declare its prototypes.  Raises if it is absent."""
    global _lib
    if _lib is not None and path is None:
        return _lib
    # (MPCX_LIB_PATH: another build of the library, for A/B runs of diagnostic builds)
    p = path or os.environ.get("MPCX_LIB_PATH") or LIB_PATH
    if not os.path.exists(p):
        raise MpcxError(
            f"{p} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(there is no CPU fallback)")
    lib = C.CDLL(p)
    lib.mpcx_version.restype = C.c_int
    lib.mpcx_last_error.restype = C.c_char_p
    lib.mpcx_opts_default.argtypes = [C.POINTER(MpcxOpts)]
    lib.mpcx_opts_default.restype = None
    lib.mpcx_device_count.restype = C.c_int
    lib.mpcx_solve_batch.argtypes = [C.POINTER(MpcxBatch), C.POINTER(MpcxOpts),
                                     C.POINTER(MpcxResult), C.c_int]
    lib.mpcx_solve_batch.restype = C.c_int
    lib.mpcx_workspace_bytes.argtypes = [C.c_int32, C.c_int32]
    lib.mpcx_workspace_bytes.restype = C.c_size_t
    lib.mpcx_solve_batch_device.argtypes = [C.POINTER(MpcxBatch), C.POINTER(MpcxOpts),
                                            C.POINTER(MpcxResult), C.c_void_p,
                                            C.c_size_t, C.c_void_p]
    lib.mpcx_solve_batch_device.restype = C.c_int
    lib.mpcx_initial_guess_device.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p]
    lib.mpcx_initial_guess_device.restype = C.c_int
    lib.mpcx_initial_guess.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_int]
    lib.mpcx_initial_guess.restype = C.c_int
    lib.mpcx_batch_stats_device.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p]
    lib.mpcx_batch_stats_device.restype = C.c_int
    lib.mpcx_probe_sincos.argtypes = [C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_int]
    lib.mpcx_probe_sincos.restype = C.c_int
    lib.mpcx_max_horizon.restype = C.c_int
    lib.mpcx_ctx_create.argtypes = [C.POINTER(C.c_void_p), C.c_int]
    lib.mpcx_ctx_create.restype = C.c_int
    lib.mpcx_ctx_solve.argtypes = [C.c_void_p, C.POINTER(MpcxBatch), C.POINTER(MpcxOpts),
                                   C.POINTER(MpcxResult), C.c_void_p]
    lib.mpcx_ctx_solve.restype = C.c_int
    lib.mpcx_ctx_destroy.argtypes = [C.c_void_p]
    lib.mpcx_ctx_destroy.restype = None
    if hasattr(lib, "mpcx_solve_batch_static"):
        lib.mpcx_solve_batch_static.argtypes = [C.POINTER(MpcxBatch), C.POINTER(MpcxOpts),
                                                C.POINTER(MpcxResult), C.c_void_p]
        lib.mpcx_solve_batch_static.restype = C.c_int
        lib.mpcx_static_max_horizon.restype = C.c_int
        lib.mpcx_static_max_wavefronts.argtypes = [C.c_int32]
        lib.mpcx_static_max_wavefronts.restype = C.c_int
    if path is None:
        _lib = lib
    return lib


def default_opts(lib=None):
    lib = lib or load_library()
    o = MpcxOpts()
    lib.mpcx_opts_default(C.byref(o))
    return o


def _check(lib, rc, what):
    if rc != 0:
        msg = lib.mpcx_last_error()
        raise MpcxError(f"{what} failed ({rc}): {msg.decode() if msg else ''}")


def alloc_result(B, N):
    """Host arrays of one call's outputs, in the layout of mpcx_result."""
    return {
        "X": np.empty((B, N + 1, 5), dtype=np.float64),
        "U": np.empty((B, N, 2), dtype=np.float64),
        "status": np.empty(B, dtype=np.int32),
        "sqp_iter": np.empty(B, dtype=np.int32),
        "qp_iter": np.empty(B, dtype=np.int32),
        "cost": np.empty(B, dtype=np.float64),
        "res": np.empty((4, B), dtype=np.float64),
    }


def batch_struct(packed):
    """mpcx_batch over the host arrays of a PackedBatch (xguess None = NULL: the library's guess;
    order None = NULL: the library's own fetch order)."""
    assert packed.pdata.flags.c_contiguous and packed.pdata.dtype == np.float64
    assert packed.pdata.shape == (MPCX_NFIELDS, packed.B)
    assert packed.n_obs.flags.c_contiguous and packed.n_obs.dtype == np.int32
    xg = packed.xguess
    if xg is not None:
        assert xg.flags.c_contiguous and xg.dtype == np.float64 and xg.shape == (packed.N + 1, 5, packed.B)
    ug = getattr(packed, "uguess", None)
    if ug is not None:
        assert ug.flags.c_contiguous and ug.dtype == np.float64 and ug.shape == (packed.N, 2, packed.B)
    order = getattr(packed, "order", None)
    if order is not None:
        assert order.flags.c_contiguous and order.dtype == np.int32 and order.shape == (packed.B,)
    return MpcxBatch(packed.B, packed.N, packed.pdata.ctypes.data,
                     packed.n_obs.ctypes.data, xg.ctypes.data if xg is not None else None,
                     ug.ctypes.data if ug is not None else None,
                     order.ctypes.data if order is not None else None)


def result_struct(out):
    return MpcxResult(*(out[k].ctypes.data for k in
                        ("X", "U", "status", "sqp_iter", "qp_iter", "cost", "res")))


def solve_packed(packed, opts=None, device=0):
    """Solve one PackedBatch on HIP device ``device``; returns the output dict."""
    lib = load_library()
    if lib.mpcx_device_count() <= 0:
        raise MpcxError("no HIP device available (there is no CPU fallback)")
    opts = opts or default_opts(lib)
    out = alloc_result(packed.B, packed.N)
    if packed.B == 0:
        return out
    b = batch_struct(packed)
    r = result_struct(out)
    _check(lib, lib.mpcx_solve_batch(C.byref(b), C.byref(opts), C.byref(r), int(device)),
           "mpcx_solve_batch")
    return out


class Context:
    """Solver handle (mpcx_ctx): device buffers, workspace and page-locked result arrays kept from
    call to call.  ``solve`` returns the output dict as numpy VIEWS of the handle's result arrays:
    valid until the next ``solve`` / ``close`` on this handle (copy what must outlive that)."""

    def __init__(self, device=0):
        self.lib = load_library()
        if self.lib.mpcx_device_count() <= 0:
            raise MpcxError("no HIP device available (there is no CPU fallback)")
        self._h = C.c_void_p()
        _check(self.lib, self.lib.mpcx_ctx_create(C.byref(self._h), int(device)), "mpcx_ctx_create")
        self.stats = np.zeros(MPCX_NSTATS, dtype=np.int64)

    def solve(self, packed, opts=None):
        opts = opts or default_opts(self.lib)
        b = batch_struct(packed)
        r = MpcxResult()
        _check(self.lib, self.lib.mpcx_ctx_solve(self._h, C.byref(b), C.byref(opts), C.byref(r),
                                                 self.stats.ctypes.data), "mpcx_ctx_solve")
        B, N = packed.B, packed.N

        def view(ptr, ctype, shape):
            n = int(np.prod(shape))
            if n == 0:
                return np.empty(shape, dtype=np.dtype(ctype))
            return np.ctypeslib.as_array(C.cast(ptr, C.POINTER(ctype)), shape=(n,)).reshape(shape)
        return {"X": view(r.X, C.c_double, (B, N + 1, 5)), "U": view(r.U, C.c_double, (B, N, 2)),
                "status": view(r.status, C.c_int32, (B,)), "sqp_iter": view(r.sqp_iter, C.c_int32, (B,)),
                "qp_iter": view(r.qp_iter, C.c_int32, (B,)), "cost": view(r.cost, C.c_double, (B,)),
                "res": view(r.res, C.c_double, (4, B))}

    def close(self):
        if self._h:
            self.lib.mpcx_ctx_destroy(self._h)
            self._h = C.c_void_p()

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()
        return False

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def initial_guess_on_device(pdata, width, N, device=0, n_obs=None):
    """xguess [N+1, 5, B] computed on HIP device ``device`` from the problem table (host arrays in,
    host array out); bit-identical to batch.library_guess, and without obstacles (n_obs None or
    zeros) to batch.initial_guess_batch.  width None = the table's own corridor-width column."""
    lib = load_library()
    if lib.mpcx_device_count() <= 0:
        raise MpcxError("no HIP device available (there is no CPU fallback)")
    pdata = np.ascontiguousarray(pdata, dtype=np.float64)
    if width is not None:
        width = np.ascontiguousarray(width, dtype=np.float64)
    if n_obs is not None:
        n_obs = np.ascontiguousarray(n_obs, dtype=np.int32)
    B = pdata.shape[1]
    out = np.empty((N + 1, 5, B), dtype=np.float64)
    _check(lib, lib.mpcx_initial_guess(pdata.ctypes.data, n_obs.ctypes.data if n_obs is not None else None,
                                       width.ctypes.data if width is not None else None,
                                       B, N, out.ctypes.data, int(device)),
           "mpcx_initial_guess")
    return out
