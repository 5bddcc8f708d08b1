"""GPU: the entry point that allocates nothing (mpcx_solve_batch_static, workspace = the device
array built into the library) against mpcx_solve_batch_device with a caller workspace, through
the C ABI: bit-identical outputs on two calls in a row, the second with a smaller input, no change
of the device's free memory across a call, the same bits from a captured and replayed call, a
refused overlapping call, and a clear refusal of a horizon the built-in scratch cannot hold."""
import ctypes as C

import numpy as np
import pytest
import torch

from mpc_amd import _lib
from mpc_amd.batch import pack_problems
from mpc_amd.synthetic import synthetic_batch

import cases

pytestmark = pytest.mark.gpu

KEYS = ("X", "U", "status", "sqp_iter", "qp_iter", "cost", "res")


def _device_solve(lib, packed, static, setup=False):
    dev = torch.device("cuda", 0)
    B, N = packed.B, packed.N
    d_pd = torch.from_numpy(packed.pdata).to(dev)
    d_no = torch.from_numpy(packed.n_obs).to(dev)
    d_xg = torch.from_numpy(packed.guess()).to(dev)
    f64, i32 = dict(dtype=torch.float64, device=dev), dict(dtype=torch.int32, device=dev)
    out = {"X": torch.full((B, N + 1, 5), np.nan, **f64), "U": torch.full((B, N, 2), np.nan, **f64),
           "status": torch.full((B,), -1, **i32), "sqp_iter": torch.full((B,), -1, **i32),
           "qp_iter": torch.full((B,), -1, **i32), "cost": torch.full((B,), np.nan, **f64),
           "res": torch.full((4, B), np.nan, **f64)}
    bs = _lib.MpcxBatch(B, N, d_pd.data_ptr(), d_no.data_ptr(), d_xg.data_ptr(), None)
    rs = _lib.MpcxResult(*(out[k].data_ptr() for k in KEYS))
    o = _lib.default_opts(lib)
    stream = torch.cuda.current_stream().cuda_stream
    if static:
        # device-wide free memory (hipMemGetInfo), which a hipMalloc inside the library would lower;
        # torch's own counters only see torch's allocator
        torch.cuda.synchronize()
        before, before_torch = torch.cuda.mem_get_info()[0], torch.cuda.memory_allocated()
        rc = lib.mpcx_solve_batch_static(C.byref(bs), C.byref(o), C.byref(rs), stream)
        # (the set-up call on a device may: the array's address is looked up then, and the driver makes the
        # library's device globals resident at that point)
        assert setup or (torch.cuda.mem_get_info()[0] == before and torch.cuda.memory_allocated() == before_torch)
        torch.cuda.synchronize()
        assert setup or torch.cuda.mem_get_info()[0] == before
    else:
        ws_bytes = lib.mpcx_workspace_bytes(B, N)
        ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
        rc = lib.mpcx_solve_batch_device(C.byref(bs), C.byref(o), C.byref(rs), ws.data_ptr(), ws_bytes, stream)
    assert rc == 0, lib.mpcx_last_error()
    torch.cuda.synchronize()
    return {k: v.cpu().numpy() for k, v in out.items()}


def test_static_scratch_entry_point_matches_device_entry_point():
    lib = _lib.load_library()
    first = synthetic_batch(3000)                                     # more problems than the scratch holds at once
    second = pack_problems(*cases.obstacle_problems(40, 13.0))       # smaller, another horizon, obstacle rows
    _device_solve(lib, synthetic_batch(8), static=True, setup=True)   # set-up call: symbol lookup, code load
    for packed in (first, second):
        want = _device_solve(lib, packed, static=False)
        got = _device_solve(lib, packed, static=True)
        cases.assert_same_solution(got, want, f"static scratch, B={packed.B} N={packed.N}")
        assert (got["status"] >= 0).all() and (want["status"] == 0).any()
    assert lib.mpcx_static_max_wavefronts(first.N) >= 1 and second.N <= lib.mpcx_static_max_horizon()
    # a horizon whose single-wavefront workspace exceeds the built-in scratch is refused, nothing is launched
    big = synthetic_batch(2, duration=200.0)
    assert big.N == 2000 > lib.mpcx_static_max_horizon()
    dev = torch.device("cuda", 0)
    d_pd = torch.from_numpy(big.pdata).to(dev)
    d_no = torch.from_numpy(big.n_obs).to(dev)
    dummy = torch.zeros(16, dtype=torch.float64, device=dev)
    bs = _lib.MpcxBatch(big.B, big.N, d_pd.data_ptr(), d_no.data_ptr(), None, None)
    rs = _lib.MpcxResult(*([dummy.data_ptr()] * 7))
    o = _lib.default_opts(lib)
    assert lib.mpcx_solve_batch_static(C.byref(bs), C.byref(o), C.byref(rs), None) == -1
    assert b"built-in scratch" in lib.mpcx_last_error()


def _device_arrays(packed, fill=np.nan):
    dev = torch.device("cuda", 0)
    B, N = packed.B, packed.N
    keep = [torch.from_numpy(packed.pdata).to(dev), torch.from_numpy(packed.n_obs).to(dev),
            torch.from_numpy(packed.guess()).to(dev)]
    f64, i32 = dict(dtype=torch.float64, device=dev), dict(dtype=torch.int32, device=dev)
    out = {"X": torch.full((B, N + 1, 5), fill, **f64), "U": torch.full((B, N, 2), fill, **f64),
           "status": torch.full((B,), -1, **i32), "sqp_iter": torch.full((B,), -1, **i32),
           "qp_iter": torch.full((B,), -1, **i32), "cost": torch.full((B,), fill, **f64),
           "res": torch.full((4, B), fill, **f64)}
    bs = _lib.MpcxBatch(B, N, keep[0].data_ptr(), keep[1].data_ptr(), keep[2].data_ptr(), None)
    rs = _lib.MpcxResult(*(out[k].data_ptr() for k in KEYS))
    return keep, out, bs, rs


def test_static_entry_point_under_graph_capture():
    """What the entry point is for: captured into a HIP graph on a side stream (capture forbids
    allocation) and replayed, it leaves the bits of an eager call."""
    lib = _lib.load_library()
    packed = synthetic_batch(500)
    want = _device_solve(lib, packed, static=True, setup=True)     # eager (the set-up call when run alone)
    keep, out, bs, rs = _device_arrays(packed)
    o = _lib.default_opts(lib)
    side = torch.cuda.Stream()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph, stream=side):
        rc = lib.mpcx_solve_batch_static(C.byref(bs), C.byref(o), C.byref(rs), torch.cuda.current_stream().cuda_stream)
    assert rc == 0, lib.mpcx_last_error()
    torch.cuda.synchronize()
    assert int(out["status"].min()) == -1                    # captured, not run
    for _ in range(2):
        for v in out.values():
            v.fill_(-1)
        graph.replay()
        torch.cuda.synchronize()
        cases.assert_same_solution({k: v.cpu().numpy() for k, v in out.items()}, want, "captured static call")


def test_overlapping_static_calls_are_refused():
    """All static calls on a GPU share one scratch.  Queued on one stream they run one after the other;
    a call on another stream while the previous one is still running is refused without a launch."""
    lib = _lib.load_library()
    packed = synthetic_batch(20000)                         # long enough to be still running at the second call
    want = _device_solve(lib, packed, static=False)
    keep, out, bs, rs = _device_arrays(packed)
    keep2, out2, bs2, rs2 = _device_arrays(packed)
    o = _lib.default_opts(lib)
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    torch.cuda.synchronize()
    assert lib.mpcx_solve_batch_static(C.byref(bs), C.byref(o), C.byref(rs), s1.cuda_stream) == 0
    assert lib.mpcx_solve_batch_static(C.byref(bs2), C.byref(o), C.byref(rs2), s2.cuda_stream) == -1
    assert b"overlapping" in lib.mpcx_last_error()
    assert lib.mpcx_solve_batch_static(C.byref(bs2), C.byref(o), C.byref(rs2), s1.cuda_stream) == 0   # same stream: serialised
    torch.cuda.synchronize()
    for got in (out, out2):
        cases.assert_same_solution({k: v.cpu().numpy() for k, v in got.items()}, want, "serialised static calls")
    assert lib.mpcx_solve_batch_static(C.byref(bs2), C.byref(o), C.byref(rs2), s2.cuda_stream) == 0   # idle again: any stream
    torch.cuda.synchronize()
