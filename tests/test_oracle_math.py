"""Building blocks of the oracle against independent references."""
import ctypes as C

import numpy as np

from mpc_amd.core_solver import solve_groups

import cases


def test_sincos_accuracy(oracle):
    lib = oracle.load()
    rng = np.random.default_rng(3)
    xs = np.concatenate([rng.uniform(-8, 8, 5000), rng.uniform(-300, 300, 1000), [0.0, np.pi, -np.pi / 2, 1e-9]])
    s, c = C.c_double(), C.c_double()
    worst = 0.0
    for x in xs:
        lib.mpcx_oracle_sincos(float(x), C.byref(s), C.byref(c))
        worst = max(worst, abs(s.value - np.sin(x)) / np.spacing(max(abs(np.sin(x)), 1e-300)),
                    abs(c.value - np.cos(x)) / np.spacing(max(abs(np.cos(x)), 1e-300)))
    assert worst <= 2.0, worst                     # within 2 ulp of libm
    lib.mpcx_oracle_sincos(float("inf"), C.byref(s), C.byref(c))
    assert np.isnan(s.value) and np.isnan(c.value)


def _rk4(lib, h, L, x, u):
    xn = np.zeros(5); A = np.zeros(25); Bm = np.zeros(10)
    dp = lambda a: a.ctypes.data_as(C.POINTER(C.c_double))
    lib.mpcx_oracle_rk4(h, L, dp(np.ascontiguousarray(x, dtype=float)), dp(np.ascontiguousarray(u, dtype=float)), dp(xn), dp(A), dp(Bm))
    return xn, A.reshape(5, 5), Bm.reshape(5, 2)


def test_rk4_step_and_sensitivities(oracle):
    lib = oracle.load()
    h, L = 0.1, 2.7

    def f(x, u):
        return np.array([x[3] * np.cos(x[2]), x[3] * np.sin(x[2]), x[3] * np.tan(x[4]) / L, u[0], u[1]])

    def step(x, u):
        k1 = f(x, u); k2 = f(x + h / 2 * k1, u); k3 = f(x + h / 2 * k2, u); k4 = f(x + h * k3, u)
        return x + h / 6 * (k1 + 2 * k2 + 2 * k3 + k4)

    rng = np.random.default_rng(1)
    for _ in range(20):
        x = rng.uniform([-5, -5, -3, -3, -0.5], [5, 5, 3, 3, 0.5])
        u = rng.uniform([-2, -0.4], [2, 0.4])
        xn, A, Bm = _rk4(lib, h, L, x, u)
        np.testing.assert_allclose(xn, step(x, u), rtol=0, atol=1e-13)
        eps = 1e-6
        for j in range(5):
            d = np.zeros(5); d[j] = eps
            np.testing.assert_allclose(A[:, j], (step(x + d, u) - step(x - d, u)) / (2 * eps), atol=1e-8)
        for j in range(2):
            d = np.zeros(2); d[j] = eps
            np.testing.assert_allclose(Bm[:, j], (step(x, u + d) - step(x, u - d)) / (2 * eps), atol=1e-8)
        # structure the kernels rely on
        assert A[0, 0] == 1 and A[1, 1] == 1 and A[2, 2] == 1 and A[3, 3] == 1 and A[4, 4] == 1
        assert A[0, 1] == 0 and A[1, 0] == 0 and A[3, 4] == 0 and A[2, 0] == 0 and Bm[3, 1] == 0 and Bm[4, 0] == 0
        assert Bm[3, 0] == Bm[4, 1] == (h / 6.0) * 6.0


def test_solution_is_a_kkt_point(oracle):
    """Converged output satisfies the problem's constraints to the tolerance."""
    for name in ("corner_turn", "line_change", "path_finder"):
        ego, sim = cases.make_scenario(name)
        st, _, _, res = solve_groups([ego], [sim], solver=oracle.solve_packed)
        assert st[0] == 0
        x, u = res[0]["x"], res[0]["u"]
        tol = 2e-3
        assert np.abs(x[0] - np.asarray(ego.state_start, float)).max() < 1e-9
        assert (u[:, 0] <= ego.acceleration_max + tol).all() and (u[:, 0] >= ego.acceleration_min - tol).all()
        assert (u[:, 1] <= ego.steering_rate_max + tol).all() and (u[:, 1] >= ego.steering_rate_min - tol).all()
        assert (np.abs(x[:, 4]) <= 0.5 + tol).all() and (x[1:, 3] >= ego.velocity_min - tol).all()
        eps = np.array([0.1, 0.1, 0.01, 0.01, 0.01])
        assert (np.abs(x[-1] - np.asarray(ego.state_final, float)) <= eps + tol).all()
        from mpc_amd import corridor_cal
        c1, a1, b1, c2, a2, b2 = corridor_cal(tuple(ego.state_start[:2]), tuple(ego.state_final[:2]), ego.corridor_width)
        assert (c1 * x[:, 1] + a1 * x[:, 0] + b1 > -tol).all() and (c2 * x[:, 1] + a2 * x[:, 0] + b2 < tol).all()


def test_input_guess_warm_start(oracle):
    """A converged solution handed back as state AND input guess passes the residual test
    at once (0 QP solves); with the state guess alone (inputs zero, as the reference starts)
    it does not."""
    import copy
    from mpc_amd.synthetic import synthetic_batch
    packed = synthetic_batch(256)
    ref = oracle.solve_packed(packed)
    ok = np.nonzero(ref["status"] == 0)[0]
    assert ok.size >= 10
    sub = [packed.slice(int(b), int(b) + 1) for b in ok[:10]]
    for b, p in zip(ok[:10], sub):
        p.xguess = np.ascontiguousarray(np.transpose(ref["X"][b:b + 1], (1, 2, 0)))
        cold = oracle.solve_packed(p)
        p.uguess = np.ascontiguousarray(np.transpose(ref["U"][b:b + 1], (1, 2, 0)))
        warm = oracle.solve_packed(p)
        assert warm["status"][0] == 0 and warm["sqp_iter"][0] <= 1
        assert cold["sqp_iter"][0] >= 1
        assert np.allclose(warm["X"][0], ref["X"][b], atol=1e-3)


def test_plant_step_matches_oracle_rk4(oracle):
    from mpc_amd.rti import plant_step
    import ctypes as C
    lib = oracle.load()
    rng = np.random.default_rng(3)
    x = rng.uniform(-1, 1, (20, 5)); x[:, 4] *= 0.4
    u = rng.uniform(-1, 1, (20, 2))
    h = np.full(20, 0.1)
    xn = plant_step(np, x, u, h, np.full(20, 2.7))
    for i in range(20):
        out = (C.c_double * 5)()
        A = (C.c_double * 25)()
        Bm = (C.c_double * 10)()
        lib.mpcx_oracle_rk4(0.1, 2.7, (C.c_double * 5)(*x[i]), (C.c_double * 2)(*u[i]), out, A, Bm)
        assert np.allclose(np.array(out[:]), xn[i], rtol=0, atol=1e-13)
