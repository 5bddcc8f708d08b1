// mpcx_kernels.hip — batched kinematic-car OCP solve on gfx950 (MI355X).
//
// One LANE per problem, 64 problems per wavefront, the whole SQP solve fused in
// one persistent kernel.  The per-stage 7x7 algebra of the Riccati recursion is
// a serial dependency chain over the N stages; giving each lane its own chain
// keeps all 64 lanes of every FP64 instruction busy, which a wavefront that
// cooperates on one 7x7 stage cannot.  Each lane's working set (iterate,
// linearisation, multipliers, Riccati factors of all N+1 stages) lives in a
// workspace in HBM laid out [wave][stage][field][lane], so every load and store
// of a wavefront is one contiguous 512-byte segment.
//
// The problem solved is the one reference src/core_solver.py:127-351 poses and
// hands to acados at :354/:502; the algorithm is the one selected by its
// options block :242-253 (SQP, ERK4 + forward sensitivities, Gauss-Newton +
// Levenberg-Marquardt, interior-point QP with stage-wise Riccati recursion).
//
// Arithmetic contract: built with -ffp-contract=off; every fused multiply-add
// is an explicit fma(); sin/cos are the polynomials below, not the device
// library's.  The order of operations is fixed, so a CPU program that performs
// the same sequence gets the same bits.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/mpcx.h"
#include "mpcx_kernels.h"

#define NX 5
#define NU 2
#define NZ 7
#define ROW_G0 7

#define DEV __device__ __forceinline__

namespace {

// ---------------------------------------------------------------------------
// workspace field map (doubles per stage), R = rows per stage = 9 + NH
// ---------------------------------------------------------------------------
template <int NH>
struct Map {
    static constexpr int R = 9 + NH;
    static constexpr int G = 2 + NH;           // general rows
    static constexpr int X = 0;                // 5  NLP state
    static constexpr int U = X + 5;            // 2  NLP input
    static constexpr int PI = U + 2;           // 5  NLP dynamics multiplier
    static constexpr int NLL = PI + 5;         // R  NLP multipliers, lower sides
    static constexpr int NLU = NLL + R;        // R
    static constexpr int A = NLU + R;          // 25 row-major
    static constexpr int B = A + 25;           // 10 row-major
    static constexpr int BV = B + 10;          // 5  dynamics gap
    static constexpr int C = BV + 5;           // R  row values
    static constexpr int JX = C + R;           // G  general-row Jacobians
    static constexpr int JY = JX + G;          // G
    static constexpr int V = JY + G;           // 7  QP primal [u; x]
    static constexpr int QPI = V + 7;          // 5
    static constexpr int LAML = QPI + 5;       // R
    static constexpr int LAMU = LAML + R;
    static constexpr int TL = LAMU + R;
    static constexpr int TU = TL + R;
    static constexpr int RG = TU + R;          // 7
    static constexpr int RB = RG + 7;          // 5
    static constexpr int RDL = RB + 5;         // R
    static constexpr int RDU = RDL + R;
    static constexpr int HT = RDU + R;         // 10: diag(7), hxx, hxy, hyy
    static constexpr int GT = HT + 10;         // 7
    static constexpr int K = GT + 7;           // 10 row-major [2][5]
    static constexpr int KV = K + 10;          // 2
    static constexpr int LDL = KV + 2;         // 3: l10, d0, d1
    static constexpr int P = LDL + 3;          // 25 full symmetric
    static constexpr int PV = P + 25;          // 5
    static constexpr int DV = PV + 5;          // 7
    static constexpr int DPI = DV + 7;         // 5
    static constexpr int DVA = DPI + 5;        // 7
    static constexpr int F = DVA + 7;          // doubles per stage
};

struct Lane {
    double *w;            // workspace of this lane: element (k, f) at w[(k*F + f) * 64]
    const double *pd;     // problem table column of this lane: field f at pd[f * B]
    size_t B;
    int N;
    int nobs;
    int F;
};

DEV double ldw(const Lane &L, int k, int f) { return L.w[((size_t)k * L.F + f) * 64]; }
DEV void stw(const Lane &L, int k, int f, double v) { L.w[((size_t)k * L.F + f) * 64] = v; }
DEV double ldp(const Lane &L, int f) { return L.pd[(size_t)f * L.B]; }

DEV double dabs(double a) { return (a < 0.0) ? -a : a; }
DEV double dmax(double a, double b) { return (b > a) ? b : a; }
DEV bool finite_d(double a) { return (a - a) == 0.0; }

// ---------------------------------------------------------------------------
// sin / cos: Cody-Waite reduction by pi/2 in two FMA steps + minimax
// polynomials on [-pi/4, pi/4] (fdlibm k_sin / k_cos coefficients)
// ---------------------------------------------------------------------------
DEV void mx_sincos(double x, double &s, double &c)
{
    const double TWO_OVER_PI = 0.63661977236758134308;
    const double MAGIC = 6755399441055744.0;
    const double P1 = 1.5707963267948966;
    const double P2 = 6.123233995736766e-17;
    const double S1 = -1.66666666666666324348e-01, S2 = 8.33333333332248946124e-03,
                 S3 = -1.98412698298579493134e-04, S4 = 2.75573137070700676789e-06,
                 S5 = -2.50507602534068634195e-08, S6 = 1.58969099521155010221e-10;
    const double C1 = 4.16666666666666019037e-02, C2 = -1.38888888888741095749e-03,
                 C3 = 2.48015872894767294178e-05, C4 = -2.75573143513906633035e-07,
                 C5 = 2.08757232129817482790e-09, C6 = -1.13596475577881948265e-11;
    if (!(dabs(x) <= 1.0e6)) {
        s = c = __builtin_nan("");
        return;
    }
    double kf = (x * TWO_OVER_PI + MAGIC) - MAGIC;
    double r = fma(-kf, P1, x);
    r = fma(-kf, P2, r);
    long long q = ((long long)kf) & 3;
    double z = r * r;
    double ps = fma(z, S6, S5);
    ps = fma(z, ps, S4);
    ps = fma(z, ps, S3);
    ps = fma(z, ps, S2);
    ps = fma(z, ps, S1);
    double sn = fma(r * z, ps, r);
    double pc = fma(z, C6, C5);
    pc = fma(z, pc, C4);
    pc = fma(z, pc, C3);
    pc = fma(z, pc, C2);
    pc = fma(z, pc, C1);
    double hz = 0.5 * z;
    double w = 1.0 - hz;
    double cs = w + ((1.0 - w) - hz + (z * z) * pc);
    if (q == 0) { s = sn; c = cs; }
    else if (q == 1) { s = cs; c = -sn; }
    else if (q == 2) { s = -sn; c = -cs; }
    else { s = -cs; c = sn; }
}

// ---------------------------------------------------------------------------
// model (reference core_solver.py:148-155) and its variational equation
// ---------------------------------------------------------------------------
struct Jac { double j02, j03, j12, j13, j23, j24; };

DEV void model_eval(double invL, const double *x, const double *u, double *f, Jac &J)
{
    double sth, cth, sde, cde;
    mx_sincos(x[2], sth, cth);
    mx_sincos(x[4], sde, cde);
    double tde = sde / cde;
    double v = x[3];
    f[0] = v * cth;
    f[1] = v * sth;
    f[2] = (v * tde) * invL;
    f[3] = u[0];
    f[4] = u[1];
    J.j02 = -f[1];
    J.j03 = cth;
    J.j12 = f[0];
    J.j13 = sth;
    J.j23 = tde * invL;
    J.j24 = (v * fma(tde, tde, 1.0)) * invL;
}

// Kout = J * S + dF/du; S, Kout are 5 x 7, columns (x0..x4, u0, u1)
DEV void vde_eval(const Jac &J, const double (*S)[NZ], double (*Ko)[NZ])
{
#pragma unroll
    for (int c = 0; c < NZ; ++c) {
        Ko[0][c] = fma(J.j03, S[3][c], J.j02 * S[2][c]);
        Ko[1][c] = fma(J.j13, S[3][c], J.j12 * S[2][c]);
        Ko[2][c] = fma(J.j24, S[4][c], J.j23 * S[3][c]);
        Ko[3][c] = 0.0;
        Ko[4][c] = 0.0;
    }
    Ko[3][NX + 0] = 1.0;
    Ko[4][NX + 1] = 1.0;
}

DEV void rk4_sens(double h, double invL, const double *x, const double *u,
                  double *xn, double (*A)[NX], double (*Bm)[NU])
{
    const double hh = 0.5 * h, h6 = h / 6.0;
    double k1[NX], k2[NX], k3[NX], k4[NX], xs[NX];
    double S0[NX][NZ], Ss[NX][NZ], K1[NX][NZ], K2[NX][NZ], K3[NX][NZ], K4[NX][NZ];
    Jac J;
#pragma unroll
    for (int i = 0; i < NX; ++i)
#pragma unroll
        for (int c = 0; c < NZ; ++c) S0[i][c] = (i == c) ? 1.0 : 0.0;

    model_eval(invL, x, u, k1, J);
    vde_eval(J, S0, K1);
#pragma unroll
    for (int i = 0; i < NX; ++i) {
        xs[i] = fma(hh, k1[i], x[i]);
#pragma unroll
        for (int c = 0; c < NZ; ++c) Ss[i][c] = fma(hh, K1[i][c], S0[i][c]);
    }
    model_eval(invL, xs, u, k2, J);
    vde_eval(J, Ss, K2);
#pragma unroll
    for (int i = 0; i < NX; ++i) {
        xs[i] = fma(hh, k2[i], x[i]);
#pragma unroll
        for (int c = 0; c < NZ; ++c) Ss[i][c] = fma(hh, K2[i][c], S0[i][c]);
    }
    model_eval(invL, xs, u, k3, J);
    vde_eval(J, Ss, K3);
#pragma unroll
    for (int i = 0; i < NX; ++i) {
        xs[i] = fma(h, k3[i], x[i]);
#pragma unroll
        for (int c = 0; c < NZ; ++c) Ss[i][c] = fma(h, K3[i][c], S0[i][c]);
    }
    model_eval(invL, xs, u, k4, J);
    vde_eval(J, Ss, K4);
#pragma unroll
    for (int i = 0; i < NX; ++i) {
        double t = k1[i] + k4[i];
        t = fma(2.0, k2[i], t);
        t = fma(2.0, k3[i], t);
        xn[i] = fma(h6, t, x[i]);
#pragma unroll
        for (int c = 0; c < NZ; ++c) {
            double ts = K1[i][c] + K4[i][c];
            ts = fma(2.0, K2[i][c], ts);
            ts = fma(2.0, K3[i][c], ts);
            double sv = fma(h6, ts, S0[i][c]);
            if (c < NX) A[i][c] = sv; else Bm[i][c - NX] = sv;
        }
    }
}

// ---------------------------------------------------------------------------
// rows of a stage.  r: 0-1 input box, 2-6 state box, 7-8 corridor, 9.. obstacles
// ---------------------------------------------------------------------------
DEV void row_range(const Lane &L, int k, int &r0, int &r1)
{
    if (k == 0) { r0 = 0; r1 = NU; }
    else if (k < L.N) { r0 = 0; r1 = 9 + L.nobs; }
    else { r0 = NU; r1 = 9; }
}

DEV double row_lo(const Lane &L, int k, int r)
{
    if (r < NU) return ldp(L, MPCX_F_LBU + r);
    if (r < ROW_G0) return (k == L.N) ? ldp(L, MPCX_F_LBXE + r - NU) : ldp(L, MPCX_F_LBX + r - NU);
    if (r < 9) return ldp(L, MPCX_F_LG + r - ROW_G0);
    return ldp(L, MPCX_F_LH);
}

DEV double row_up(const Lane &L, int k, int r)
{
    if (r < NU) return ldp(L, MPCX_F_UBU + r);
    if (r < ROW_G0) return (k == L.N) ? ldp(L, MPCX_F_UBXE + r - NU) : ldp(L, MPCX_F_UBX + r - NU);
    if (r < 9) return ldp(L, MPCX_F_UG + r - ROW_G0);
    return ldp(L, MPCX_F_UH);
}

// J_r * v, v = [u; x]; jx, jy are the general row's Jacobian entries
template <int NH>
DEV double row_dot(const Lane &L, int k, int r, const double *v)
{
    typedef Map<NH> M;
    if (r < ROW_G0) return v[r];
    double jx = ldw(L, k, M::JX + r - ROW_G0), jy = ldw(L, k, M::JY + r - ROW_G0);
    return fma(jy, v[3], jx * v[2]);
}

// ---------------------------------------------------------------------------
// linearisation of one SQP iteration
// ---------------------------------------------------------------------------
template <int NH>
DEV void linearize(const Lane &L, const mpcx_opts &o)
{
    typedef Map<NH> M;
    const int N = L.N;
    const double h = ldp(L, MPCX_F_H);
    const double invL = 1.0 / ldp(L, MPCX_F_L);
    double xk[NX], xn1[NX];
#pragma unroll
    for (int i = 0; i < NX; ++i) xk[i] = ldw(L, 0, M::X + i);
    for (int k = 0; k <= N; ++k) {
        double uk[NU] = {0.0, 0.0};
        if (k < N) {
#pragma unroll
            for (int c = 0; c < NU; ++c) uk[c] = ldw(L, k, M::U + c);
#pragma unroll
            for (int i = 0; i < NX; ++i) xn1[i] = ldw(L, k + 1, M::X + i);
            double xn[NX], A[NX][NX], Bm[NX][NU];
            rk4_sens(h, invL, xk, uk, xn, A, Bm);
#pragma unroll
            for (int i = 0; i < NX; ++i) {
                stw(L, k, M::BV + i, xn[i] - xn1[i]);
#pragma unroll
                for (int j = 0; j < NX; ++j) stw(L, k, M::A + i * NX + j, A[i][j]);
#pragma unroll
                for (int c = 0; c < NU; ++c) stw(L, k, M::B + i * NU + c, Bm[i][c]);
            }
        }
        int r0, r1;
        row_range(L, k, r0, r1);
#pragma unroll
        for (int r = 0; r < M::R; ++r) {
            if (r < r0 || r >= r1) continue;
            double c;
            if (r < NU) c = uk[r];
            else if (r < ROW_G0) c = xk[r - NU];
            else if (r < 9) {
                double jx = ldp(L, MPCX_F_CG + 2 * (r - ROW_G0));
                double jy = ldp(L, MPCX_F_CG + 2 * (r - ROW_G0) + 1);
                stw(L, k, M::JX + r - ROW_G0, jx);
                stw(L, k, M::JY + r - ROW_G0, jy);
                c = fma(jy, xk[1], jx * xk[0]);
            } else {
                double ox = ldp(L, MPCX_F_OBS + 3 * (r - 9));
                double oy = ldp(L, MPCX_F_OBS + 3 * (r - 9) + 1);
                double rr = ldp(L, MPCX_F_OBS + 3 * (r - 9) + 2);
                double dx = xk[0] - ox, dy = xk[1] - oy;
                c = fma(dy, dy, dx * dx) - rr;
                stw(L, k, M::JX + r - ROW_G0, 2.0 * dx);
                stw(L, k, M::JY + r - ROW_G0, 2.0 * dy);
            }
            stw(L, k, M::C + r, c);
        }
#pragma unroll
        for (int i = 0; i < NX; ++i) xk[i] = xn1[i];
    }
}

// cost gradient g and Gauss-Newton + LM diagonal Hd of stage k at (x, u)
DEV void cost_terms(const Lane &L, const mpcx_opts &o, int k, const double *x, const double *u,
                    double *g, double *Hd)
{
    if (k < L.N) {
        const double h = ldp(L, MPCX_F_H);
        const double lmk = o.lm_scale_by_h ? h * o.lm : o.lm;
#pragma unroll
        for (int j = 0; j < NZ; ++j) {
            double hw = h * ldp(L, MPCX_F_W + j);
            double z = (j < NU) ? u[j] : (x[j - NU] - ldp(L, MPCX_F_GOAL + j - NU));
            Hd[j] = hw + lmk;
            g[j] = hw * z;
        }
    } else {
        Hd[0] = Hd[1] = 0.0;
        g[0] = g[1] = 0.0;
#pragma unroll
        for (int i = 0; i < NX; ++i) {
            double we = ldp(L, MPCX_F_WE + i);
            Hd[NU + i] = we + o.lm;
            g[NU + i] = we * (x[i] - ldp(L, MPCX_F_GOAL + i));
        }
    }
}

template <int NH>
DEV void load_xu(const Lane &L, int k, double *x, double *u)
{
    typedef Map<NH> M;
#pragma unroll
    for (int i = 0; i < NX; ++i) x[i] = ldw(L, k, M::X + i);
    u[0] = u[1] = 0.0;
    if (k < L.N) {
        u[0] = ldw(L, k, M::U + 0);
        u[1] = ldw(L, k, M::U + 1);
    }
}

template <int NH>
DEV void load_AB(const Lane &L, int k, double (*A)[NX], double (*Bm)[NU])
{
    typedef Map<NH> M;
#pragma unroll
    for (int i = 0; i < NX; ++i) {
#pragma unroll
        for (int j = 0; j < NX; ++j) A[i][j] = ldw(L, k, M::A + i * NX + j);
#pragma unroll
        for (int c = 0; c < NU; ++c) Bm[i][c] = ldw(L, k, M::B + i * NU + c);
    }
}

// NLP residuals: stationarity, dynamics gap, inequality violation, complementarity
template <int NH>
DEV void nlp_residuals(const Lane &L, const mpcx_opts &o, double *res)
{
    typedef Map<NH> M;
    const int N = L.N;
    double rs = 0.0, re = 0.0, ri = 0.0, rc = 0.0;
    for (int k = 0; k <= N; ++k) {
        double x[NX], u[NU], g[NZ], Hd[NZ], gl[NZ];
        load_xu<NH>(L, k, x, u);
        cost_terms(L, o, k, x, u, g, Hd);
#pragma unroll
        for (int j = 0; j < NZ; ++j) gl[j] = g[j];
        if (k < N) {
            double A[NX][NX], Bm[NX][NU], pn[NX];
            load_AB<NH>(L, k, A, Bm);
#pragma unroll
            for (int i = 0; i < NX; ++i) pn[i] = ldw(L, k + 1, M::PI + i);
#pragma unroll
            for (int c = 0; c < NU; ++c)
#pragma unroll
                for (int i = 0; i < NX; ++i) gl[c] = fma(Bm[i][c], pn[i], gl[c]);
#pragma unroll
            for (int c = 0; c < NX; ++c)
#pragma unroll
                for (int i = 0; i < NX; ++i) gl[NU + c] = fma(A[i][c], pn[i], gl[NU + c]);
        }
        if (k > 0) {
#pragma unroll
            for (int i = 0; i < NX; ++i) gl[NU + i] = gl[NU + i] - ldw(L, k, M::PI + i);
        }
        int r0, r1;
        row_range(L, k, r0, r1);
#pragma unroll
        for (int r = 0; r < M::R; ++r) {
            if (r < r0 || r >= r1) continue;
            double ll = ldw(L, k, M::NLL + r), lu = ldw(L, k, M::NLU + r);
            double dl = lu - ll;
            if (r < ROW_G0) gl[r] = gl[r] + dl;
            else {
                gl[2] = fma(ldw(L, k, M::JX + r - ROW_G0), dl, gl[2]);
                gl[3] = fma(ldw(L, k, M::JY + r - ROW_G0), dl, gl[3]);
            }
            double c = ldw(L, k, M::C + r);
            double sl = c - row_lo(L, k, r), su = row_up(L, k, r) - c;
            ri = dmax(ri, -sl);
            ri = dmax(ri, -su);
            rc = dmax(rc, dabs(ll * sl));
            rc = dmax(rc, dabs(lu * su));
        }
#pragma unroll
        for (int j = 0; j < NZ; ++j) {
            bool in = (j < NU) ? (k < N) : (k > 0);
            if (in) rs = dmax(rs, dabs(gl[j]));
        }
        if (k < N) {
#pragma unroll
            for (int i = 0; i < NX; ++i) re = dmax(re, dabs(ldw(L, k, M::BV + i)));
        }
        if (k == 0) {
#pragma unroll
            for (int i = 0; i < NX; ++i) ri = dmax(ri, dabs(ldp(L, MPCX_F_X0 + i) - x[i]));
        }
    }
    res[0] = rs; res[1] = re; res[2] = ri; res[3] = rc;
}

// ---------------------------------------------------------------------------
// interior-point QP
// ---------------------------------------------------------------------------
struct QpRes { double g, b, d, m, mu; };

template <int NH>
DEV void qp_residuals(const Lane &L, const mpcx_opts &o, int m_total, QpRes &R)
{
    typedef Map<NH> M;
    const int N = L.N;
    double ng = 0.0, nb = 0.0, nd = 0.0, nm = 0.0, sum = 0.0;
    for (int k = 0; k <= N; ++k) {
        double x[NX], u[NU], g[NZ], Hd[NZ], v[NZ], rg[NZ];
        load_xu<NH>(L, k, x, u);
        cost_terms(L, o, k, x, u, g, Hd);
#pragma unroll
        for (int j = 0; j < NZ; ++j) {
            v[j] = ldw(L, k, M::V + j);
            rg[j] = fma(Hd[j], v[j], g[j]);
        }
        double A[NX][NX], Bm[NX][NU];
        if (k < N) {
            double pn[NX];
            load_AB<NH>(L, k, A, Bm);
#pragma unroll
            for (int i = 0; i < NX; ++i) pn[i] = ldw(L, k + 1, M::QPI + i);
#pragma unroll
            for (int c = 0; c < NU; ++c)
#pragma unroll
                for (int i = 0; i < NX; ++i) rg[c] = fma(Bm[i][c], pn[i], rg[c]);
#pragma unroll
            for (int c = 0; c < NX; ++c)
#pragma unroll
                for (int i = 0; i < NX; ++i) rg[NU + c] = fma(A[i][c], pn[i], rg[NU + c]);
        }
        if (k > 0) {
#pragma unroll
            for (int i = 0; i < NX; ++i) rg[NU + i] = rg[NU + i] - ldw(L, k, M::QPI + i);
        }
        int r0, r1;
        row_range(L, k, r0, r1);
#pragma unroll
        for (int r = 0; r < M::R; ++r) {
            if (r < r0 || r >= r1) continue;
            double jv = row_dot<NH>(L, k, r, v);
            double ll = ldw(L, k, M::LAML + r), lu = ldw(L, k, M::LAMU + r);
            double tl = ldw(L, k, M::TL + r), tu = ldw(L, k, M::TU + r);
            double dl = lu - ll;
            if (r < ROW_G0) rg[r] = rg[r] + dl;
            else {
                rg[2] = fma(ldw(L, k, M::JX + r - ROW_G0), dl, rg[2]);
                rg[3] = fma(ldw(L, k, M::JY + r - ROW_G0), dl, rg[3]);
            }
            double c = ldw(L, k, M::C + r);
            double rdl = ((row_lo(L, k, r) - c) - jv) + tl;
            double rdu = (jv - (row_up(L, k, r) - c)) + tu;
            stw(L, k, M::RDL + r, rdl);
            stw(L, k, M::RDU + r, rdu);
            nd = dmax(nd, dabs(rdl));
            nd = dmax(nd, dabs(rdu));
            double ml = ll * tl, mu_ = lu * tu;
            nm = dmax(nm, dabs(ml));
            nm = dmax(nm, dabs(mu_));
            sum = sum + ml;
            sum = sum + mu_;
        }
#pragma unroll
        for (int j = 0; j < NZ; ++j) {
            stw(L, k, M::RG + j, rg[j]);
            bool in = (j < NU) ? (k < N) : (k > 0);
            if (in) ng = dmax(ng, dabs(rg[j]));
        }
        if (k < N) {
#pragma unroll
            for (int i = 0; i < NX; ++i) {
                double acc = ldw(L, k, M::BV + i);
#pragma unroll
                for (int j = 0; j < NX; ++j) acc = fma(A[i][j], v[NU + j], acc);
#pragma unroll
                for (int c = 0; c < NU; ++c) acc = fma(Bm[i][c], v[c], acc);
                double rb = acc - ldw(L, k + 1, M::V + NU + i);
                stw(L, k, M::RB + i, rb);
                nb = dmax(nb, dabs(rb));
            }
        }
    }
    R.g = ng; R.b = nb; R.d = nd; R.m = nm;
    R.mu = sum / (double)m_total;
}

template <int NH>
DEV void qp_init(const Lane &L, const mpcx_opts &o)
{
    typedef Map<NH> M;
    const int N = L.N;
    const double thr0 = o.thr0, mu0 = o.mu0;
    for (int k = 0; k <= N; ++k) {
        double v[NZ];
#pragma unroll
        for (int j = 0; j < NZ; ++j) v[j] = 0.0;
#pragma unroll
        for (int i = 0; i < NX; ++i) stw(L, k, M::QPI + i, 0.0);
        if (k == 0) {
#pragma unroll
            for (int i = 0; i < NX; ++i) v[NU + i] = ldp(L, MPCX_F_X0 + i) - ldw(L, 0, M::X + i);
        }
        int r0, r1;
        row_range(L, k, r0, r1);
#pragma unroll
        for (int r = 0; r < M::R; ++r) {
            if (r < r0 || r >= r1) continue;
            double c = ldw(L, k, M::C + r);
            double dl = row_lo(L, k, r) - c, du = row_up(L, k, r) - c;
            double tl, tu;
            if (r < ROW_G0) {
                tl = v[r] - dl;
                tu = du - v[r];
                if (tl < thr0) {
                    if (tu < thr0) {
                        v[r] = 0.5 * (dl + du);
                        tl = thr0;
                        tu = thr0;
                    } else {
                        tl = thr0;
                        v[r] = dl + thr0;
                    }
                } else if (tu < thr0) {
                    tu = thr0;
                    v[r] = du - thr0;
                }
            } else {
                double jv = row_dot<NH>(L, k, r, v);
                tl = jv - dl;
                tu = du - jv;
                if (tl < thr0) tl = thr0;
                if (tu < thr0) tu = thr0;
            }
            stw(L, k, M::TL + r, tl);
            stw(L, k, M::TU + r, tu);
            stw(L, k, M::LAML + r, mu0 / tl);
            stw(L, k, M::LAMU + r, mu0 / tu);
        }
#pragma unroll
        for (int j = 0; j < NZ; ++j) stw(L, k, M::V + j, v[j]);
    }
}

// complementarity target of one row side for the three kinds of step
//   mode 0 affine: lam*t;  mode 1 corrector: lam*t + dt_a*dlam_a - sigma*mu;
//   mode 2 centering: lam*t - sigma*mu
DEV void row_rm(int mode, double sigma_mu, double jda, double ll, double lu, double tl, double tu,
                double til, double tiu, double rdl, double rdu, double &rml, double &rmu)
{
    rml = ll * tl;
    rmu = lu * tu;
    if (mode == 1) {
        double dtl = jda - rdl;
        double dtu = -jda - rdu;
        double dll = -(fma(ll, dtl, rml) * til);
        double dlu = -(fma(lu, dtu, rmu) * tiu);
        rml = fma(dtl, dll, rml) - sigma_mu;
        rmu = fma(dtu, dlu, rmu) - sigma_mu;
    } else if (mode == 2) {
        rml = rml - sigma_mu;
        rmu = rmu - sigma_mu;
    }
}

template <int NH>
DEV void qp_build_rhs(const Lane &L, const mpcx_opts &o, int mode, double sigma_mu)
{
    typedef Map<NH> M;
    const int N = L.N;
    for (int k = 0; k <= N; ++k) {
        double Ht[NZ], hxx = 0.0, hxy = 0.0, hyy = 0.0, gt[NZ], dva[NZ];
        if (mode == 0) {
            double x[NX], u[NU], g[NZ];
            load_xu<NH>(L, k, x, u);
            cost_terms(L, o, k, x, u, g, Ht);
        }
#pragma unroll
        for (int j = 0; j < NZ; ++j) {
            gt[j] = ldw(L, k, M::RG + j);
            dva[j] = (mode == 1) ? ldw(L, k, M::DVA + j) : 0.0;
        }
        int r0, r1;
        row_range(L, k, r0, r1);
#pragma unroll
        for (int r = 0; r < M::R; ++r) {
            if (r < r0 || r >= r1) continue;
            double ll = ldw(L, k, M::LAML + r), lu = ldw(L, k, M::LAMU + r);
            double tl = ldw(L, k, M::TL + r), tu = ldw(L, k, M::TU + r);
            double rdl = ldw(L, k, M::RDL + r), rdu = ldw(L, k, M::RDU + r);
            double til = 1.0 / tl, tiu = 1.0 / tu;
            double jx = 0.0, jy = 0.0;
            if (r >= ROW_G0) {
                jx = ldw(L, k, M::JX + r - ROW_G0);
                jy = ldw(L, k, M::JY + r - ROW_G0);
            }
            double jda = 0.0;
            if (mode == 1) jda = (r < ROW_G0) ? dva[r] : fma(jy, dva[3], jx * dva[2]);
            double rml, rmu;
            row_rm(mode, sigma_mu, jda, ll, lu, tl, tu, til, tiu, rdl, rdu, rml, rmu);
            double ql = fma(-ll, rdl, rml) * til;
            double qu = fma(-lu, rdu, rmu) * tiu;
            double q = ql - qu;
            if (r < ROW_G0) gt[r] = gt[r] + q;
            else {
                gt[2] = fma(jx, q, gt[2]);
                gt[3] = fma(jy, q, gt[3]);
            }
            if (mode == 0) {
                double w = ll * til + lu * tiu;
                if (r < ROW_G0) Ht[r] = Ht[r] + w;
                else {
                    double wx = w * jx, wy = w * jy;
                    hxx = fma(wx, jx, hxx);
                    hxy = fma(wy, jx, hxy);
                    hyy = fma(wy, jy, hyy);
                }
            }
        }
#pragma unroll
        for (int j = 0; j < NZ; ++j) stw(L, k, M::GT + j, gt[j]);
        if (mode == 0) {
#pragma unroll
            for (int j = 0; j < NZ; ++j) stw(L, k, M::HT + j, Ht[j]);
            stw(L, k, M::HT + 7, hxx);
            stw(L, k, M::HT + 8, hxy);
            stw(L, k, M::HT + 9, hyy);
        }
    }
}

// entry (r, c), r >= c, of the barrier Hessian given its stored pieces
DEV double ht_entry(const double *Ht, int r, int c)
{
    if (r == c) {
        if (r == 2) return Ht[2] + Ht[7];
        if (r == 3) return Ht[3] + Ht[9];
        return Ht[r];
    }
    if (r == 3 && c == 2) return Ht[8];
    return 0.0;
}

template <int NH>
DEV void riccati_factor(const Lane &L)
{
    typedef Map<NH> M;
    const int N = L.N;
    double Pn[NX][NX];
    {
        double Ht[10];
#pragma unroll
        for (int j = 0; j < 10; ++j) Ht[j] = ldw(L, N, M::HT + j);
#pragma unroll
        for (int i = 0; i < NX; ++i)
#pragma unroll
            for (int j = 0; j <= i; ++j) {
                double e = ht_entry(Ht, NU + i, NU + j);
                Pn[i][j] = e;
                Pn[j][i] = e;
            }
#pragma unroll
        for (int i = 0; i < NX; ++i)
#pragma unroll
            for (int j = 0; j < NX; ++j) stw(L, N, M::P + i * NX + j, Pn[i][j]);
    }
    for (int k = N - 1; k >= 0; --k) {
        double A[NX][NX], Bm[NX][NU], Ht[10];
        load_AB<NH>(L, k, A, Bm);
#pragma unroll
        for (int j = 0; j < 10; ++j) Ht[j] = ldw(L, k, M::HT + j);
        double PBA[NX][NZ], Mm[NZ][NZ];
#pragma unroll
        for (int i = 0; i < NX; ++i)
#pragma unroll
            for (int c = 0; c < NZ; ++c) {
                double acc = Pn[i][0] * ((c < NU) ? Bm[0][c] : A[0][c - NU]);
#pragma unroll
                for (int j = 1; j < NX; ++j)
                    acc = fma(Pn[i][j], (c < NU) ? Bm[j][c] : A[j][c - NU], acc);
                PBA[i][c] = acc;
            }
#pragma unroll
        for (int r = 0; r < NZ; ++r)
#pragma unroll
            for (int c = 0; c <= r; ++c) {
                double acc = ht_entry(Ht, r, c);
#pragma unroll
                for (int i = 0; i < NX; ++i)
                    acc = fma((r < NU) ? Bm[i][r] : A[i][r - NU], PBA[i][c], acc);
                Mm[r][c] = acc;
            }
        double d0 = Mm[0][0];
        double l10 = Mm[1][0] / d0;
        double d1 = fma(-l10, Mm[1][0], Mm[1][1]);
        stw(L, k, M::LDL + 0, l10);
        stw(L, k, M::LDL + 1, d0);
        stw(L, k, M::LDL + 2, d1);
        double K[NU][NX];
#pragma unroll
        for (int i = 0; i < NX; ++i) {
            double y0 = Mm[NU + i][0];
            double y1 = fma(-l10, y0, Mm[NU + i][1]);
            double z0 = y0 / d0, z1 = y1 / d1;
            double s1 = z1;
            double s0 = fma(-l10, s1, z0);
            K[0][i] = -s0;
            K[1][i] = -s1;
            stw(L, k, M::K + i, K[0][i]);
            stw(L, k, M::K + NX + i, K[1][i]);
        }
#pragma unroll
        for (int i = 0; i < NX; ++i)
#pragma unroll
            for (int j = 0; j <= i; ++j) {
                double e = fma(Mm[NU + i][0], K[0][j], Mm[NU + i][NU + j]);
                e = fma(Mm[NU + i][1], K[1][j], e);
                Pn[i][j] = e;
                Pn[j][i] = e;
            }
#pragma unroll
        for (int i = 0; i < NX; ++i)
#pragma unroll
            for (int j = 0; j < NX; ++j) stw(L, k, M::P + i * NX + j, Pn[i][j]);
    }
}

template <int NH>
DEV void riccati_solve(const Lane &L)
{
    typedef Map<NH> M;
    const int N = L.N;
    double pn[NX];
#pragma unroll
    for (int i = 0; i < NX; ++i) {
        pn[i] = ldw(L, N, M::GT + NU + i);
        stw(L, N, M::PV + i, pn[i]);
    }
    for (int k = N - 1; k >= 0; --k) {
        double A[NX][NX], Bm[NX][NU], Pn[NX][NX], rb[NX], gt[NZ], w[NX], m[NZ];
        load_AB<NH>(L, k, A, Bm);
#pragma unroll
        for (int i = 0; i < NX; ++i) {
            rb[i] = ldw(L, k, M::RB + i);
#pragma unroll
            for (int j = 0; j < NX; ++j) Pn[i][j] = ldw(L, k + 1, M::P + i * NX + j);
        }
#pragma unroll
        for (int j = 0; j < NZ; ++j) gt[j] = ldw(L, k, M::GT + j);
#pragma unroll
        for (int i = 0; i < NX; ++i) {
            double acc = pn[i];
#pragma unroll
            for (int j = 0; j < NX; ++j) acc = fma(Pn[i][j], rb[j], acc);
            w[i] = acc;
        }
#pragma unroll
        for (int c = 0; c < NZ; ++c) {
            double acc = gt[c];
#pragma unroll
            for (int i = 0; i < NX; ++i) acc = fma((c < NU) ? Bm[i][c] : A[i][c - NU], w[i], acc);
            m[c] = acc;
        }
        double l10 = ldw(L, k, M::LDL + 0), d0 = ldw(L, k, M::LDL + 1), d1 = ldw(L, k, M::LDL + 2);
        double y0 = m[0];
        double y1 = fma(-l10, y0, m[1]);
        double z0 = y0 / d0, z1 = y1 / d1;
        double s1 = z1;
        double s0 = fma(-l10, s1, z0);
        stw(L, k, M::KV + 0, -s0);
        stw(L, k, M::KV + 1, -s1);
#pragma unroll
        for (int i = 0; i < NX; ++i) {
            double k0 = ldw(L, k, M::K + i), k1 = ldw(L, k, M::K + NX + i);
            pn[i] = fma(k1, m[1], fma(k0, m[0], m[NU + i]));
            stw(L, k, M::PV + i, pn[i]);
        }
    }
    double dx[NX];
#pragma unroll
    for (int i = 0; i < NX; ++i) {
        dx[i] = 0.0;
        stw(L, 0, M::DV + NU + i, 0.0);
    }
    for (int k = 0; k < N; ++k) {
        double A[NX][NX], Bm[NX][NU], du[NU], dxn[NX];
        load_AB<NH>(L, k, A, Bm);
#pragma unroll
        for (int c = 0; c < NU; ++c) {
            double acc = ldw(L, k, M::KV + c);
#pragma unroll
            for (int i = 0; i < NX; ++i) acc = fma(ldw(L, k, M::K + c * NX + i), dx[i], acc);
            du[c] = acc;
            stw(L, k, M::DV + c, acc);
        }
#pragma unroll
        for (int i = 0; i < NX; ++i) {
            double acc = ldw(L, k, M::RB + i);
#pragma unroll
            for (int j = 0; j < NX; ++j) acc = fma(A[i][j], dx[j], acc);
#pragma unroll
            for (int c = 0; c < NU; ++c) acc = fma(Bm[i][c], du[c], acc);
            dxn[i] = acc;
            stw(L, k + 1, M::DV + NU + i, acc);
        }
#pragma unroll
        for (int i = 0; i < NX; ++i) {
            double acc = ldw(L, k + 1, M::PV + i);
#pragma unroll
            for (int j = 0; j < NX; ++j) acc = fma(ldw(L, k + 1, M::P + i * NX + j), dxn[j], acc);
            stw(L, k + 1, M::DPI + i, acc);
        }
#pragma unroll
        for (int i = 0; i < NX; ++i) dx[i] = dxn[i];
    }
    stw(L, N, M::DV + 0, 0.0);
    stw(L, N, M::DV + 1, 0.0);
}

// one pass over all row sides: either the longest feasible step and the sums
// S1, S2 (update == false) or the update itself with step alpha (update == true)
template <int NH, bool UPDATE>
DEV double qp_rows_pass(const Lane &L, const mpcx_opts &o, int mode, double sigma_mu,
                        double alpha_in, double &S1, double &S2)
{
    typedef Map<NH> M;
    const int N = L.N;
    double alpha = 1.0, s1 = 0.0, s2 = 0.0;
    for (int k = 0; k <= N; ++k) {
        double dv[NZ], dva[NZ];
#pragma unroll
        for (int j = 0; j < NZ; ++j) {
            dv[j] = ldw(L, k, M::DV + j);
            dva[j] = (mode == 1) ? ldw(L, k, M::DVA + j) : 0.0;
        }
        int r0, r1;
        row_range(L, k, r0, r1);
#pragma unroll
        for (int r = 0; r < M::R; ++r) {
            if (r < r0 || r >= r1) continue;
            double ll = ldw(L, k, M::LAML + r), lu = ldw(L, k, M::LAMU + r);
            double tl = ldw(L, k, M::TL + r), tu = ldw(L, k, M::TU + r);
            double rdl = ldw(L, k, M::RDL + r), rdu = ldw(L, k, M::RDU + r);
            double til = 1.0 / tl, tiu = 1.0 / tu;
            double jx = 0.0, jy = 0.0;
            if (r >= ROW_G0) {
                jx = ldw(L, k, M::JX + r - ROW_G0);
                jy = ldw(L, k, M::JY + r - ROW_G0);
            }
            double jda = 0.0;
            if (mode == 1) jda = (r < ROW_G0) ? dva[r] : fma(jy, dva[3], jx * dva[2]);
            double rml, rmu;
            row_rm(mode, sigma_mu, jda, ll, lu, tl, tu, til, tiu, rdl, rdu, rml, rmu);
            double jd = (r < ROW_G0) ? dv[r] : fma(jy, dv[3], jx * dv[2]);
            double dtl = jd - rdl;
            double dtu = -jd - rdu;
            double dll = -(fma(ll, dtl, rml) * til);
            double dlu = -(fma(lu, dtu, rmu) * tiu);
            if (!UPDATE) {
                if (dll < 0.0) { double c = -ll / dll; if (c < alpha) alpha = c; }
                if (dtl < 0.0) { double c = -tl / dtl; if (c < alpha) alpha = c; }
                if (dlu < 0.0) { double c = -lu / dlu; if (c < alpha) alpha = c; }
                if (dtu < 0.0) { double c = -tu / dtu; if (c < alpha) alpha = c; }
                s1 = fma(ll, dtl, s1);
                s1 = fma(tl, dll, s1);
                s1 = fma(lu, dtu, s1);
                s1 = fma(tu, dlu, s1);
                s2 = fma(dtl, dll, s2);
                s2 = fma(dtu, dlu, s2);
            } else {
                double v;
                v = fma(alpha_in, dll, ll); stw(L, k, M::LAML + r, (v < o.lam_min) ? o.lam_min : v);
                v = fma(alpha_in, dlu, lu); stw(L, k, M::LAMU + r, (v < o.lam_min) ? o.lam_min : v);
                v = fma(alpha_in, dtl, tl); stw(L, k, M::TL + r, (v < o.t_min) ? o.t_min : v);
                v = fma(alpha_in, dtu, tu); stw(L, k, M::TU + r, (v < o.t_min) ? o.t_min : v);
            }
        }
        if (UPDATE) {
#pragma unroll
            for (int j = 0; j < NZ; ++j) {
                bool in = (j < NU) ? (k < N) : (k > 0);
                if (in) stw(L, k, M::V + j, fma(alpha_in, dv[j], ldw(L, k, M::V + j)));
            }
            if (k > 0) {
#pragma unroll
                for (int i = 0; i < NX; ++i)
                    stw(L, k, M::QPI + i, fma(alpha_in, ldw(L, k, M::DPI + i), ldw(L, k, M::QPI + i)));
            }
        }
    }
    S1 = s1;
    S2 = s2;
    return alpha;
}

// HPIPM-style status: 0 solved, 1 iteration limit, 2 minimum step, 3 NaN
template <int NH>
DEV int qp_solve(const Lane &L, const mpcx_opts &o, int &iters)
{
    typedef Map<NH> M;
    const int N = L.N;
    // row sides: stage 0 has the 2 input rows, stages 1..N-1 all rows, stage N the 7 state rows
    const int m_total = 2 * (NU + (N - 1) * (9 + L.nobs) + 7);
    QpRes R;
    qp_init<NH>(L, o);
    qp_residuals<NH>(L, o, m_total, R);
    int it = 0, status = 1;
    for (;;) {
        if (!(finite_d(R.g) && finite_d(R.b) && finite_d(R.d) && finite_d(R.m))) { status = 3; break; }
        if (R.g < o.qp_tol_stat && R.b < o.qp_tol_eq && R.d < o.qp_tol_ineq && R.m < o.qp_tol_comp) { status = 0; break; }
        if (it >= o.max_qp_iter) { status = 1; break; }
        qp_build_rhs<NH>(L, o, 0, 0.0);
        riccati_factor<NH>(L);
        riccati_solve<NH>(L);
        double S1, S2, d1, d2;
        double a_aff = qp_rows_pass<NH, false>(L, o, 0, 0.0, 0.0, S1, S2);
        double S0 = R.mu * (double)m_total;
        double mu_aff = fma(a_aff, fma(a_aff, S2, S1), S0) / (double)m_total;
        if (mu_aff < 0.0) mu_aff = 0.0;
        double ratio = mu_aff / R.mu;
        double sigma = ratio * ratio * ratio;
        double sigma_mu = sigma * R.mu;
        for (int k = 0; k <= N; ++k) {
#pragma unroll
            for (int j = 0; j < NZ; ++j) stw(L, k, M::DVA + j, ldw(L, k, M::DV + j));
        }
        int mode = 1;
        qp_build_rhs<NH>(L, o, 1, sigma_mu);
        riccati_solve<NH>(L);
        double alpha = qp_rows_pass<NH, false>(L, o, 1, sigma_mu, 0.0, d1, d2);
        if (o.cond_pred_corr && alpha < a_aff) {
            mode = 2;
            qp_build_rhs<NH>(L, o, 2, sigma_mu);
            riccati_solve<NH>(L);
            alpha = qp_rows_pass<NH, false>(L, o, 2, sigma_mu, 0.0, d1, d2);
        }
        if (!(alpha > o.alpha_min)) { status = finite_d(alpha) ? 2 : 3; ++it; break; }
        alpha = alpha * fma(alpha, 0.9999999, (1.0 - alpha) * 0.99);
        qp_rows_pass<NH, true>(L, o, mode, sigma_mu, alpha, d1, d2);
        ++it;
        qp_residuals<NH>(L, o, m_total, R);
    }
    iters = it;
    return status;
}

template <int NH>
DEV double nlp_cost(const Lane &L)
{
    typedef Map<NH> M;
    const int N = L.N;
    const double h = ldp(L, MPCX_F_H);
    double c = 0.0;
    for (int k = 0; k < N; ++k) {
#pragma unroll
        for (int j = 0; j < NZ; ++j) {
            double z = (j < NU) ? ldw(L, k, M::U + j)
                                : (ldw(L, k, M::X + j - NU) - ldp(L, MPCX_F_GOAL + j - NU));
            double hw = (0.5 * h) * ldp(L, MPCX_F_W + j);
            c = fma(hw * z, z, c);
        }
    }
#pragma unroll
    for (int i = 0; i < NX; ++i) {
        double z = ldw(L, N, M::X + i) - ldp(L, MPCX_F_GOAL + i);
        c = fma((0.5 * ldp(L, MPCX_F_WE + i)) * z, z, c);
    }
    return c;
}

// ---------------------------------------------------------------------------
// the fused solve: one lane = one problem, from initial guess to result
// ---------------------------------------------------------------------------
template <int NH>
__global__ void __launch_bounds__(64)
mpcx_sqp_kernel(mpcx_batch in, mpcx_opts o, mpcx_result out, double *workspace, int n_groups)
{
    typedef Map<NH> M;
    const int lane = threadIdx.x;
    const int N = in.N;
    const size_t B = (size_t)in.B;
    const size_t wave_doubles = (size_t)(N + 1) * M::F * 64;
    double *wbase = workspace + (size_t)blockIdx.x * wave_doubles + lane;

    for (int grp = blockIdx.x; grp < n_groups; grp += gridDim.x) {
        const size_t b = (size_t)grp * 64 + lane;
        if (b >= B) continue;          // no barrier is used anywhere below
        Lane L;
        L.w = wbase;
        L.pd = in.pdata + b;
        L.B = B;
        L.N = N;
        L.F = M::F;
        L.nobs = in.n_obs ? in.n_obs[b] : 0;
        if (L.nobs < 0 || L.nobs > NH) {   // host validates; keep memory safe regardless
            out.status[b] = MPCX_NAN;
            continue;
        }
        // initial iterate: x from the guess, u = 0, all multipliers 0
        for (int k = 0; k <= N; ++k) {
#pragma unroll
            for (int i = 0; i < NX; ++i) {
                stw(L, k, M::X + i, in.xguess[((size_t)k * NX + i) * B + b]);
                stw(L, k, M::PI + i, 0.0);
            }
            stw(L, k, M::U + 0, 0.0);
            stw(L, k, M::U + 1, 0.0);
#pragma unroll
            for (int r = 0; r < M::R; ++r) {
                stw(L, k, M::NLL + r, 0.0);
                stw(L, k, M::NLU + r, 0.0);
            }
        }
        int status = MPCX_MAXITER, it = 0, qp_total = 0;
        double res[4] = {0.0, 0.0, 0.0, 0.0};
        for (;;) {
            linearize<NH>(L, o);
            nlp_residuals<NH>(L, o, res);
            if (!(finite_d(res[0]) && finite_d(res[1]) && finite_d(res[2]) && finite_d(res[3]))) {
                status = MPCX_NAN;
                break;
            }
            if (!o.rti && res[0] < o.tol_stat && res[1] < o.tol_eq && res[2] < o.tol_ineq &&
                res[3] < o.tol_comp) {
                status = MPCX_SUCCESS;
                break;
            }
            if (it >= o.max_sqp_iter) { status = MPCX_MAXITER; break; }
            int qit = 0;
            int qs = qp_solve<NH>(L, o, qit);
            qp_total += qit;
            ++it;
            if (qs != 0 && qs != 1) { status = MPCX_QP_FAILURE; break; }
            for (int k = 0; k <= N; ++k) {
#pragma unroll
                for (int i = 0; i < NX; ++i) {
                    stw(L, k, M::X + i, ldw(L, k, M::X + i) + ldw(L, k, M::V + NU + i));
                    if (k > 0) stw(L, k, M::PI + i, ldw(L, k, M::QPI + i));
                }
                if (k < N) {
#pragma unroll
                    for (int c = 0; c < NU; ++c)
                        stw(L, k, M::U + c, ldw(L, k, M::U + c) + ldw(L, k, M::V + c));
                }
                int r0, r1;
                row_range(L, k, r0, r1);
#pragma unroll
                for (int r = 0; r < M::R; ++r) {
                    if (r < r0 || r >= r1) continue;
                    stw(L, k, M::NLL + r, ldw(L, k, M::LAML + r));
                    stw(L, k, M::NLU + r, ldw(L, k, M::LAMU + r));
                }
            }
            if (o.rti) { status = MPCX_SUCCESS; break; }
        }
        // results
        for (int k = 0; k <= N; ++k) {
#pragma unroll
            for (int i = 0; i < NX; ++i)
                out.X[(b * (size_t)(N + 1) + k) * NX + i] = ldw(L, k, M::X + i);
            if (k < N) {
                out.U[(b * (size_t)N + k) * NU + 0] = ldw(L, k, M::U + 0);
                out.U[(b * (size_t)N + k) * NU + 1] = ldw(L, k, M::U + 1);
            }
        }
        out.status[b] = status;
        out.sqp_iter[b] = it;
        out.qp_iter[b] = qp_total;
        out.cost[b] = nlp_cost<NH>(L);
#pragma unroll
        for (int i = 0; i < 4; ++i) out.res[(size_t)i * B + b] = res[i];
    }
}

// sin/cos and one RK4 step, for the arithmetic parity tests
__global__ void mpcx_probe_kernel(const double *xin, int n, double *s, double *c)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) mx_sincos(xin[i], s[i], c[i]);
}

}  // namespace

// ---------------------------------------------------------------------------
// host-callable launchers (declared in mpcx_kernels.h)
// ---------------------------------------------------------------------------
size_t mpcx_stage_doubles(int nh)
{
    switch (nh) {
    case 0: return Map<0>::F;
    case 1: return Map<1>::F;
    case 2: return Map<2>::F;
    case 3: return Map<3>::F;
    default: return Map<4>::F;
    }
}

hipError_t mpcx_launch_sqp(int nh, const mpcx_batch &in, const mpcx_opts &o, const mpcx_result &out,
                           double *workspace, int n_waves, hipStream_t stream)
{
    const int n_groups = (in.B + 63) / 64;
    dim3 grid(n_waves), block(64);
    switch (nh) {
    case 0: hipLaunchKernelGGL(mpcx_sqp_kernel<0>, grid, block, 0, stream, in, o, out, workspace, n_groups); break;
    case 1: hipLaunchKernelGGL(mpcx_sqp_kernel<1>, grid, block, 0, stream, in, o, out, workspace, n_groups); break;
    case 2: hipLaunchKernelGGL(mpcx_sqp_kernel<2>, grid, block, 0, stream, in, o, out, workspace, n_groups); break;
    case 3: hipLaunchKernelGGL(mpcx_sqp_kernel<3>, grid, block, 0, stream, in, o, out, workspace, n_groups); break;
    default: hipLaunchKernelGGL(mpcx_sqp_kernel<4>, grid, block, 0, stream, in, o, out, workspace, n_groups); break;
    }
    return hipGetLastError();
}

hipError_t mpcx_launch_probe(const double *xin, int n, double *s, double *c, hipStream_t stream)
{
    hipLaunchKernelGGL(mpcx_probe_kernel, dim3((n + 255) / 256), dim3(256), 0, stream, xin, n, s, c);
    return hipGetLastError();
}
