/*
 * mpc_oracle.c — CPU restatement of the reference's kinematic-car OCP solve.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing in the product (mpc_amd/, the HIP
 * library) includes, links or calls this file; it is the checker the GPU
 * path is compared against, and the timed CPU baseline of bench.py.
 *
 * What it restates
 *   The problem is exactly the one reference src/core_solver.py:127-351 poses
 *   (model :139-166, LINEAR_LS cost :172-236, bounds :262-264, corridor
 *   :267-302, state bounds :305-309, initial/terminal conditions :312-329,
 *   obstacle rows :332-351).  The arithmetic that solves it lives, in the
 *   reference, in acados v0.4.x / HPIPM / BLASFEO / CasADi-generated C, which
 *   are third-party, un-vendored, un-pinned (`git clone` of HEAD, reference
 *   README.md:55-65) and absent from this machine.  This file restates their
 *   published algorithm for the options the reference selects
 *   (core_solver.py:242-253): multiple-shooting SQP with fixed full step,
 *   explicit RK4 (1 step / interval) with forward sensitivities, Gauss-Newton
 *   Hessian plus Levenberg-Marquardt 1e-2, and a Mehrotra predictor-corrector
 *   primal-dual interior-point QP solved by stage-wise Riccati recursion
 *   (HPIPM "balance" mode, cold start).  Details recalled from those codes
 *   and not checkable here are named switches in mpcx_opts.
 *
 * PARITY PINNING: the reference holds no test with expected values for this
 *   path.  Its only numerical pins are three pickled result files under
 *   docs/ (three .npy files); numpy's safe loader refuses them (object arrays), so they are
 *   not loaded.  The values the survey recorded from them (first/last rows of
 *   the `line` and `E_to_F` trajectories, active terminal faces, shapes) are
 *   committed as tests/golden/reference_pins.json and this oracle is checked
 *   against those (tests/test_oracle_pins.py).  Beyond those pins parity with
 *   acados is UNPINNED.
 *
 * Determinism contract with the HIP kernels
 *   Every floating-point operation below is written out: `fma()` where a
 *   fused multiply-add is meant, plain * and + elsewhere, compiled with
 *   -ffp-contract=off.  sin/cos come from mx_sincos() below, not libm.  The
 *   HIP kernels perform the same operations in the same order, so results
 *   are comparable bit for bit, not just to a tolerance.
 *
 * The file is valid C and valid C++: built as C++ with -DMPCX_COUNT_FLOPS the
 * scalar type counts operations (FMA = 2, + - * / = 1, sin and cos = 1 each),
 * which is how the flop constants of mpc_amd/flops.py were frozen.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <stdio.h>

#include "../include/mpcx.h"

#define NX MPCX_NX
#define NU MPCX_NU
#define NZ (NX + NU)               /* stage variable [u; x] */
#define MAXR (9 + MPCX_MAX_OBS)    /* rows: 2 input box, 5 state box, 2 corridor, obstacles */
#define ROW_G0 7                   /* first general (non-box) row */

/* ------------------------------------------------------------------ */
/* scalar type                                                          */
/* ------------------------------------------------------------------ */
#ifdef MPCX_COUNT_FLOPS
struct real;
static unsigned long long g_flops[4];
static int g_bucket = 0;
static int g_count_on = 1;
#define CNT(n) do { if (g_count_on) g_flops[g_bucket] += (n); } while (0)
struct real {
    double v;
    real() : v(0) {}
    real(double d) : v(d) {}
};
static inline real operator+(real a, real b) { CNT(1); return real(a.v + b.v); }
static inline real operator-(real a, real b) { CNT(1); return real(a.v - b.v); }
static inline real operator*(real a, real b) { CNT(1); return real(a.v * b.v); }
static inline real operator/(real a, real b) { CNT(1); return real(a.v / b.v); }
static inline real operator-(real a) { return real(-a.v); }
static inline bool operator<(real a, real b) { return a.v < b.v; }
static inline bool operator>(real a, real b) { return a.v > b.v; }
static inline bool operator<=(real a, real b) { return a.v <= b.v; }
static inline bool operator>=(real a, real b) { return a.v >= b.v; }
static inline real rfma(real a, real b, real c) { CNT(2); return real(fma(a.v, b.v, c.v)); }
static inline double R2D(real a) { return a.v; }
#else
typedef double real;
static inline real rfma(real a, real b, real c) { return fma(a, b, c); }
static inline double R2D(real a) { return a; }
#define CNT(n) ((void)0)
#endif

static inline real rabs(real a) { return (a < (real)0.0) ? -a : a; }
static inline int r_isfinite(real a) { double d = R2D(a); return (d - d) == 0.0; }

/* ------------------------------------------------------------------ */
/* sin / cos: same operations on CPU and GPU                            */
/* ------------------------------------------------------------------ */
/* Cody-Waite reduction by pi/2 in two FMA steps, then the classic minimax
 * polynomials on [-pi/4, pi/4] (coefficients of the freely distributed
 * fdlibm k_sin.c / k_cos.c).  Arguments here are headings and steering
 * angles, a few radians at most; beyond 1e6 the result is NaN on purpose. */
static void mx_sincos(real xr, real *s_out, real *c_out)
{
#ifdef MPCX_COUNT_FLOPS
    int saved = g_count_on;
    CNT(2);
    g_count_on = 0;
#endif
    const real TWO_OVER_PI = 0.63661977236758134308;
    const real MAGIC = 6755399441055744.0;              /* 1.5 * 2^52 */
    const real P1 = 1.5707963267948966;                 /* fl(pi/2) */
    const real P2 = 6.123233995736766e-17;              /* fl(pi/2 - P1) */
    const real S1 = -1.66666666666666324348e-01, S2 = 8.33333333332248946124e-03,
               S3 = -1.98412698298579493134e-04, S4 = 2.75573137070700676789e-06,
               S5 = -2.50507602534068634195e-08, S6 = 1.58969099521155010221e-10;
    const real C1 = 4.16666666666666019037e-02, C2 = -1.38888888888741095749e-03,
               C3 = 2.48015872894767294178e-05, C4 = -2.75573143513906633035e-07,
               C5 = 2.08757232129817482790e-09, C6 = -1.13596475577881948265e-11;
    real s, c;
    if (!(rabs(xr) <= (real)1.0e6)) {
        s = c = (real)NAN;
    } else {
        real kf = (xr * TWO_OVER_PI + MAGIC) - MAGIC;
        real r = rfma(-kf, P1, xr);
        r = rfma(-kf, P2, r);
        long long q = ((long long)R2D(kf)) & 3;
        real z = r * r;
        real ps = rfma(z, S6, S5);
        ps = rfma(z, ps, S4);
        ps = rfma(z, ps, S3);
        ps = rfma(z, ps, S2);
        ps = rfma(z, ps, S1);
        real sn = rfma(r * z, ps, r);
        real pc = rfma(z, C6, C5);
        pc = rfma(z, pc, C4);
        pc = rfma(z, pc, C3);
        pc = rfma(z, pc, C2);
        pc = rfma(z, pc, C1);
        real hz = (real)0.5 * z;
        real w = (real)1.0 - hz;
        real cs = w + (((real)1.0 - w) - hz + (z * z) * pc);
        if (q == 0) { s = sn; c = cs; }
        else if (q == 1) { s = cs; c = -sn; }
        else if (q == 2) { s = -sn; c = -cs; }
        else { s = -cs; c = sn; }
    }
    *s_out = s;
    *c_out = c;
#ifdef MPCX_COUNT_FLOPS
    g_count_on = saved;
#endif
}

/* ------------------------------------------------------------------ */
/* problem and work data                                                */
/* ------------------------------------------------------------------ */
typedef struct {
    int N, nobs;
    real h, invL;
    real x0[NX], goal[NX];
    real W[NZ], We[NX];
    real lo_s[MAXR], up_s[MAXR];   /* bounds of rows at stages 0..N-1 */
    real lo_e[MAXR], up_e[MAXR];   /* bounds of rows at stage N        */
    real cg[2][2];
    real obs[MPCX_MAX_OBS][3];
} problem;

typedef struct {
    /* NLP iterate */
    real x[NX], u[NU], pi[NX];
    real nl_l[MAXR], nl_u[MAXR];       /* NLP multipliers of lower / upper sides */
    /* linearisation at the iterate */
    real A[NX][NX], B[NX][NU], b[NX];  /* interval k -> k+1 */
    real g[NZ], Hd[NZ];
    real c[MAXR], jx[MAXR], jy[MAXR];
    int r0, r1;                         /* rows present: r0 <= r < r1 and bit r of rmask set */
    unsigned rmask;
    /* QP iterate */
    real v[NZ], qpi[NX];
    real lam_l[MAXR], lam_u[MAXR], t_l[MAXR], t_u[MAXR];
    /* residuals */
    real rg[NZ], rb[NX], rd_l[MAXR], rd_u[MAXR];
    /* Newton system */
    real Ht[NZ], hxx, hxy, hyy;        /* barrier Hessian: diagonal + (x,y) block */
    real gt[NZ];
    real gd[NZ];                        /* right-hand side of the correction to the affine step */
    real ga[NZ], gb[NZ];                /* its two parts: gd = ga - sigma*mu * gb (corrector), -sigma*mu * gb (centring) */
    real pa[NX];                        /* p of the affine solve (study switch only) */
    real K[NU][NX], kv[NU], l10, d0, d1;   /* d0, d1: RECIPROCALS of the pivots of M_uu */
    real P[NX][NX], p[NX];             /* full symmetric storage */
    real dv[NZ], dpi[NX];
    real dva[NZ];                       /* affine (predictor) step */
} stage;

static inline const real *row_lo(const problem *pb, int k) { return k == pb->N ? pb->lo_e : pb->lo_s; }
static inline const real *row_up(const problem *pb, int k) { return k == pb->N ? pb->up_e : pb->up_s; }

/* J_r * v for row r of stage s (v = [u; x]) */
static inline real row_dot(const stage *s, int r, const real *v)
{
    if (r < ROW_G0) return v[r];
    return rfma(s->jy[r], v[3], s->jx[r] * v[2]);
}

/* ------------------------------------------------------------------ */
/* model: f(x,u) and the non-trivial entries of df/dx                   */
/* (core_solver.py:148-155)                                             */
/* ------------------------------------------------------------------ */
typedef struct { real j02, j03, j12, j13, j23, j24; } jac_t;

static void model_eval(const problem *pb, const real *x, const real *u, real *f, jac_t *J)
{
    real sth, cth, sde, cde;
    mx_sincos(x[2], &sth, &cth);
    mx_sincos(x[4], &sde, &cde);
    real tde = sde / cde;
    real v = x[3];
    f[0] = v * cth;
    f[1] = v * sth;
    f[2] = (v * tde) * pb->invL;
    f[3] = u[0];
    f[4] = u[1];
    J->j02 = -f[1];
    J->j03 = cth;
    J->j12 = f[0];
    J->j13 = sth;
    J->j23 = tde * pb->invL;
    J->j24 = (v * rfma(tde, tde, (real)1.0)) * pb->invL;
}

/* K = J * S + dF/du, S = [Sx | Su] stored as 5 x 7 with columns (x0..x4,u0,u1) */
static void vde_eval(const jac_t *J, real S[NX][NZ], real Kout[NX][NZ])
{
    for (int c = 0; c < NZ; ++c) {
        Kout[0][c] = rfma(J->j03, S[3][c], J->j02 * S[2][c]);
        Kout[1][c] = rfma(J->j13, S[3][c], J->j12 * S[2][c]);
        Kout[2][c] = rfma(J->j24, S[4][c], J->j23 * S[3][c]);
        Kout[3][c] = (real)0.0;
        Kout[4][c] = (real)0.0;
    }
    Kout[3][NX + 0] = (real)1.0;
    Kout[4][NX + 1] = (real)1.0;
}

/* one explicit RK4 step with forward sensitivities (acados ERK, 4 stages, 1 step) */
static void rk4_sens(const problem *pb, const real *x, const real *u,
                     real *xn, real A[NX][NX], real Bm[NX][NU])
{
    const real h = pb->h, hh = (real)0.5 * pb->h, h6 = pb->h / (real)6.0;
    real k1[NX], k2[NX], k3[NX], k4[NX], xs[NX];
    real S0[NX][NZ], Ss[NX][NZ], K1[NX][NZ], K2[NX][NZ], K3[NX][NZ], K4[NX][NZ];
    jac_t J;
    for (int i = 0; i < NX; ++i)
        for (int c = 0; c < NZ; ++c) S0[i][c] = (i == c) ? (real)1.0 : (real)0.0;

    model_eval(pb, x, u, k1, &J);
    vde_eval(&J, S0, K1);
    for (int i = 0; i < NX; ++i) {
        xs[i] = rfma(hh, k1[i], x[i]);
        for (int c = 0; c < NZ; ++c) Ss[i][c] = rfma(hh, K1[i][c], S0[i][c]);
    }
    model_eval(pb, xs, u, k2, &J);
    vde_eval(&J, Ss, K2);
    for (int i = 0; i < NX; ++i) {
        xs[i] = rfma(hh, k2[i], x[i]);
        for (int c = 0; c < NZ; ++c) Ss[i][c] = rfma(hh, K2[i][c], S0[i][c]);
    }
    model_eval(pb, xs, u, k3, &J);
    vde_eval(&J, Ss, K3);
    for (int i = 0; i < NX; ++i) {
        xs[i] = rfma(h, k3[i], x[i]);
        for (int c = 0; c < NZ; ++c) Ss[i][c] = rfma(h, K3[i][c], S0[i][c]);
    }
    model_eval(pb, xs, u, k4, &J);
    vde_eval(&J, Ss, K4);
    /* weighted sum in the order the stages are produced: ((k1 + 2 k2) + 2 k3) + k4 */
    for (int i = 0; i < NX; ++i) {
        real t = rfma((real)2.0, k2[i], k1[i]);
        t = rfma((real)2.0, k3[i], t);
        t = t + k4[i];
        xn[i] = rfma(h6, t, x[i]);
        for (int c = 0; c < NZ; ++c) {
            real ts = rfma((real)2.0, K2[i][c], K1[i][c]);
            ts = rfma((real)2.0, K3[i][c], ts);
            ts = ts + K4[i][c];
            real sv = rfma(h6, ts, S0[i][c]);
            if (c < NX) A[i][c] = sv; else Bm[i][c - NX] = sv;
        }
    }
}

static int g_stage0_rows = 0;      /* MPCX_ORACLE_STAGE0_ROWS, see linearize() */
/* MPCX_ORACLE_COND_BLOCKS (study only): emulates HPIPM's partial condensing to this many blocks
 * (reference core_solver.py:248 selects 5).  In exact arithmetic an interior-point method on the
 * condensed QP takes the steps of the stage-wise method IF the latter starts from states that satisfy
 * the dynamics inside every block, and its stopping test sees the multipliers of the eliminated
 * dynamics rows at their optimal values.  So: (i) the cold start propagates the states through each
 * block from the block's first state (states inside a block are no optimisation variables there and
 * are not moved into their boxes: those rows are general constraints of the condensed QP), (ii) the
 * multipliers of the dynamics rows inside a block are recomputed in every residual evaluation so that
 * the stationarity of the eliminated states holds exactly. */
static int g_cond_blocks = 0;
static inline int block_start(const problem *pb, int k)
{
    if (g_cond_blocks <= 0) return 1;
    const int N = pb->N, nb = g_cond_blocks;
    if (k == 0 || k >= N) return 1;
    /* HPIPM's partition: N / nb stages per block, the first N % nb blocks one longer */
    const int q = N / nb, rmd = N % nb;
    int start = 0;
    for (int j = 0; j < nb; ++j) {
        if (k == start) return 1;
        start += q + (j < rmd ? 1 : 0);
        if (start > k) return 0;
    }
    return 1;
}

/* ------------------------------------------------------------------ */
/* linearisation of one SQP iteration                                   */
/* ------------------------------------------------------------------ */
static void linearize(const problem *pb, const mpcx_opts *o, stage *st)
{
    const int N = pb->N;
    for (int k = 0; k <= N; ++k) {
        stage *s = &st[k];
        /* rows present at this stage */
        if (k == 0) { s->r0 = 0; s->r1 = NU; }
        else if (k < N) { s->r0 = 0; s->r1 = 9 + pb->nobs; }
        else { s->r0 = NU; s->r1 = 9; }
        s->rmask = ~0u;
        /* study switch (never set by the product or the parity tests): also impose, at stage 0,
         * the corridor rows (1) or the corridor and obstacle rows (2).  With x0 fixed they are
         * constant rows: they enter the complementarity average, the step length and the
         * residual norms only.  The state box stays off stage 0 (x0 is an equality there). */
        if (k == 0 && g_stage0_rows > 0) {
            s->r1 = (g_stage0_rows >= 2) ? 9 + pb->nobs : 9;
            s->rmask = ~0u & ~(((1u << NX) - 1u) << NU);
        }
        /* dynamics */
        if (k < N) {
            real xn[NX];
            rk4_sens(pb, s->x, s->u, xn, s->A, s->B);
            for (int i = 0; i < NX; ++i) s->b[i] = xn[i] - st[k + 1].x[i];
        }
        /* cost: Gauss-Newton Hessian (diagonal) + Levenberg-Marquardt, gradient */
        if (k < N) {
            real lmk = o->lm_scale_by_h ? pb->h * (real)o->lm : (real)o->lm;
            for (int j = 0; j < NZ; ++j) {
                real hw = pb->h * pb->W[j];
                real z = (j < NU) ? s->u[j] : (s->x[j - NU] - pb->goal[j - NU]);
                s->Hd[j] = hw + lmk;
                s->g[j] = hw * z;
            }
        } else {
            s->Hd[0] = s->Hd[1] = (real)0.0;
            s->g[0] = s->g[1] = (real)0.0;
            for (int i = 0; i < NX; ++i) {
                s->Hd[NU + i] = pb->We[i] + (real)o->lm;
                s->g[NU + i] = pb->We[i] * (s->x[i] - pb->goal[i]);
            }
        }
        /* constraint rows: value and Jacobian */
        for (int r = s->r0; r < s->r1; ++r) {
            if (!((s->rmask >> r) & 1u)) continue;
            if (r < NU) s->c[r] = s->u[r];
            else if (r < ROW_G0) s->c[r] = s->x[r - NU];
            else if (r < 9) {
                s->jx[r] = pb->cg[r - ROW_G0][0];
                s->jy[r] = pb->cg[r - ROW_G0][1];
                s->c[r] = rfma(s->jy[r], s->x[1], s->jx[r] * s->x[0]);
            } else {
                const real *ob = pb->obs[r - 9];
                real dx = s->x[0] - ob[0], dy = s->x[1] - ob[1];
                s->c[r] = rfma(dy, dy, dx * dx) - ob[2];
                s->jx[r] = (real)2.0 * dx;
                s->jy[r] = (real)2.0 * dy;
            }
        }
    }
}

static int g_debug = 0;

static inline real rmax(real a, real b) { return (b > a) ? b : a; }

/* value of row r at (x, u), as linearize() forms it */
static inline real row_value(const problem *pb, int r, const real *x, const real *u)
{
    if (r < NU) return u[r];
    if (r < ROW_G0) return x[r - NU];
    if (r < 9) return rfma(pb->cg[r - ROW_G0][1], x[1], pb->cg[r - ROW_G0][0] * x[0]);
    const real *ob = pb->obs[r - 9];
    real dx = x[0] - ob[0], dy = x[1] - ob[1];
    return rfma(dy, dy, dx * dx) - ob[2];
}

/* NLP residuals at (x,u,pi,lam) with the current linearisation.
 * res[0] stationarity, [1] dynamics gap, [2] inequality violation, [3] complementarity */
static void nlp_residuals(const problem *pb, const stage *st, real res[4])
{
    const int N = pb->N;
    real rs = 0.0, re = 0.0, ri = 0.0, rc = 0.0;
    for (int k = 0; k <= N; ++k) {
        const stage *s = &st[k];
        const real *lo = row_lo(pb, k), *up = row_up(pb, k);
        real gl[NZ];
        for (int j = 0; j < NZ; ++j) gl[j] = s->g[j];
        if (k < N) {
            const real *pn = st[k + 1].pi;
            for (int c = 0; c < NU; ++c)
                for (int i = 0; i < NX; ++i) gl[c] = rfma(s->B[i][c], pn[i], gl[c]);
            for (int c = 0; c < NX; ++c)
                for (int i = 0; i < NX; ++i) gl[NU + c] = rfma(s->A[i][c], pn[i], gl[NU + c]);
        }
        if (k > 0)
            for (int i = 0; i < NX; ++i) gl[NU + i] = gl[NU + i] - s->pi[i];
        for (int r = s->r0; r < s->r1; ++r) {
            if (!((s->rmask >> r) & 1u)) continue;
            real dl = s->nl_u[r] - s->nl_l[r];
            if (r < ROW_G0) gl[r] = gl[r] + dl;
            else {
                gl[2] = rfma(s->jx[r], dl, gl[2]);
                gl[3] = rfma(s->jy[r], dl, gl[3]);
            }
            real sl = s->c[r] - lo[r], su = up[r] - s->c[r];
            ri = rmax(ri, -sl);
            ri = rmax(ri, -su);
            rc = rmax(rc, rabs(s->nl_l[r] * sl));
            rc = rmax(rc, rabs(s->nl_u[r] * su));
        }
        int j0 = (k == 0) ? 0 : 0, j1 = (k == 0) ? NU : NZ;
        if (k == N) j0 = NU;
        for (int j = j0; j < j1; ++j) rs = rmax(rs, rabs(gl[j]));
        if (k < N)
            for (int i = 0; i < NX; ++i) re = rmax(re, rabs(s->b[i]));
        if (k == 0)
            for (int i = 0; i < NX; ++i) ri = rmax(ri, rabs(pb->x0[i] - s->x[i]));
    }
    res[0] = rs; res[1] = re; res[2] = ri; res[3] = rc;
}

/* ------------------------------------------------------------------ */
/* interior-point QP                                                    */
/* ------------------------------------------------------------------ */
typedef struct { real g, b, d, m, mu; } qp_res;

/* d_l = lo - c, d_u = up - c are the QP's bounds on J*v */
static void qp_residuals(const problem *pb, stage *st, int m_total, qp_res *R)
{
    const int N = pb->N;
    real ng = 0.0, nb = 0.0, nd = 0.0, nm = 0.0, sum = 0.0;
    /* stages are visited last to first (the order in which a backward Riccati
     * sweep meets them), which fixes the summation order of `sum` */
    for (int k = N; k >= 0; --k) {
        stage *s = &st[k];
        const real *lo = row_lo(pb, k), *up = row_up(pb, k);
        for (int j = 0; j < NZ; ++j) s->rg[j] = rfma(s->Hd[j], s->v[j], s->g[j]);
        if (k < N) {
            const real *pn = st[k + 1].qpi;
            for (int c = 0; c < NU; ++c)
                for (int i = 0; i < NX; ++i) s->rg[c] = rfma(s->B[i][c], pn[i], s->rg[c]);
            for (int c = 0; c < NX; ++c)
                for (int i = 0; i < NX; ++i) s->rg[NU + c] = rfma(s->A[i][c], pn[i], s->rg[NU + c]);
        }
        const int eliminate = (k > 0) && !block_start(pb, k);      /* study switch, see g_cond_blocks */
        if (k > 0 && !eliminate)
            for (int i = 0; i < NX; ++i) s->rg[NU + i] = s->rg[NU + i] - s->qpi[i];
        real ssum = 0.0;     /* complementarity products of this stage, then added to the total */
        for (int r = s->r0; r < s->r1; ++r) {
            if (!((s->rmask >> r) & 1u)) continue;
            real jv = row_dot(s, r, s->v);
            real dl = s->lam_u[r] - s->lam_l[r];
            if (r < ROW_G0) s->rg[r] = s->rg[r] + dl;
            else {
                s->rg[2] = rfma(s->jx[r], dl, s->rg[2]);
                s->rg[3] = rfma(s->jy[r], dl, s->rg[3]);
            }
            s->rd_l[r] = ((lo[r] - s->c[r]) - jv) + s->t_l[r];
            s->rd_u[r] = (jv - (up[r] - s->c[r])) + s->t_u[r];
            nd = rmax(nd, rabs(s->rd_l[r]));
            nd = rmax(nd, rabs(s->rd_u[r]));
            real ml = s->lam_l[r] * s->t_l[r], mu_ = s->lam_u[r] * s->t_u[r];
            nm = rmax(nm, rabs(ml));
            nm = rmax(nm, rabs(mu_));
            ssum = ssum + ml;
            ssum = ssum + mu_;
        }
        sum = sum + ssum;
        if (eliminate)
            for (int i = 0; i < NX; ++i) { s->qpi[i] = s->rg[NU + i]; s->rg[NU + i] = (real)0.0; }
        int j0 = (k == N) ? NU : 0, j1 = (k == 0) ? NU : NZ;
        for (int j = j0; j < j1; ++j) ng = rmax(ng, rabs(s->rg[j]));
        if (k < N) {
            const real *vn = &st[k + 1].v[NU];
            for (int i = 0; i < NX; ++i) {
                real acc = s->b[i];
                for (int j = 0; j < NX; ++j) acc = rfma(s->A[i][j], s->v[NU + j], acc);
                for (int c = 0; c < NU; ++c) acc = rfma(s->B[i][c], s->v[c], acc);
                s->rb[i] = acc - vn[i];
                nb = rmax(nb, rabs(s->rb[i]));
            }
        }
    }
    R->g = ng; R->b = nb; R->d = nd; R->m = nm;
    R->mu = sum / (real)(double)m_total;
}

/* cold start of the interior point (HPIPM init_var, warm_start = 0) */
static void qp_init(const problem *pb, const mpcx_opts *o, stage *st)
{
    const int N = pb->N;
    const real thr0 = o->thr0, mu0 = o->mu0;
    for (int k = 0; k <= N; ++k) {
        stage *s = &st[k];
        const real *lo = row_lo(pb, k), *up = row_up(pb, k);
        for (int j = 0; j < NZ; ++j) s->v[j] = (real)0.0;
        for (int i = 0; i < NX; ++i) s->qpi[i] = (real)0.0;
        if (k == 0)
            for (int i = 0; i < NX; ++i) s->v[NU + i] = pb->x0[i] - s->x[i];
        const int inside = !block_start(pb, k);
        if (inside) {      /* study switch: the state follows from the stage before */
            const stage *p = &st[k - 1];
            for (int i = 0; i < NX; ++i) {
                real acc = p->b[i];
                for (int j = 0; j < NX; ++j) acc = rfma(p->A[i][j], p->v[NU + j], acc);
                for (int c = 0; c < NU; ++c) acc = rfma(p->B[i][c], p->v[c], acc);
                s->v[NU + i] = acc;
            }
        }
        for (int r = s->r0; r < s->r1; ++r) {
            if (!((s->rmask >> r) & 1u)) continue;
            real dl = lo[r] - s->c[r], du = up[r] - s->c[r];
            real tl, tu;
            if (r < ROW_G0 && !(inside && r >= NU)) {
                tl = s->v[r] - dl;
                tu = du - s->v[r];
                if (tl < thr0) {
                    if (tu < thr0) {
                        s->v[r] = (real)0.5 * (dl + du);
                        tl = thr0;
                        tu = thr0;
                    } else {
                        tl = thr0;
                        s->v[r] = dl + thr0;
                    }
                } else if (tu < thr0) {
                    tu = thr0;
                    s->v[r] = du - thr0;
                }
            } else {
                real jv = row_dot(s, r, s->v);
                tl = jv - dl;
                tu = du - jv;
                if (tl < thr0) tl = thr0;
                if (tu < thr0) tu = thr0;
            }
            s->t_l[r] = tl;
            s->t_u[r] = tu;
            s->lam_l[r] = mu0 / tl;
            s->lam_u[r] = mu0 / tu;
        }
    }
}

/* barrier Hessian (mode 0) and right-hand side gt for the three kinds of step:
 *   mode 0: affine predictor,        rm = lam*t
 *   mode 1: Mehrotra corrector,      rm = lam*t + dt_a*dlam_a - sigma*mu
 *   mode 2: centering only,          rm = lam*t - sigma*mu                */
static void qp_build_rhs(const problem *pb, stage *st, int mode, real sigma_mu)
{
    const int N = pb->N;
    for (int k = 0; k <= N; ++k) {
        stage *s = &st[k];
        if (mode == 0) {
            for (int j = 0; j < NZ; ++j) s->Ht[j] = s->Hd[j];
            s->hxx = s->hxy = s->hyy = (real)0.0;
        }
        for (int j = 0; j < NZ; ++j) {
            s->gt[j] = s->rg[j];
            /* what modes 1 and 2 add to the affine right-hand side, from the two sums the affine
             * step-length pass left per variable (qp_step_length) */
            if (mode != 0) s->gd[j] = (mode == 1) ? rfma(-sigma_mu, s->gb[j], s->ga[j]) : (-sigma_mu) * s->gb[j];
        }
        for (int r = s->r0; r < s->r1; ++r) {
            if (!((s->rmask >> r) & 1u)) continue;
            real til = (real)1.0 / s->t_l[r], tiu = (real)1.0 / s->t_u[r];
            real rml = s->lam_l[r] * s->t_l[r], rmu = s->lam_u[r] * s->t_u[r];
            if (mode == 1) {
                real jd = row_dot(s, r, s->dva);
                real dtl = jd - s->rd_l[r];
                real dtu = -jd - s->rd_u[r];
                real dll = -(rfma(s->lam_l[r], dtl, rml) * til);
                real dlu = -(rfma(s->lam_u[r], dtu, rmu) * tiu);
                rml = rfma(dtl, dll, rml) - sigma_mu;
                rmu = rfma(dtu, dlu, rmu) - sigma_mu;
            } else if (mode == 2) {
                rml = rml - sigma_mu;
                rmu = rmu - sigma_mu;
            }
            real ql = rfma(-s->lam_l[r], s->rd_l[r], rml) * til;
            real qu = rfma(-s->lam_u[r], s->rd_u[r], rmu) * tiu;
            real q = ql - qu;
            if (r < ROW_G0) s->gt[r] = s->gt[r] + q;
            else {
                s->gt[2] = rfma(s->jx[r], q, s->gt[2]);
                s->gt[3] = rfma(s->jy[r], q, s->gt[3]);
            }
            if (mode == 0) {
                real w = s->lam_l[r] * til + s->lam_u[r] * tiu;
                if (r < ROW_G0) s->Ht[r] = s->Ht[r] + w;
                else {
                    real wx = w * s->jx[r], wy = w * s->jy[r];
                    s->hxx = rfma(wx, s->jx[r], s->hxx);
                    s->hxy = rfma(wy, s->jx[r], s->hxy);
                    s->hyy = rfma(wy, s->jy[r], s->hyy);
                }
            }
        }
    }
}

/* entry (r,c), r >= c, of the barrier Hessian of a stage */
static inline real Ht_entry(const stage *s, int r, int c)
{
    if (r == c) {
        if (r == 2) return s->Ht[2] + s->hxx;
        if (r == 3) return s->Ht[3] + s->hyy;
        return s->Ht[r];
    }
    if (r == 3 && c == 2) return s->hxy;
    return (real)0.0;
}

/* backward Riccati factorisation: K, LDL' of M_uu, P for every stage */
static void riccati_factor(const problem *pb, stage *st)
{
    const int N = pb->N;
    stage *sN = &st[N];
    for (int i = 0; i < NX; ++i)
        for (int j = 0; j <= i; ++j) {
            real e = Ht_entry(sN, NU + i, NU + j);
            sN->P[i][j] = e;
            sN->P[j][i] = e;
        }
    for (int k = N - 1; k >= 0; --k) {
        stage *s = &st[k];
        real (*Pn)[NX] = st[k + 1].P;
        real PBA[NX][NZ], M[NZ][NZ];
        /* BA(i, c): c < NU -> B column, else A column */
        for (int i = 0; i < NX; ++i)
            for (int c = 0; c < NZ; ++c) {
                real acc = Pn[i][0] * ((c < NU) ? s->B[0][c] : s->A[0][c - NU]);
                for (int j = 1; j < NX; ++j)
                    acc = rfma(Pn[i][j], (c < NU) ? s->B[j][c] : s->A[j][c - NU], acc);
                PBA[i][c] = acc;
            }
        for (int r = 0; r < NZ; ++r)
            for (int c = 0; c <= r; ++c) {
                real acc = Ht_entry(s, r, c);
                for (int i = 0; i < NX; ++i)
                    acc = rfma((r < NU) ? s->B[i][r] : s->A[i][r - NU], PBA[i][c], acc);
                M[r][c] = acc;
            }
        /* LDL' of M_uu; the two pivots are inverted once and multiplied with from then on */
        s->d0 = (real)1.0 / M[0][0];
        s->l10 = M[1][0] * s->d0;
        s->d1 = (real)1.0 / rfma(-s->l10, M[1][0], M[1][1]);
        for (int i = 0; i < NX; ++i) {
            real y0 = M[NU + i][0];
            real y1 = rfma(-s->l10, y0, M[NU + i][1]);
            real z0 = y0 * s->d0, z1 = y1 * s->d1;
            real s1 = z1;
            real s0 = rfma(-s->l10, s1, z0);
            s->K[0][i] = -s0;
            s->K[1][i] = -s1;
        }
        for (int i = 0; i < NX; ++i)
            for (int j = 0; j <= i; ++j) {
                real e = rfma(M[NU + i][0], s->K[0][j], M[NU + i][NU + j]);
                e = rfma(M[NU + i][1], s->K[1][j], e);
                s->P[i][j] = e;
                s->P[j][i] = e;
            }
    }
}

static int g_dpi_costate = 0;     /* MPCX_ORACLE_DPI_COSTATE, see riccati_solve() */
/* backward + forward substitution with the stored factors; result in dv, dpi */
static void riccati_solve(const problem *pb, stage *st, int delta)
{
    /* delta = 0: the affine step, full right-hand side gt and dynamics residual rb.
     * delta = 1: the corrector (or centring-only) step as the affine step plus a correction: the two
     * systems differ in the right-hand side of the rows only, so the correction solves the same system
     * with right-hand side gd, no dynamics residual and no initial-state defect, and dv = dva + it.
     * (Equal to solving with the full right-hand side in exact arithmetic; this form needs neither the
     * stationarity residual nor P rb nor rb in the second pair of sweeps.) */
    const int N = pb->N;
    for (int i = 0; i < NX; ++i) st[N].p[i] = delta ? st[N].gd[NU + i] : st[N].gt[NU + i];
    for (int k = N - 1; k >= 0; --k) {
        stage *s = &st[k];
        const stage *n = &st[k + 1];
        real w[NX], m[NZ];
        for (int i = 0; i < NX; ++i) {
            if (delta) { w[i] = n->p[i]; continue; }
            real acc = n->P[i][0] * s->rb[0];
            for (int j = 1; j < NX; ++j) acc = rfma(n->P[i][j], s->rb[j], acc);
            w[i] = n->p[i] + acc;
        }
        for (int c = 0; c < NZ; ++c) {
            real acc = delta ? s->gd[c] : s->gt[c];
            for (int i = 0; i < NX; ++i)
                acc = rfma((c < NU) ? s->B[i][c] : s->A[i][c - NU], w[i], acc);
            m[c] = acc;
        }
        real y0 = m[0];
        real y1 = rfma(-s->l10, y0, m[1]);
        real z0 = y0 * s->d0, z1 = y1 * s->d1;
        real s1 = z1;
        real s0 = rfma(-s->l10, s1, z0);
        s->kv[0] = -s0;
        s->kv[1] = -s1;
        for (int i = 0; i < NX; ++i)
            s->p[i] = rfma(s->K[1][i], m[1], rfma(s->K[0][i], m[0], m[NU + i]));
    }
    /* forward: the initial state is fixed, so its step is zero */
    for (int i = 0; i < NX; ++i) st[0].dv[NU + i] = (real)0.0;
    for (int k = 0; k < N; ++k) {
        stage *s = &st[k];
        stage *n = &st[k + 1];
        for (int c = 0; c < NU; ++c) {
            real acc = s->kv[c];
            for (int i = 0; i < NX; ++i) acc = rfma(s->K[c][i], s->dv[NU + i], acc);
            s->dv[c] = acc;
        }
        for (int i = 0; i < NX; ++i) {
            real acc = delta ? (real)0.0 : s->rb[i];
            for (int j = 0; j < NX; ++j) acc = rfma(s->A[i][j], s->dv[NU + j], acc);
            for (int c = 0; c < NU; ++c) acc = rfma(s->B[i][c], s->dv[c], acc);
            n->dv[NU + i] = acc;
        }
    }
    st[N].dv[0] = st[N].dv[1] = (real)0.0;
    if (delta) {
        for (int k = 0; k <= N; ++k)
            for (int j = 0; j < NZ; ++j) st[k].dv[j] = st[k].dva[j] + st[k].dv[j];
    } else {
        for (int k = 0; k <= N; ++k)
            for (int i = 0; i < NX; ++i) st[k].pa[i] = st[k].p[i];
    }
    if (g_dpi_costate) {
        /* study switch (MPCX_ORACLE_DPI_COSTATE=1; never set by the product or the parity tests): the
         * form HPIPM uses, dpi_k = P_k dx_k + p_k */
        for (int k = 1; k <= N; ++k) {
            stage *n = &st[k];
            for (int i = 0; i < NX; ++i) {
                real acc = delta ? n->pa[i] + n->p[i] : n->p[i];
                for (int j = 0; j < NX; ++j) acc = rfma(n->P[i][j], n->dv[NU + j], acc);
                n->dpi[i] = acc;
            }
        }
        return;
    }
    /* step of the dynamics multipliers: the x_k block of the Newton system,
     *   Ht_xx dx_k + A_k' dpi_(k+1) - dpi_k = -gt_x,
     * solved for dpi_k from the last stage backwards (equal to P_k dx_k + p_k in exact
     * arithmetic; this form needs neither P nor p once the forward pass is done) */
    for (int k = N; k >= 1; --k) {
        stage *s = &st[k];
        const real *dx = &s->dv[NU];
        for (int i = 0; i < NX; ++i) {
            real acc = s->gt[NU + i];
            if (i == 0) acc = rfma(s->hxy, dx[1], rfma(Ht_entry(s, 2, 2), dx[0], acc));
            else if (i == 1) acc = rfma(Ht_entry(s, 3, 3), dx[1], rfma(s->hxy, dx[0], acc));
            else acc = rfma(s->Ht[NU + i], dx[i], acc);
            if (k < N)
                for (int r = 0; r < NX; ++r) acc = rfma(s->A[r][i], st[k + 1].dpi[r], acc);
            s->dpi[i] = acc;
        }
    }
}

/* step of the slacks and multipliers implied by dv, longest feasible step,
 * and the sums S1 = sum(lam*dt + t*dlam), S2 = sum(dt*dlam) */
static real qp_step_length(const problem *pb, stage *st, int mode, real sigma_mu,
                           real *S1, real *S2)
{
    const int N = pb->N;
    /* The longest feasible step is the smallest ratio n/d over the sides that
     * move towards zero (n = current value, d = -step > 0).  The running
     * minimum of a stage is kept as the pair (num, den) and candidates are
     * compared by cross multiplication, so a stage costs one division. */
    real alpha = 1.0, s1 = 0.0, s2 = 0.0;
    for (int k = 0; k <= N; ++k) {
        stage *s = &st[k];
        real num = 1.0, den = 1.0;    /* smallest ratio of this stage; the step is the min over stages */
        real s1k = 0.0, s2k = 0.0;    /* sums of this stage, then added to the totals */
        if (mode == 0)
            for (int j = 0; j < NZ; ++j) { s->ga[j] = (real)0.0; s->gb[j] = (real)0.0; }
        for (int r = s->r0; r < s->r1; ++r) {
            if (!((s->rmask >> r) & 1u)) continue;
            real til = (real)1.0 / s->t_l[r], tiu = (real)1.0 / s->t_u[r];
            real rml = s->lam_l[r] * s->t_l[r], rmu = s->lam_u[r] * s->t_u[r];
            if (mode == 1) {
                real jd = row_dot(s, r, s->dva);
                real dtl = jd - s->rd_l[r];
                real dtu = -jd - s->rd_u[r];
                real dll = -(rfma(s->lam_l[r], dtl, rml) * til);
                real dlu = -(rfma(s->lam_u[r], dtu, rmu) * tiu);
                rml = rfma(dtl, dll, rml) - sigma_mu;
                rmu = rfma(dtu, dlu, rmu) - sigma_mu;
            } else if (mode == 2) {
                rml = rml - sigma_mu;
                rmu = rmu - sigma_mu;
            }
            real jd = row_dot(s, r, s->dv);
            real dtl = jd - s->rd_l[r];
            real dtu = -jd - s->rd_u[r];
            real dll = -(rfma(s->lam_l[r], dtl, rml) * til);
            real dlu = -(rfma(s->lam_u[r], dtu, rmu) * tiu);
            if (dll < (real)0.0 && s->lam_l[r] * den < num * (-dll)) { num = s->lam_l[r]; den = -dll; }
            if (dtl < (real)0.0 && s->t_l[r] * den < num * (-dtl)) { num = s->t_l[r]; den = -dtl; }
            if (dlu < (real)0.0 && s->lam_u[r] * den < num * (-dlu)) { num = s->lam_u[r]; den = -dlu; }
            if (dtu < (real)0.0 && s->t_u[r] * den < num * (-dtu)) { num = s->t_u[r]; den = -dtu; }
            s1k = rfma(s->lam_l[r], dtl, s1k);
            s1k = rfma(s->t_l[r], dll, s1k);
            s1k = rfma(s->lam_u[r], dtu, s1k);
            s1k = rfma(s->t_u[r], dlu, s1k);
            s2k = rfma(dtl, dll, s2k);
            s2k = rfma(dtu, dlu, s2k);
            if (mode == 0) {
                /* the rows' part of the corrector's and the centring step's right-hand sides, up to the
                 * factor sigma*mu that is known only after this pass:
                 *   ga = J' ((dt_a dlam_a)_l / t_l - (dt_a dlam_a)_u / t_u),  gb = J' (1 / t_l - 1 / t_u) */
                real ar = (dtl * dll) * til - (dtu * dlu) * tiu;
                real br = til - tiu;
                if (r < ROW_G0) { s->ga[r] = s->ga[r] + ar; s->gb[r] = s->gb[r] + br; }
                else {
                    s->ga[2] = rfma(s->jx[r], ar, s->ga[2]);
                    s->ga[3] = rfma(s->jy[r], ar, s->ga[3]);
                    s->gb[2] = rfma(s->jx[r], br, s->gb[2]);
                    s->gb[3] = rfma(s->jy[r], br, s->gb[3]);
                }
            }
        }
        s1 = s1 + s1k;
        s2 = s2 + s2k;
        real ak = num / den;
        if (ak < alpha) alpha = ak;
    }
    if (S1) *S1 = s1;
    if (S2) *S2 = s2;
    return alpha;
}

static void qp_update(const problem *pb, const mpcx_opts *o, stage *st, int mode,
                      real sigma_mu, real alpha)
{
    const int N = pb->N;
    const real lam_min = o->lam_min, t_min = o->t_min;
    for (int k = 0; k <= N; ++k) {
        stage *s = &st[k];
        for (int r = s->r0; r < s->r1; ++r) {
            if (!((s->rmask >> r) & 1u)) continue;
            real til = (real)1.0 / s->t_l[r], tiu = (real)1.0 / s->t_u[r];
            real rml = s->lam_l[r] * s->t_l[r], rmu = s->lam_u[r] * s->t_u[r];
            if (mode == 1) {
                real jd = row_dot(s, r, s->dva);
                real dtl = jd - s->rd_l[r];
                real dtu = -jd - s->rd_u[r];
                real dll = -(rfma(s->lam_l[r], dtl, rml) * til);
                real dlu = -(rfma(s->lam_u[r], dtu, rmu) * tiu);
                rml = rfma(dtl, dll, rml) - sigma_mu;
                rmu = rfma(dtu, dlu, rmu) - sigma_mu;
            } else if (mode == 2) {
                rml = rml - sigma_mu;
                rmu = rmu - sigma_mu;
            }
            real jd = row_dot(s, r, s->dv);
            real dtl = jd - s->rd_l[r];
            real dtu = -jd - s->rd_u[r];
            real dll = -(rfma(s->lam_l[r], dtl, rml) * til);
            real dlu = -(rfma(s->lam_u[r], dtu, rmu) * tiu);
            real v;
            v = rfma(alpha, dll, s->lam_l[r]); s->lam_l[r] = (v < lam_min) ? lam_min : v;
            v = rfma(alpha, dlu, s->lam_u[r]); s->lam_u[r] = (v < lam_min) ? lam_min : v;
            v = rfma(alpha, dtl, s->t_l[r]);   s->t_l[r] = (v < t_min) ? t_min : v;
            v = rfma(alpha, dtu, s->t_u[r]);   s->t_u[r] = (v < t_min) ? t_min : v;
        }
        int j0 = (k == N) ? NU : 0, j1 = (k == 0) ? NU : NZ;
        for (int j = j0; j < j1; ++j) s->v[j] = rfma(alpha, s->dv[j], s->v[j]);
        if (k > 0)
            for (int i = 0; i < NX; ++i) s->qpi[i] = rfma(alpha, s->dpi[i], s->qpi[i]);
    }
}

/* returns HPIPM-style status: 0 solved, 1 iteration limit, 2 minimum step, 3 NaN */
static int qp_solve(const problem *pb, const mpcx_opts *o, stage *st, int *iters)
{
    const int N = pb->N;
    int m_total = 0;
    for (int k = 0; k <= N; ++k)
        for (int r = st[k].r0; r < st[k].r1; ++r) m_total += 2 * (int)((st[k].rmask >> r) & 1u);
    qp_res R;
    qp_init(pb, o, st);
    qp_residuals(pb, st, m_total, &R);
    int it = 0, status = 1;
    for (;;) {
        if (!(r_isfinite(R.g) && r_isfinite(R.b) && r_isfinite(R.d) && r_isfinite(R.m))) { status = 3; break; }
        if (R.g < (real)o->qp_tol_stat && R.b < (real)o->qp_tol_eq &&
            R.d < (real)o->qp_tol_ineq && R.m < (real)o->qp_tol_comp) { status = 0; break; }
        if (it >= o->max_qp_iter) { status = 1; break; }
        /* predictor */
        qp_build_rhs(pb, st, 0, (real)0.0);
        riccati_factor(pb, st);
        riccati_solve(pb, st, 0);
        real S1, S2;
        real a_aff = qp_step_length(pb, st, 0, (real)0.0, &S1, &S2);
        real S0 = R.mu * (real)(double)m_total;
        real mu_aff = rfma(a_aff, rfma(a_aff, S2, S1), S0) / (real)(double)m_total;
        if (mu_aff < (real)0.0) mu_aff = (real)0.0;
        real ratio = mu_aff / R.mu;
        real sigma = ratio * ratio * ratio;
        real sigma_mu = sigma * R.mu;
        for (int k = 0; k <= N; ++k)
            for (int j = 0; j < NZ; ++j) st[k].dva[j] = st[k].dv[j];
        /* corrector */
        int mode = 1;
        qp_build_rhs(pb, st, 1, sigma_mu);
        riccati_solve(pb, st, 1);
        real alpha = qp_step_length(pb, st, 1, sigma_mu, NULL, NULL);
        if (o->cond_pred_corr && alpha < a_aff) {
            mode = 2;
            qp_build_rhs(pb, st, 2, sigma_mu);
            riccati_solve(pb, st, 1);
            alpha = qp_step_length(pb, st, 2, sigma_mu, NULL, NULL);
        }
        if (g_debug)
            fprintf(stderr, "   ipm %2d: res g %.2e b %.2e d %.2e m %.2e mu %.2e | a_aff %.3e sigma %.2e alpha %.3e mode %d\n",
                    it, R2D(R.g), R2D(R.b), R2D(R.d), R2D(R.m), R2D(R.mu), R2D(a_aff), R2D(sigma), R2D(alpha), mode);
        if (!(alpha > (real)o->alpha_min)) { status = r_isfinite(alpha) ? 2 : 3; ++it; break; }
        alpha = alpha * rfma(alpha, (real)0.9999999, ((real)1.0 - alpha) * (real)0.99);
        qp_update(pb, o, st, mode, sigma_mu, alpha);
        ++it;
        qp_residuals(pb, st, m_total, &R);
    }
    *iters = it;
    return status;
}

/* ------------------------------------------------------------------ */
/* SQP driver for one problem                                           */
/* ------------------------------------------------------------------ */
static real nlp_cost(const problem *pb, const stage *st)
{
    real c = 0.0;      /* stage costs are summed per stage first, then over the stages */
    for (int k = 0; k < pb->N; ++k) {
        real ck = 0.0;
        for (int j = 0; j < NZ; ++j) {
            real z = (j < NU) ? st[k].u[j] : (st[k].x[j - NU] - pb->goal[j - NU]);
            real hw = ((real)0.5 * pb->h) * pb->W[j];
            ck = rfma(hw * z, z, ck);
        }
        c = c + ck;
    }
    real ce = 0.0;
    for (int i = 0; i < NX; ++i) {
        real z = st[pb->N].x[i] - pb->goal[i];
        ce = rfma(((real)0.5 * pb->We[i]) * z, z, ce);
    }
    return c + ce;
}

typedef struct { int status, sqp_iter, qp_iter; real cost, res[4]; } outcome;

/* Line search (mpcx_opts.globalization = 1): objective, largest dynamics gap and largest violation of
 * an inequality row at w + alpha * d, d = the QP's primal solution; *msum = sum of the absolute values
 * of the QP's multipliers (dynamics rows and both sides of the inequality rows).  Nothing is changed.
 * Order of the sums: per stage, then over the stages in ascending order. */
static void merit_eval(const problem *pb, const stage *st, real alpha, real *cost, real *re, real *ri, real *msum)
{
    const int N = pb->N;
    real c = 0.0, gap = 0.0, viol = 0.0, ms = 0.0;
    real xk[NX], xn1[NX] = {0.0, 0.0, 0.0, 0.0, 0.0};
    for (int i = 0; i < NX; ++i) xk[i] = st[0].x[i] + alpha * st[0].v[NU + i];
    for (int k = 0; k <= N; ++k) {
        const stage *s = &st[k];
        real uk[NU] = {0.0, 0.0};
        if (k < N) {
            for (int cc = 0; cc < NU; ++cc) uk[cc] = s->u[cc] + alpha * s->v[cc];
            for (int i = 0; i < NX; ++i) xn1[i] = st[k + 1].x[i] + alpha * st[k + 1].v[NU + i];
            real xn[NX], A[NX][NX], B[NX][NU];
            rk4_sens(pb, xk, uk, xn, A, B);
            for (int i = 0; i < NX; ++i) gap = rmax(gap, rabs(xn[i] - xn1[i]));
            real ck = 0.0;
            for (int j = 0; j < NZ; ++j) {
                real z = (j < NU) ? uk[j] : (xk[j - NU] - pb->goal[j - NU]);
                real hw = ((real)0.5 * pb->h) * pb->W[j];
                ck = rfma(hw * z, z, ck);
            }
            c = c + ck;
        } else {
            real ce = 0.0;
            for (int i = 0; i < NX; ++i) {
                real z = xk[i] - pb->goal[i];
                ce = rfma(((real)0.5 * pb->We[i]) * z, z, ce);
            }
            c = c + ce;
        }
        if (k == 0)
            for (int i = 0; i < NX; ++i) viol = rmax(viol, rabs(pb->x0[i] - xk[i]));
        const real *lo = row_lo(pb, k), *up = row_up(pb, k);
        real sk = 0.0;
        if (k > 0) for (int i = 0; i < NX; ++i) sk = sk + rabs(s->qpi[i]);
        for (int r = s->r0; r < s->r1; ++r) {
            if (!((s->rmask >> r) & 1u)) continue;
            real cv = row_value(pb, r, xk, uk);
            viol = rmax(viol, lo[r] - cv);
            viol = rmax(viol, cv - up[r]);
            sk = sk + rabs(s->lam_l[r]);
            sk = sk + rabs(s->lam_u[r]);
        }
        ms = ms + sk;
        for (int i = 0; i < NX; ++i) xk[i] = xn1[i];
    }
    *cost = c; *re = gap; *ri = viol; *msum = ms;
}

static real line_search(const problem *pb, const mpcx_opts *o, const stage *st, real cost0, real viol0, real *rho)
{
    real alpha = 1.0;
    real phi0 = 0.0;
    for (int trial = 0;; ++trial) {
        real c, re, ri, ms;
        merit_eval(pb, st, alpha, &c, &re, &ri, &ms);
        if (trial == 0) {
            real want = (real)o->ls_rho_factor * ms;
            if (want > *rho) *rho = want;
            phi0 = rfma(*rho, viol0, cost0);
        }
        real phi = rfma(*rho, rmax(re, ri), c);
        if (phi <= phi0 - (((real)o->ls_c1 * alpha) * *rho) * viol0) return alpha;
        if (alpha * (real)o->ls_shrink < (real)o->ls_alpha_min) return alpha;
        alpha = alpha * (real)o->ls_shrink;
    }
}

static void sqp_solve(const problem *pb, const mpcx_opts *o, stage *st, outcome *out)
{
    const int N = pb->N;
    int status = MPCX_MAXITER, it = 0, qp_total = 0;
    real res[4] = {0.0, 0.0, 0.0, 0.0};
    real cost0 = 0.0, rho = 0.0;      /* line search: objective at the current iterate, penalty weight */
    for (int k = 0; k <= N; ++k) {
        for (int i = 0; i < NX; ++i) st[k].pi[i] = (real)0.0;
        for (int r = 0; r < MAXR; ++r) st[k].nl_l[r] = st[k].nl_u[r] = (real)0.0;
    }
    for (;;) {
#ifdef MPCX_COUNT_FLOPS
        g_bucket = 0;
#endif
        linearize(pb, o, st);
        nlp_residuals(pb, st, res);
        if (o->globalization) cost0 = nlp_cost(pb, st);
        if (!(r_isfinite(res[0]) && r_isfinite(res[1]) && r_isfinite(res[2]) && r_isfinite(res[3]))) {
            status = MPCX_NAN; break;
        }
        if (g_debug)
            fprintf(stderr, "sqp %3d: stat %.3e eq %.3e ineq %.3e comp %.3e\n", it,
                    R2D(res[0]), R2D(res[1]), R2D(res[2]), R2D(res[3]));
        if (!o->rti && res[0] < (real)o->tol_stat && res[1] < (real)o->tol_eq &&
            res[2] < (real)o->tol_ineq && res[3] < (real)o->tol_comp) { status = MPCX_SUCCESS; break; }
        if (it >= o->max_sqp_iter) { status = MPCX_MAXITER; break; }
#ifdef MPCX_COUNT_FLOPS
        g_bucket = 1;
#endif
        int qit = 0;
        int qs = qp_solve(pb, o, st, &qit);
        qp_total += qit;
        ++it;
#ifdef MPCX_COUNT_FLOPS
        g_bucket = 0;
#endif
        if (qs != 0 && qs != 1) { status = MPCX_QP_FAILURE; break; }
        real alpha = 1.0;
        if (o->globalization && !o->rti) alpha = line_search(pb, o, st, cost0, rmax(res[1], res[2]), &rho);
        for (int k = 0; k <= N; ++k) {
            stage *s = &st[k];
            for (int i = 0; i < NX; ++i) s->x[i] = s->x[i] + alpha * s->v[NU + i];
            if (k < N) for (int c = 0; c < NU; ++c) s->u[c] = s->u[c] + alpha * s->v[c];
            if (k > 0) for (int i = 0; i < NX; ++i) s->pi[i] = s->qpi[i];
            for (int r = s->r0; r < s->r1; ++r)
                if ((s->rmask >> r) & 1u) { s->nl_l[r] = s->lam_l[r]; s->nl_u[r] = s->lam_u[r]; }
        }
        if (o->rti) { status = MPCX_SUCCESS; break; }
    }
    out->status = status;
    out->sqp_iter = it;
    out->qp_iter = qp_total;
    out->cost = nlp_cost(pb, st);
    for (int i = 0; i < 4; ++i) out->res[i] = res[i];
}

/* ------------------------------------------------------------------ */
/* batch front end: same tables as the HIP library                      */
/* ------------------------------------------------------------------ */
static void load_problem(const mpcx_batch *in, int b, problem *pb)
{
    const int B = in->B;
#define PD(f) ((real)in->pdata[(size_t)(f) * B + b])
    pb->N = in->N;
    pb->nobs = in->n_obs ? in->n_obs[b] : 0;
    pb->h = PD(MPCX_F_H);
    pb->invL = (real)1.0 / PD(MPCX_F_L);
    for (int i = 0; i < NX; ++i) { pb->x0[i] = PD(MPCX_F_X0 + i); pb->goal[i] = PD(MPCX_F_GOAL + i); pb->We[i] = PD(MPCX_F_WE + i); }
    for (int j = 0; j < NZ; ++j) pb->W[j] = PD(MPCX_F_W + j);
    for (int r = 0; r < MAXR; ++r) { pb->lo_s[r] = pb->lo_e[r] = (real)0.0; pb->up_s[r] = pb->up_e[r] = (real)0.0; }
    for (int c = 0; c < NU; ++c) { pb->lo_s[c] = PD(MPCX_F_LBU + c); pb->up_s[c] = PD(MPCX_F_UBU + c); }
    for (int i = 0; i < NX; ++i) {
        pb->lo_s[NU + i] = PD(MPCX_F_LBX + i);  pb->up_s[NU + i] = PD(MPCX_F_UBX + i);
        pb->lo_e[NU + i] = PD(MPCX_F_LBXE + i); pb->up_e[NU + i] = PD(MPCX_F_UBXE + i);
    }
    for (int r = 0; r < 2; ++r) {
        pb->cg[r][0] = PD(MPCX_F_CG + 2 * r);
        pb->cg[r][1] = PD(MPCX_F_CG + 2 * r + 1);
        pb->lo_s[ROW_G0 + r] = pb->lo_e[ROW_G0 + r] = PD(MPCX_F_LG + r);
        pb->up_s[ROW_G0 + r] = pb->up_e[ROW_G0 + r] = PD(MPCX_F_UG + r);
    }
    for (int j = 0; j < MPCX_MAX_OBS; ++j) {
        for (int q = 0; q < 3; ++q) pb->obs[j][q] = PD(MPCX_F_OBS + 3 * j + q);
        pb->lo_s[9 + j] = PD(MPCX_F_LH);
        pb->up_s[9 + j] = PD(MPCX_F_UH);
    }
#undef PD
}

#ifdef __cplusplus
extern "C" {
#endif

void mpcx_oracle_opts_default(mpcx_opts *o)
{
    memset(o, 0, sizeof(*o));
    o->max_sqp_iter = 200;
    o->max_qp_iter = 50;
    o->tol_stat = o->tol_eq = o->tol_ineq = o->tol_comp = 1e-3;
    /* QP tolerances: HPIPM's "balance" values 1e-6 / 1e-8 / 1e-8, and 1e-6 for the complementarity
     * products (measured here as max lam*t, which levels off above 1e-8 on the 1e9 sides).  With these
     * and the Levenberg-Marquardt term scaled by the interval length the reference's recorded
     * trajectories are reproduced to 6e-7, the resolution they were recorded with; with the QP
     * stopped at the NLP tolerance 1e-3 the distance is 1e-4 (profiles/r02_switch_sweep*.md). */
    o->qp_tol_stat = 1e-6;
    o->qp_tol_eq = o->qp_tol_ineq = 1e-8;
    o->qp_tol_comp = 1e-6;
    o->lm = 1e-2;
    o->lm_scale_by_h = 1;
    o->cond_pred_corr = 1;
    o->mu0 = 1e1;
    o->thr0 = 1e-1;
    o->alpha_min = 1e-8;
    o->lam_min = 1e-16;
    o->t_min = 1e-16;
    o->rti = 0;
    o->max_obs = -1;
    o->globalization = 0;
    o->ls_shrink = 0.5;
    o->ls_alpha_min = 1.0 / 64.0;
    o->ls_c1 = 1e-4;
    o->ls_rho_factor = 2.0;
}

/* Solves problems [b0, b1) of the batch; nthreads > 1 uses OpenMP. */
int mpcx_oracle_solve_batch(const mpcx_batch *in, const mpcx_opts *o, mpcx_result *out,
                            int b0, int b1, int nthreads)
{
    if (!in || !o || !out || in->N < 1 || b0 < 0 || b1 > in->B) return MPCX_E_BADARG;
    g_debug = getenv("MPCX_ORACLE_DEBUG") ? atoi(getenv("MPCX_ORACLE_DEBUG")) : 0;
    g_stage0_rows = getenv("MPCX_ORACLE_STAGE0_ROWS") ? atoi(getenv("MPCX_ORACLE_STAGE0_ROWS")) : 0;
    g_cond_blocks = getenv("MPCX_ORACLE_COND_BLOCKS") ? atoi(getenv("MPCX_ORACLE_COND_BLOCKS")) : 0;
    g_dpi_costate = getenv("MPCX_ORACLE_DPI_COSTATE") ? atoi(getenv("MPCX_ORACLE_DPI_COSTATE")) : 0;
    const int N = in->N, B = in->B;
    int fail = 0;
#ifdef _OPENMP
#pragma omp parallel num_threads(nthreads > 0 ? nthreads : 1)
#endif
    {
        stage *st = (stage *)calloc((size_t)N + 1, sizeof(stage));
        if (!st) {
            fail = 1;
        } else {
#ifdef _OPENMP
#pragma omp for schedule(dynamic, 16)
#endif
            for (int b = b0; b < b1; ++b) {
                problem pb;
                outcome oc;
                load_problem(in, b, &pb);
                if (pb.nobs < 0 || pb.nobs > MPCX_MAX_OBS) { out->status[b] = MPCX_NAN; continue; }
                for (int k = 0; k <= N; ++k)
                    for (int i = 0; i < NX; ++i)
                        st[k].x[i] = (real)in->xguess[((size_t)k * NX + i) * B + b];
                for (int k = 0; k <= N; ++k)
                    for (int c = 0; c < NU; ++c)
                        st[k].u[c] = (k < N && in->uguess) ? (real)in->uguess[((size_t)k * NU + c) * B + b]
                                                           : (real)0.0;
                sqp_solve(&pb, o, st, &oc);
                for (int k = 0; k <= N; ++k)
                    for (int i = 0; i < NX; ++i)
                        out->X[((size_t)b * (N + 1) + k) * NX + i] = R2D(st[k].x[i]);
                for (int k = 0; k < N; ++k)
                    for (int c = 0; c < NU; ++c)
                        out->U[((size_t)b * N + k) * NU + c] = R2D(st[k].u[c]);
                out->status[b] = oc.status;
                out->sqp_iter[b] = oc.sqp_iter;
                out->qp_iter[b] = oc.qp_iter;
                out->cost[b] = R2D(oc.cost);
                for (int i = 0; i < 4; ++i) out->res[(size_t)i * B + b] = R2D(oc.res[i]);
            }
            free(st);
        }
    }
    (void)nthreads;
    return fail ? MPCX_E_NOMEM : 0;
}

/* Debug entry: the constraint set of problem b's FIRST QP (linearisation at the initial guess), for
 * feasibility studies with an independent LP solver.  Doubles per stage k = 0..N, MPCX_ORACLE_QP_STRIDE each:
 *   [0,25)  A row-major   [25,35) B row-major   [35,40) gap b   (interval k -> k+1; zeros at k = N)
 *   [40,45) x0 - x_0 at k = 0 (the fixed value of the state part of the QP variable), zeros elsewhere
 *   then for every row r = 0..MAXR-1 six doubles: present (1/0), c, jx, jy, lo, up
 * so that the QP's feasible set is  dx_(k+1) = A dx_k + B du_k + b,  dx_0 = x0 - x_0,
 * lo - c <= J_r [du; dx] <= up - c  for the rows present (J_r = e_r for r < 7, (jx, jy) on (x, y) else). */
#define MPCX_ORACLE_QP_STRIDE (45 + 6 * MAXR)
int mpcx_oracle_qp_stride(void) { return MPCX_ORACLE_QP_STRIDE; }

int mpcx_oracle_first_qp(const mpcx_batch *in, const mpcx_opts *o, int b, double *out)
{
    if (!in || !o || !out || in->N < 1 || b < 0 || b >= in->B || !in->xguess) return MPCX_E_BADARG;
    g_stage0_rows = getenv("MPCX_ORACLE_STAGE0_ROWS") ? atoi(getenv("MPCX_ORACLE_STAGE0_ROWS")) : 0;
    const int N = in->N, B = in->B;
    stage *st = (stage *)calloc((size_t)N + 1, sizeof(stage));
    if (!st) return MPCX_E_NOMEM;
    problem pb;
    load_problem(in, b, &pb);
    if (pb.nobs < 0 || pb.nobs > MPCX_MAX_OBS) { free(st); return MPCX_E_BADARG; }
    for (int k = 0; k <= N; ++k) {
        for (int i = 0; i < NX; ++i) st[k].x[i] = (real)in->xguess[((size_t)k * NX + i) * B + b];
        for (int c = 0; c < NU; ++c)
            st[k].u[c] = (k < N && in->uguess) ? (real)in->uguess[((size_t)k * NU + c) * B + b] : (real)0.0;
    }
    linearize(&pb, o, st);
    for (int k = 0; k <= N; ++k) {
        const stage *s = &st[k];
        double *q = out + (size_t)k * MPCX_ORACLE_QP_STRIDE;
        for (int i = 0; i < MPCX_ORACLE_QP_STRIDE; ++i) q[i] = 0.0;
        if (k < N) {
            for (int i = 0; i < NX; ++i) {
                for (int j = 0; j < NX; ++j) q[i * NX + j] = R2D(s->A[i][j]);
                for (int c = 0; c < NU; ++c) q[25 + i * NU + c] = R2D(s->B[i][c]);
                q[35 + i] = R2D(s->b[i]);
            }
        }
        if (k == 0)
            for (int i = 0; i < NX; ++i) q[40 + i] = R2D(pb.x0[i] - s->x[i]);
        const real *lo = row_lo(&pb, k), *up = row_up(&pb, k);
        for (int r = s->r0; r < s->r1; ++r) {
            if (!((s->rmask >> r) & 1u)) continue;
            double *w = q + 45 + 6 * r;
            w[0] = 1.0;
            w[1] = R2D(s->c[r]);
            w[2] = (r >= ROW_G0) ? R2D(s->jx[r]) : 0.0;
            w[3] = (r >= ROW_G0) ? R2D(s->jy[r]) : 0.0;
            w[4] = R2D(lo[r]);
            w[5] = R2D(up[r]);
        }
    }
    free(st);
    return 0;
}

/* sin/cos as the solver computes them, for the accuracy test */
void mpcx_oracle_sincos(double x, double *s, double *c)
{
    real rs, rc;
    mx_sincos((real)x, &rs, &rc);
    *s = R2D(rs);
    *c = R2D(rc);
}

/* one RK4 step with sensitivities, for the dynamics test:
 * xn[5], A[25] row-major, Bm[10] row-major */
void mpcx_oracle_rk4(double h, double L, const double *x, const double *u,
                     double *xn, double *A, double *Bm)
{
    problem pb;
    memset((void *)&pb, 0, sizeof(pb));
    pb.h = (real)h;
    pb.invL = (real)1.0 / (real)L;
    real xr[NX], ur[NU], xo[NX], Ar[NX][NX], Br[NX][NU];
    for (int i = 0; i < NX; ++i) xr[i] = (real)x[i];
    for (int c = 0; c < NU; ++c) ur[c] = (real)u[c];
    rk4_sens(&pb, xr, ur, xo, Ar, Br);
    for (int i = 0; i < NX; ++i) {
        xn[i] = R2D(xo[i]);
        for (int j = 0; j < NX; ++j) A[i * NX + j] = R2D(Ar[i][j]);
        for (int c = 0; c < NU; ++c) Bm[i * NU + c] = R2D(Br[i][c]);
    }
}

#ifdef MPCX_COUNT_FLOPS
/* flops counted since the last reset: [0] per-SQP-iteration work, [1] interior point */
void mpcx_oracle_flops(unsigned long long *out2, int reset)
{
    out2[0] = g_flops[0];
    out2[1] = g_flops[1];
    if (reset) g_flops[0] = g_flops[1] = 0;
}
#endif

#ifdef __cplusplus
}
#endif
