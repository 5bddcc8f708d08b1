// mpcx_kernels.hip — batched kinematic-car OCP solve on gfx950 (MI355X).
//
// Mapping.  One persistent kernel, 64-lane workgroups, one wavefront per SIMD.  A
// wavefront works in one of two modes, per tick:
//
//  * one LANE per problem (the bulk of a large batch).  The Riccati recursion is a
//    serial chain over the N stages of ONE problem; 64 independent chains keep every
//    FP64 instruction full.  The price is that a lane's working set (iterate,
//    linearisation, multipliers and Riccati factors of all N+1 stages, ~1.0 KB per
//    stage) cannot live on chip: it sits in an HBM workspace laid out
//    [wave][stage][field][lane], so each wavefront load/store is one contiguous
//    512-byte segment (with the non-temporal hint: nothing is reused before megabytes
//    have passed), and the solver is organised as a few fused forward/backward
//    SWEEPS over the stages that touch each field as rarely as possible:
//
//      L   forward : apply the SQP step, RK4 + sensitivities, NLP residuals,
//                    cold start of the interior point (one per SQP iteration)
//      B0  backward: apply the previous interior-point step (the step of the dynamics
//                    multipliers is formed here, from the stationarity of the Newton
//                    system it solves), residuals, barrier Hessian, Riccati
//                    factorisation + backward substitution
//      F   forward : affine step, its longest feasible length, centring sums
//      Bs  backward: corrector AND centring-only right-hand sides, both substituted
//      F   forward : both steps and their lengths (the centring-only one is taken
//                    when the corrector would shorten the step)
//
//  * one lane per STAGE of a single problem (old problems, few problems left in the
//    wavefront, small batches, long horizons): the rows of 64 stages are computed at once;
//    the recursions run as pipelines over the lanes, except the Riccati recursion itself
//    (factors and affine substitution), whose stages are taken in turn with the algebra of
//    one stage spread over the lanes (ric_spread); see "single-problem mode" below.
//
// Both modes perform the same operations in the same order for a problem.  On large batches the two modes
// are also two kinds of wavefront: problems that have grown old in the first mode are handed over, through a
// queue in the workspace header, to worker wavefronts that do nothing but the second (see the kernel).
//
// The problem is the one reference src/core_solver.py:127-351 poses and hands
// to acados at :354/:502; the algorithm is the one its options block :242-253
// selects (SQP, ERK4 + forward sensitivities, Gauss-Newton + Levenberg-
// Marquardt, interior-point QP by stage-wise Riccati recursion).
//
// Arithmetic contract: built with -ffp-contract=off; every fused multiply-add
// is an explicit fma(); sin/cos are the polynomials below, not the device
// library's.  Structural zeros and ones of the sensitivity matrices are
// skipped, which leaves every remaining operation and its order unchanged, so
// a CPU program that performs the dense sequence gets the same bits.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#include "../../include/mpcx.h"
#include "mpcx_kernels.h"

#define NX 5
#define NU 2
#define NZ 7
#define ROW_G0 7

#define DEV __device__ __forceinline__

// Pointers that cross a (non-inlined) function boundary carry their address space in the
// type, so that the callee still emits global_load / ds_read rather than flat accesses.
typedef __attribute__((address_space(1))) double gdouble;
typedef __attribute__((address_space(3))) double ldouble;
// ... and so do the solver options (global memory, behind the problem counter in the workspace header): a
// generic pointer would make every read of them a flat access, which the hardware may complete out of order
// with the streaming loads around it and which the compiler therefore brackets with waits for everything
typedef __attribute__((address_space(1))) const mpcx_opts gopts;

// The reciprocals of the slacks are needed in all four sweeps of an interior-point iteration.
// 1: computed once per iteration and kept in the workspace; 0: recomputed where needed, which
// trades 72 divisions per stage and iteration for 90 of the ~615 doubles of workspace traffic.
#ifndef MPCX_STORE_RECIP
#define MPCX_STORE_RECIP 0
#endif

namespace {

// ---------------------------------------------------------------------------
// workspace field map (doubles per stage and lane), R = rows per stage
// ---------------------------------------------------------------------------
template <int NH>
struct Map {
    static constexpr int R = 9 + NH;
    static constexpr int X = 0;                // 5  NLP state
    static constexpr int U = X + 5;            // 2  NLP input
    static constexpr int AB = U + 2;           // 14 non-trivial sensitivities (see AbIdx)
    static constexpr int BV = AB + 14;         // 5  dynamics gap
    static constexpr int V = BV + 5;           // 7  QP primal [u; x]
    static constexpr int QPI = V + 7;          // 5  QP dynamics multiplier
    static constexpr int LAML = QPI + 5;       // R  multipliers, lower sides
    static constexpr int LAMU = LAML + R;      // R
    static constexpr int TL = LAMU + R;        // R  slacks
    static constexpr int TU = TL + R;          // R
    static constexpr int RS = MPCX_STORE_RECIP ? R : 0;
    static constexpr int TIL = TU + R;         // RS 1 / slack (only if stored, see MPCX_STORE_RECIP)
    static constexpr int TIU = TIL + RS;       // RS
    static constexpr int RG = TIU + RS;        // 5  stationarity residual, state part (what the next backward sweep reads)
    static constexpr int RB = RG + 5;          // 5  dynamics residual
    static constexpr int K = RB + 5;           // 10 feedback gain, row-major [2][5]
    static constexpr int LDL = K + 10;         // 3  l10, 1/d0, 1/d1 of the LDL' of M_uu
    static constexpr int GA = LDL + 3;         // 7  rows' part of the corrections' right-hand sides: the corrector term ...
    static constexpr int GB = GA + 7;          // 7  ... and what sigma*mu multiplies (left by the affine forward sweep)
    static constexpr int KV = GB + 7;          // 2  feed-forward
    static constexpr int DV = KV + 2;          // 7  step
    static constexpr int DVA = DV + 7;         // 7  affine (predictor) step
    // the centring-only solution, computed alongside the corrector one; same order and
    // distance as KV, DV so that "solution 2" is "solution 1" + SOL2
    static constexpr int KV2 = DVA + 7;        // 2
    static constexpr int DV2 = KV2 + 2;        // 7
    static constexpr int SOL2 = KV2 - KV;
    static constexpr int F = DV2 + 7;
    // (The cost-to-go Hessian and gradient never reach the workspace: the step of the dynamics
    // multipliers is formed by the next backward sweep from the stationarity of the Newton system.)
    static_assert(DV2 - DV == SOL2, "solution 2 layout");
};

// A = dPhi/dx and B = dPhi/du of one RK4 step have this pattern (h* = 6*(h/6)):
//   A = [1 0 a02 a03 a04; 0 1 a12 a13 a14; 0 0 1 a23 a24; 0 0 0 1 0; 0 0 0 0 1]
//   B = [b00 b01; b10 b11; b20 b21; h* 0; 0 h*]
struct AB {
    double a02, a03, a04, a12, a13, a14, a23, a24;
    double b00, b10, b20, b01, b11, b21;
};

struct Lane {
    gdouble *wave;        // workspace of this wavefront (same in all lanes); element (k, f)
                          // of lane l is wave[(k*F + f) * 64 + l]
    const gdouble *pdata; // problem table (same in all lanes); field f of problem b at [f*B + b]
    size_t B;
    unsigned lane;        // column of the workspace this lane works in, as a BYTE offset (8 * 0..63);
                          // in single-problem mode it also carries the offset of the lane's stage
    unsigned region;      // 0: the workspace is laid out [stage][field][lane]; else the wavefront never
                          // leaves single-problem mode and every problem has its own region of this
                          // many bytes, laid out [block of 64 stages][field][stage in block]
    unsigned roff;        // not 0: this lane's problem has been moved to a region of its own (same
                          // layout as above) that starts at this byte offset of the wavefront's workspace
    unsigned b;           // problem this lane is solving
    int N;
    int nobs;
    int F;
    double h, hh, h6, hB, invL;
};

// What a non-inlined single-problem sweep needs of a Lane and of the problem constants.  The
// calling convention hands structs of this size over in scratch memory, a round trip per call
// that costs microseconds on a busy GPU and four to five calls make an iteration; as separate
// scalar parameters (COOP_PARAMS, at most 32 registers with the other arguments) they travel
// in registers.
#define COOP_PARAMS                                                                                  \
    gdouble *a_wave, unsigned a_lane, unsigned a_region, int a_N, int a_nobs, int a_F, double a_h,   \
        double a_hh, double a_h6, double a_hB, double a_invL, ldouble *a_pcp, double a_lmk
#define COOP_UNPACK(NH_)                                                                             \
    const Lane C = lane_of(LaneArgs{a_wave, a_lane, a_region, a_N, a_nobs, a_F, a_h, a_hh, a_h6, a_hB, a_invL}); \
    PC<NH_> pc;                                                                                      \
    pc.p = a_pcp;                                                                                    \
    pc.lmk = a_lmk
#define COOP_PASS(C_, pc_)                                                                           \
    (C_).wave, (C_).lane, (C_).region, (C_).N, (C_).nobs, (C_).F, (C_).h, (C_).hh, (C_).h6, (C_).hB, (C_).invL, \
        (pc_).p, (pc_).lmk
struct LaneArgs {
    gdouble *wave;
    unsigned lane, region;
    int N, nobs, F;
    double h, hh, h6, hB, invL;
};
__device__ __forceinline__ Lane lane_of(const LaneArgs &a)
{
    Lane L;
    L.wave = a.wave; L.pdata = nullptr; L.B = 0; L.lane = a.lane; L.region = a.region; L.roff = 0; L.b = 0;
    L.N = a.N; L.nobs = a.nobs; L.F = a.F; L.h = a.h; L.hh = a.hh; L.h6 = a.h6; L.hB = a.hB; L.invL = a.invL;
    return L;
}

// The wave-uniform part of every address stays in scalar registers; only the
// lane (or problem) index is a vector offset.
// (a 32-bit byte offset on a uniform base is the addressing mode global_load has: base in
// scalar registers, offset in one vector register)
typedef __attribute__((address_space(1))) char gchar;
DEV gdouble *wsp(const Lane &L, int k, int f, unsigned extra)
{
    gchar *base = (gchar *)(L.wave + ((size_t)k * L.F + f) * 64);
    return (gdouble *)(base + (L.lane + extra));
}
// Stage code is instantiated per kind of stage: FIRST / MID / LAST in one-lane-per-problem
// mode, where the kind is known at compile time, and ANY in single-problem mode, where each
// lane has a stage of its own and the kind is a run-time value.
enum { FIRST = 0, MID = 1, LAST = 2, ANY = 3, MIDC = 4 };
// Workspace accesses.  In one-lane-per-problem mode a wavefront streams through megabytes
// between two uses of the same word, so those accesses carry the non-temporal hint and do not
// push the single-problem regions (reused every ~100 us) out of the L2.
#ifndef MPCX_STREAM_HINT
#define MPCX_STREAM_HINT 1
#endif
// ... except in the stages at the two ends of a sweep (the first and the last stage and this many next to
// each): the next sweep runs the other way and starts with what this one has just touched.
#ifndef MPCX_EDGE_STAGES
#define MPCX_EDGE_STAGES 8
#endif
template <int KIND> DEV double ldw(const Lane &L, int k, int f)
{
    if (MPCX_STREAM_HINT && KIND == MID) return __builtin_nontemporal_load(wsp(L, k, f, 0u));
    return *wsp(L, k, f, 0u);
}
template <int KIND> DEV void stw(const Lane &L, int k, int f, double v)
{
    if (MPCX_STREAM_HINT && KIND == MID) __builtin_nontemporal_store(v, wsp(L, k, f, 0u));
    else *wsp(L, k, f, 0u) = v;
}
DEV double ldp(const Lane &L, int f) { return (L.pdata + (size_t)f * L.B)[L.b]; }
// field f + df where df differs from lane to lane
template <int KIND> DEV double ldw_lane(const Lane &L, int k, int f, int df)
{
    if (MPCX_STREAM_HINT && KIND == MID) return __builtin_nontemporal_load(wsp(L, k, f, 512u * (unsigned)df));
    return *wsp(L, k, f, 512u * (unsigned)df);
}

// The options as the sweeps use them: fetched once per call into scalar registers (they are the same in
// every lane).  Read through the pointer where they are used, they are fetched again after every store
// the compiler cannot tell apart from them -- a dozen trips to the L2 per stage or per iteration, several
// of them one after the other in the stopping tests, microseconds each on a busy GPU.  (What a sweep does
// not use costs nothing: the copies are dropped with their loads.)
DEV double sreg(double v)
{
    const int lo = __builtin_amdgcn_readfirstlane(__double2loint(v)), hi = __builtin_amdgcn_readfirstlane(__double2hiint(v));
    return __hiloint2double(hi, lo);
}
DEV mpcx_opts opts_in_registers(gopts *op)
{
    mpcx_opts o;
    o.max_sqp_iter = __builtin_amdgcn_readfirstlane(op->max_sqp_iter);
    o.max_qp_iter = __builtin_amdgcn_readfirstlane(op->max_qp_iter);
    o.tol_stat = sreg(op->tol_stat); o.tol_eq = sreg(op->tol_eq); o.tol_ineq = sreg(op->tol_ineq); o.tol_comp = sreg(op->tol_comp);
    o.qp_tol_stat = sreg(op->qp_tol_stat); o.qp_tol_eq = sreg(op->qp_tol_eq);
    o.qp_tol_ineq = sreg(op->qp_tol_ineq); o.qp_tol_comp = sreg(op->qp_tol_comp);
    o.lm = sreg(op->lm);
    o.lm_scale_by_h = __builtin_amdgcn_readfirstlane(op->lm_scale_by_h);
    o.cond_pred_corr = __builtin_amdgcn_readfirstlane(op->cond_pred_corr);
    o.mu0 = sreg(op->mu0); o.thr0 = sreg(op->thr0); o.alpha_min = sreg(op->alpha_min);
    o.lam_min = sreg(op->lam_min); o.t_min = sreg(op->t_min);
    o.rti = __builtin_amdgcn_readfirstlane(op->rti);
    o.max_obs = __builtin_amdgcn_readfirstlane(op->max_obs);
    o.globalization = __builtin_amdgcn_readfirstlane(op->globalization);
    o.reserved_ = 0;
    o.ls_shrink = sreg(op->ls_shrink); o.ls_alpha_min = sreg(op->ls_alpha_min);
    o.ls_c1 = sreg(op->ls_c1); o.ls_rho_factor = sreg(op->ls_rho_factor);
    return o;
}

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
// model (reference core_solver.py:148-155) and one RK4 step with the
// non-trivial part of its forward sensitivities
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

// rows 0..2 of K = J*S + dF/du for the columns (theta, v, delta, u0, u1).  Of
// S only the heading row (s23 s24 s25 s26; its theta entry is 1) and the
// entries s35 = dv/du0, s46 = ddelta/du1 enter.
struct KS {
    double k02, k12;                 // column theta (k22 = 0)
    double k03, k13, k23;            // column v
    double k04, k14, k24;            // column delta
    double k05, k15, k25;            // column u0
    double k06, k16, k26;            // column u1
};

DEV void vde_eval(const Jac &J, double s23, double s24, double s25, double s26,
                  double s35, double s46, KS &K)
{
    K.k02 = J.j02;
    K.k12 = J.j12;
    K.k03 = fma(J.j03, 1.0, J.j02 * s23);
    K.k13 = fma(J.j13, 1.0, J.j12 * s23);
    K.k23 = J.j23;
    K.k04 = J.j02 * s24;
    K.k14 = J.j12 * s24;
    K.k24 = J.j24;
    K.k05 = fma(J.j03, s35, J.j02 * s25);
    K.k15 = fma(J.j13, s35, J.j12 * s25);
    K.k25 = J.j23 * s35;
    K.k06 = J.j02 * s26;
    K.k16 = J.j12 * s26;
    K.k26 = J.j24 * s46;
}

DEV void rk4_sens(const Lane &L, const double *x, const double *u, double *xn, AB &ab)
{
    const double h = L.h, hh = L.hh, h6 = L.h6;
    double f[NX], xs[NX], xa[NX];
    Jac J;
    KS K, A;
    // stage 1 at S = [I | 0]
    model_eval(L.invL, x, u, f, J);
    vde_eval(J, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, K);
    A = K;
#pragma unroll
    for (int i = 0; i < NX; ++i) { xa[i] = f[i]; xs[i] = fma(hh, f[i], x[i]); }
    // stage 2 at S = S0 + h/2 K1
    model_eval(L.invL, xs, u, f, J);
    {
        KS Kn;
        vde_eval(J, hh * K.k23, hh * K.k24, hh * K.k25, hh * K.k26, hh, hh, Kn);
        K = Kn;
    }
#define ACC2(m) A.m = fma(2.0, K.m, A.m)
    ACC2(k02); ACC2(k12); ACC2(k03); ACC2(k13); ACC2(k23); ACC2(k04); ACC2(k14); ACC2(k24);
    ACC2(k05); ACC2(k15); ACC2(k25); ACC2(k06); ACC2(k16); ACC2(k26);
#pragma unroll
    for (int i = 0; i < NX; ++i) { xa[i] = fma(2.0, f[i], xa[i]); xs[i] = fma(hh, f[i], x[i]); }
    // stage 3 at S = S0 + h/2 K2
    model_eval(L.invL, xs, u, f, J);
    {
        KS Kn;
        vde_eval(J, hh * K.k23, hh * K.k24, hh * K.k25, hh * K.k26, hh, hh, Kn);
        K = Kn;
    }
    ACC2(k02); ACC2(k12); ACC2(k03); ACC2(k13); ACC2(k23); ACC2(k04); ACC2(k14); ACC2(k24);
    ACC2(k05); ACC2(k15); ACC2(k25); ACC2(k06); ACC2(k16); ACC2(k26);
#undef ACC2
#pragma unroll
    for (int i = 0; i < NX; ++i) { xa[i] = fma(2.0, f[i], xa[i]); xs[i] = fma(h, f[i], x[i]); }
    // stage 4 at S = S0 + h K3
    model_eval(L.invL, xs, u, f, J);
    {
        KS Kn;
        vde_eval(J, h * K.k23, h * K.k24, h * K.k25, h * K.k26, h, h, Kn);
        K = Kn;
    }
#pragma unroll
    for (int i = 0; i < NX; ++i) xn[i] = fma(h6, xa[i] + f[i], x[i]);
    ab.a02 = h6 * (A.k02 + K.k02);
    ab.a12 = h6 * (A.k12 + K.k12);
    ab.a03 = h6 * (A.k03 + K.k03);
    ab.a13 = h6 * (A.k13 + K.k13);
    ab.a23 = h6 * (A.k23 + K.k23);
    ab.a04 = h6 * (A.k04 + K.k04);
    ab.a14 = h6 * (A.k14 + K.k14);
    ab.a24 = h6 * (A.k24 + K.k24);
    ab.b00 = h6 * (A.k05 + K.k05);
    ab.b10 = h6 * (A.k15 + K.k15);
    ab.b20 = h6 * (A.k25 + K.k25);
    ab.b01 = h6 * (A.k06 + K.k06);
    ab.b11 = h6 * (A.k16 + K.k16);
    ab.b21 = h6 * (A.k26 + K.k26);
}

template <int NH, int KIND>
DEV void load_ab(const Lane &L, int k, AB &ab)
{
    typedef Map<NH> M;
    ab.a02 = ldw<KIND>(L, k, M::AB + 0);  ab.a03 = ldw<KIND>(L, k, M::AB + 1);  ab.a04 = ldw<KIND>(L, k, M::AB + 2);
    ab.a12 = ldw<KIND>(L, k, M::AB + 3);  ab.a13 = ldw<KIND>(L, k, M::AB + 4);  ab.a14 = ldw<KIND>(L, k, M::AB + 5);
    ab.a23 = ldw<KIND>(L, k, M::AB + 6);  ab.a24 = ldw<KIND>(L, k, M::AB + 7);
    ab.b00 = ldw<KIND>(L, k, M::AB + 8);  ab.b10 = ldw<KIND>(L, k, M::AB + 9);  ab.b20 = ldw<KIND>(L, k, M::AB + 10);
    ab.b01 = ldw<KIND>(L, k, M::AB + 11); ab.b11 = ldw<KIND>(L, k, M::AB + 12); ab.b21 = ldw<KIND>(L, k, M::AB + 13);
}

template <int NH, int KIND>
DEV void store_ab(const Lane &L, int k, const AB &ab)
{
    typedef Map<NH> M;
    stw<KIND>(L, k, M::AB + 0, ab.a02);  stw<KIND>(L, k, M::AB + 1, ab.a03);  stw<KIND>(L, k, M::AB + 2, ab.a04);
    stw<KIND>(L, k, M::AB + 3, ab.a12);  stw<KIND>(L, k, M::AB + 4, ab.a13);  stw<KIND>(L, k, M::AB + 5, ab.a14);
    stw<KIND>(L, k, M::AB + 6, ab.a23);  stw<KIND>(L, k, M::AB + 7, ab.a24);
    stw<KIND>(L, k, M::AB + 8, ab.b00);  stw<KIND>(L, k, M::AB + 9, ab.b10);  stw<KIND>(L, k, M::AB + 10, ab.b20);
    stw<KIND>(L, k, M::AB + 11, ab.b01); stw<KIND>(L, k, M::AB + 12, ab.b11); stw<KIND>(L, k, M::AB + 13, ab.b21);
}

// acc[c] += sum_i BA[i][c] * w[i], c = 0..6 ([B A] columns), i ascending
DEV void add_BAt(const Lane &L, const AB &ab, const double *w, double *acc)
{
    // B column 0 = [b00 b10 b20 h* 0], column 1 = [b01 b11 b21 0 h*]
    acc[0] = fma(L.hB, w[3], fma(ab.b20, w[2], fma(ab.b10, w[1], fma(ab.b00, w[0], acc[0]))));
    acc[1] = fma(L.hB, w[4], fma(ab.b21, w[2], fma(ab.b11, w[1], fma(ab.b01, w[0], acc[1]))));
    // A columns: e0, e1, [a02 a12 1 0 0], [a03 a13 a23 1 0], [a04 a14 a24 0 1]
    acc[2] = acc[2] + w[0];
    acc[3] = acc[3] + w[1];
    acc[4] = fma(ab.a12, w[1], fma(ab.a02, w[0], acc[4])) + w[2];
    acc[5] = fma(ab.a23, w[2], fma(ab.a13, w[1], fma(ab.a03, w[0], acc[5]))) + w[3];
    acc[6] = fma(ab.a24, w[2], fma(ab.a14, w[1], fma(ab.a04, w[0], acc[6]))) + w[4];
}

// out[i] = acc0[i] + A[i][:] * dx + B[i][:] * du  (row-wise chains, j then c ascending)
DEV void mul_AB(const Lane &L, const AB &ab, const double *dx, const double *du, const double *acc0,
                double *out)
{
    out[0] = fma(ab.b01, du[1], fma(ab.b00, du[0],
             fma(ab.a04, dx[4], fma(ab.a03, dx[3], fma(ab.a02, dx[2], acc0[0] + dx[0])))));
    out[1] = fma(ab.b11, du[1], fma(ab.b10, du[0],
             fma(ab.a14, dx[4], fma(ab.a13, dx[3], fma(ab.a12, dx[2], acc0[1] + dx[1])))));
    out[2] = fma(ab.b21, du[1], fma(ab.b20, du[0],
             fma(ab.a24, dx[4], fma(ab.a23, dx[3], acc0[2] + dx[2]))));
    out[3] = fma(L.hB, du[0], acc0[3] + dx[3]);
    out[4] = fma(L.hB, du[1], acc0[4] + dx[4]);
}

// ---------------------------------------------------------------------------
// initial state guess of an obstacle-free problem at node i (reference core_solver.py:477-499):
// the start-goal interpolation at t = i/N, y clamped to the reference's centre-line formula
// +- width/2, heading interpolated, v = 1 - t, delta = 0; node N is the goal state.  The same
// operations in the same order as the reference's loop, so a host program gets the same bits.
// ---------------------------------------------------------------------------
struct GuessData { double s[NX], g[NX], a, b1, b2, half; int nobs; double ox[MPCX_MAX_OBS], oy[MPCX_MAX_OBS], tr[MPCX_MAX_OBS]; };

// problem b's column of the table; width = the caller's column, or NULL = the table's own;
// nobs = the problem's obstacle count (0: the obstacle columns are not read)
DEV void guess_data(const double *pdata, const double *width, size_t Bs, size_t b, int nobs, GuessData &G)
{
#pragma unroll
    for (int i = 0; i < NX; ++i) {
        G.s[i] = pdata[(size_t)(MPCX_F_X0 + i) * Bs + b];
        G.g[i] = pdata[(size_t)(MPCX_F_GOAL + i) * Bs + b];
    }
    // corridor rows are stored as (-a1, -c1, a2, c2) and (b1, -b2)
    G.a = pdata[(size_t)(MPCX_F_CG + 2) * Bs + b];
    G.b1 = pdata[(size_t)(MPCX_F_UG + 0) * Bs + b];
    G.b2 = -pdata[(size_t)(MPCX_F_UG + 1) * Bs + b];
    G.half = (width ? width[b] : pdata[(size_t)MPCX_F_CW * Bs + b]) / 2;
    G.nobs = nobs;
#pragma unroll
    for (int j = 0; j < MPCX_MAX_OBS; ++j) {
        G.ox[j] = G.oy[j] = 0.0;
        G.tr[j] = -1.0;
        if (j < nobs) {
            G.ox[j] = pdata[(size_t)(MPCX_F_OBS + 3 * j) * Bs + b];
            G.oy[j] = pdata[(size_t)(MPCX_F_OBS + 3 * j + 1) * Bs + b];
            // the table holds (r + margin)^2 as the product of the rounded sum with itself, whose
            // correctly rounded root is that sum again; the guess keeps 0.05 m more (core_solver.py:393)
            G.tr[j] = sqrt(pdata[(size_t)(MPCX_F_OBS + 3 * j + 2) * Bs + b]) + 0.05;
        }
    }
}

// A guess point moved off the obstacles (reference core_solver.py:360-473): if an obstacle straddles
// the start-goal line the whole line is shifted sideways by the largest intrusion + 0.1 m (to the side
// of the point; the side sticks once negative) and the point projected onto the shifted segment;
// otherwise a point inside an obstacle is pushed out radially to its inflated radius, obstacle after
// obstacle.  Two places are not the reference's arithmetic to the last bit: its squares are pow(x, 2)
// where these are products, and its radial direction is (cos, sin) of arctan2(dy, dx) where this is
// (dx, dy) / hypotenuse (the same unit vector, free of the host's libm).
DEV void guess_adjust(const GuessData &G, double &x, double &y)
{
    if (G.nobs <= 0) return;
    const double sx = G.s[0], sy = G.s[1];
    double pdx = G.g[0] - sx, pdy = G.g[1] - sy;
    const double plen = sqrt(pdx * pdx + pdy * pdy);
    if (plen < 1e-6) return;
    pdx = pdx / plen;
    pdy = pdy / plen;
    const double ndx = -pdy, ndy = pdx;
    double path_shift = 0.0, dir = 1.0;
#pragma unroll
    for (int j = 0; j < MPCX_MAX_OBS; ++j) {
        if (j >= G.nobs) continue;
        const double odx = G.ox[j] - sx, ody = G.oy[j] - sy;
        const double projection = odx * pdx + ody * pdy;
        const double cx = sx + projection * pdx, cy = sy + projection * pdy;
        const double ex = cx - G.ox[j], ey = cy - G.oy[j];
        const double dist = sqrt(ex * ex + ey * ey);
        if (dist < G.tr[j] && 0.0 <= projection && projection <= plen) {
            const double side = (x - sx) * ndx + (y - sy) * ndy;
            if (side < 0.0) dir = -1.0;
            const double needed = G.tr[j] - dist;
            if (needed > path_shift) path_shift = needed;
        }
    }
    if (path_shift > 0.0) {
        const double amount = path_shift + 0.1;
        const double ssx = sx + dir * ndx * amount, ssy = sy + dir * ndy * amount;
        double proj = (x - ssx) * pdx + (y - ssy) * pdy;
        proj = (proj < plen) ? proj : plen;
        proj = (proj > 0.0) ? proj : 0.0;
        x = ssx + proj * pdx;
        y = ssy + proj * pdy;
        return;
    }
#pragma unroll
    for (int j = 0; j < MPCX_MAX_OBS; ++j) {
        if (j >= G.nobs) continue;
        const double dx = x - G.ox[j], dy = y - G.oy[j];
        const double d = sqrt(dx * dx + dy * dy);
        if (d < G.tr[j]) {
            const double ux = (d > 0.0) ? dx / d : 1.0, uy = (d > 0.0) ? dy / d : 0.0;
            x = G.ox[j] + G.tr[j] * ux;
            y = G.oy[j] + G.tr[j] * uy;
        }
    }
}

DEV void guess_node(const GuessData &G, int i, int N, double *o)
{
    if (i >= N) {
#pragma unroll
        for (int c = 0; c < NX; ++c) o[c] = G.g[c];
        return;
    }
    const double t = (double)i / (double)N;
    double x = G.s[0] + t * (G.g[0] - G.s[0]);
    double y = G.s[1] + t * (G.g[1] - G.s[1]);
    const double yc = -(((G.a * x + G.b1 / 2) + G.a * x) + G.b2 / 2) / 2;
    const double lo = yc - G.half, hi = yc + G.half;
    y = (y > lo) ? y : lo;
    y = (y < hi) ? y : hi;
    guess_adjust(G, x, y);
    o[0] = x;
    o[1] = y;
    o[2] = G.s[2] + t * (G.g[2] - G.s[2]);
    o[3] = 1.0 * (1 - t);
    o[4] = 0.0;
}

// ---------------------------------------------------------------------------
// rows of a stage.  r: 0-1 input box, 2-6 state box, 7-8 corridor, 9.. obstacles.
// Stage 0 has the input rows only, stage N the state and corridor rows only,
// the stages between have all of them (obstacle row j only if j < nobs).
// The per-stage code is instantiated once per KIND so that the set of rows is
// known at compile time and a stage is ONE basic block: all its loads can be
// issued before the first one is needed.
// ---------------------------------------------------------------------------
// ANY: the kind is a run-time property of the lane (single-problem mode, where the lanes of a
// wavefront work on different stages).  The code of a stage is then the code of a middle
// stage, every row and variable computed, and what the stage does not have is kept out of the
// results by selects; for the three fixed kinds the same selects fold away at compile time.

struct KR { bool first, last; };

template <int KIND> DEV constexpr bool row_present(int r)      // can a stage of this kind have row r
{
    return (KIND == ANY) ? true : (r < NU) ? (KIND != LAST) : (r < 9) ? (KIND != FIRST) : (KIND == MID || KIND == MIDC);
}
template <int KIND> DEV constexpr bool var_present(int j)      // ... have v[j] as a variable
{
    return (KIND == ANY) ? true : (j < NU) ? (KIND != LAST) : (KIND != FIRST);
}
template <int KIND> DEV bool is_first(const KR &kr) { return (KIND == ANY) ? kr.first : (KIND == FIRST); }
template <int KIND> DEV bool is_last(const KR &kr) { return (KIND == ANY) ? kr.last : (KIND == LAST); }
template <int KIND> DEV bool row_on(const KR &kr, int r)       // does this lane's stage have row r
{
    if (KIND != ANY) return row_present<KIND>(r);
    return (r < NU) ? !kr.last : (r < 9) ? !kr.first : !(kr.first || kr.last);
}
template <int KIND> DEV bool var_on(const KR &kr, int j)
{
    if (KIND != ANY) return var_present<KIND>(j);
    return (j < NU) ? !kr.last : !kr.first;
}

// Per-lane problem constants live in LDS as [field][lane] (one ds_read_b64 per
// use, conflict-free), written when the lane takes a problem.  Keeping them out
// of the vector registers is what lets a whole stage's global loads be in
// flight at once without spilling.
enum {
    C_LBU = 0, C_UBU = 2, C_LBX = 4, C_UBX = 9, C_CG = 14, C_LG = 18, C_UG = 20, C_LH = 22, C_UH = 23,
    C_GOAL = 24, C_W = 29, C_HW = 36, C_LBXE = 43, C_UBXE = 48, C_WE = 53, C_X0 = 58, C_OBS = 63
};
template <int NH> struct PCN { static constexpr int N = C_OBS + 3 * NH; };

template <int NH>
struct PC {
    ldouble *p;         // LDS, already offset by the lane
    double lmk;
    DEV double get(int f) const { return p[f * 64]; }
};

template <int NH>
DEV void fill_pc(const Lane &L, const mpcx_opts &o, PC<NH> &pc)
{
    auto cp = [&](int dst, int src, int n) {
        for (int i = 0; i < n; ++i) pc.p[(dst + i) * 64] = ldp(L, src + i);
    };
    cp(C_LBU, MPCX_F_LBU, 2);   cp(C_UBU, MPCX_F_UBU, 2);
    cp(C_LBX, MPCX_F_LBX, 5);   cp(C_UBX, MPCX_F_UBX, 5);
    cp(C_CG, MPCX_F_CG, 4);     cp(C_LG, MPCX_F_LG, 2);     cp(C_UG, MPCX_F_UG, 2);
    cp(C_LH, MPCX_F_LH, 1);     cp(C_UH, MPCX_F_UH, 1);
    cp(C_GOAL, MPCX_F_GOAL, 5); cp(C_W, MPCX_F_W, 7);
    cp(C_LBXE, MPCX_F_LBXE, 5); cp(C_UBXE, MPCX_F_UBXE, 5); cp(C_WE, MPCX_F_WE, 5);
    cp(C_X0, MPCX_F_X0, 5);     cp(C_OBS, MPCX_F_OBS, 3 * NH);
    for (int j = 0; j < NZ; ++j) pc.p[(C_HW + j) * 64] = L.h * ldp(L, MPCX_F_W + j);
    pc.lmk = o.lm_scale_by_h ? L.h * o.lm : o.lm;
}

// bounds of row r (terminal state box for KIND == LAST)
template <int NH, int KIND>
DEV void row_bounds(const Lane &L, const PC<NH> &pc, const KR &kr, int r, double &lo, double &up)
{
    if (r < NU) { lo = pc.get(C_LBU + r); up = pc.get(C_UBU + r); }
    else if (r < ROW_G0) {
        if (KIND == LAST) { lo = pc.get(C_LBXE + r - NU); up = pc.get(C_UBXE + r - NU); }
        else if (KIND == ANY) {
            const double le = pc.get(C_LBXE + r - NU), ue = pc.get(C_UBXE + r - NU);
            const double ls = pc.get(C_LBX + r - NU), us = pc.get(C_UBX + r - NU);
            lo = kr.last ? le : ls;
            up = kr.last ? ue : us;
        }
        else { lo = pc.get(C_LBX + r - NU); up = pc.get(C_UBX + r - NU); }
    }
    else if (r < 9) { lo = pc.get(C_LG + r - ROW_G0); up = pc.get(C_UG + r - ROW_G0); }
    else { lo = pc.get(C_LH); up = pc.get(C_UH); }
}

// value c of row r at (x, u) and, for general rows, its Jacobian (jx, jy) w.r.t. (x, y)
template <int NH>
DEV void row_geom(const PC<NH> &pc, int r, const double *x, const double *u,
                  double &c, double &jx, double &jy)
{
    jx = 0.0;
    jy = 0.0;
    if (r < NU) c = u[r];
    else if (r < ROW_G0) c = x[r - NU];
    else if (r < 9) {
        jx = pc.get(C_CG + 2 * (r - ROW_G0));
        jy = pc.get(C_CG + 2 * (r - ROW_G0) + 1);
        c = fma(jy, x[1], jx * x[0]);
    } else {
        const int j = (r - 9 < NH) ? r - 9 : 0;
        double dx = x[0] - pc.get(C_OBS + 3 * j), dy = x[1] - pc.get(C_OBS + 3 * j + 1);
        c = fma(dy, dy, dx * dx) - pc.get(C_OBS + 3 * j + 2);
        jx = 2.0 * dx;
        jy = 2.0 * dy;
    }
}

// J_r * v for v = [u; x]
DEV double jdot(int r, double jx, double jy, const double *v)
{
    if (r < ROW_G0) return v[r];
    return fma(jy, v[3], jx * v[2]);
}

// cost gradient g and Gauss-Newton + LM diagonal Hd of a stage at (x, u)
template <int NH, int KIND>
DEV void cost_terms(const Lane &L, const mpcx_opts &o, const PC<NH> &pc, const KR &kr, const double *x,
                    const double *u, double *g, double *Hd)
{
    if (KIND != LAST) {
#pragma unroll
        for (int j = 0; j < NZ; ++j) {
            double z = (j < NU) ? u[j] : (x[j - NU] - pc.get(C_GOAL + j - NU));
            const double hw = pc.get(C_HW + j);
            Hd[j] = hw + pc.lmk;
            g[j] = hw * z;
        }
    }
    if (KIND == LAST) {
        Hd[0] = Hd[1] = 0.0;
        g[0] = g[1] = 0.0;
#pragma unroll
        for (int i = 0; i < NX; ++i) {
            double we = pc.get(C_WE + i);
            Hd[NU + i] = we + o.lm;
            g[NU + i] = we * (x[i] - pc.get(C_GOAL + i));
        }
    }
    if (KIND == ANY) {
        const bool e = kr.last;
        Hd[0] = e ? 0.0 : Hd[0];
        Hd[1] = e ? 0.0 : Hd[1];
        g[0] = e ? 0.0 : g[0];
        g[1] = e ? 0.0 : g[1];
#pragma unroll
        for (int i = 0; i < NX; ++i) {
            double we = pc.get(C_WE + i);
            const double he = we + o.lm, ge = we * (x[i] - pc.get(C_GOAL + i));
            Hd[NU + i] = e ? he : Hd[NU + i];
            g[NU + i] = e ? ge : g[NU + i];
        }
    }
}

template <int NH, int KIND>
DEV void load_xu(const Lane &L, int k, const KR &kr, double *x, double *u)
{
    typedef Map<NH> M;
#pragma unroll
    for (int i = 0; i < NX; ++i) x[i] = ldw<KIND>(L, k, M::X + i);
    u[0] = u[1] = 0.0;
    if (KIND != LAST) {
        const double u0 = ldw<KIND>(L, k, M::U + 0), u1 = ldw<KIND>(L, k, M::U + 1);
        u[0] = is_last<KIND>(kr) ? 0.0 : u0;
        u[1] = is_last<KIND>(kr) ? 0.0 : u1;
    }
}

// complementarity targets of the two sides of a row for the three kinds of step
//   mode 0 affine: lam*t;  mode 1 corrector: lam*t + dt_a*dlam_a - sigma*mu;
//   mode 2 centring: lam*t - sigma*mu            (jda = J_r * affine step)
// All three are computed and the lane's `mode` selects, so lanes in different
// modes share one instruction stream.
DEV void row_rm(int mode, double sigma_mu, double jda, double ll, double lu, double tl, double tu,
                double til, double tiu, double rdl, double rdu, double &rml, double &rmu)
{
    const double ml = ll * tl, mu_ = lu * tu;
    const double dtl = jda - rdl;
    const double dtu = -jda - rdu;
    const double dll = -(fma(ll, dtl, ml) * til);
    const double dlu = -(fma(lu, dtu, mu_) * tiu);
    const double rml1 = fma(dtl, dll, ml) - sigma_mu;
    const double rmu1 = fma(dtu, dlu, mu_) - sigma_mu;
    const double rml2 = ml - sigma_mu;
    const double rmu2 = mu_ - sigma_mu;
    rml = (mode == 1) ? rml1 : (mode == 2) ? rml2 : ml;
    rmu = (mode == 1) ? rmu1 : (mode == 2) ? rmu2 : mu_;
}

// ---------------------------------------------------------------------------
// sweep L (forward): SQP step, linearisation, NLP residuals, cold start
// ---------------------------------------------------------------------------
struct LAcc { double rs, re, ri, rc, cost; };

// nxt (single-problem mode only): state, step and multiplier of stage k+1, read by the caller
// before any lane has overwritten them; otherwise they are loaded here
// first: the problem's first linearisation.  Its multipliers are zero by definition and are not read from
// the workspace (nothing has been written there yet); zero_u: nor are its inputs (no input guess given)
// alpha: length of the SQP step (1 unless the line search shortened it; 1 * v is v exactly)
template <int NH, int KIND, bool COOP = false>
DEV void stage_L(const Lane &L, const mpcx_opts &o, const PC<NH> &pc, int k, bool first, bool zero_u, double alpha,
                 double *xk, double *pik, LAcc &a, const KR &kr = KR{false, false}, const double *nxt = nullptr)
{
    typedef Map<NH> M;
    const double thr0 = o.thr0, mu0 = o.mu0;
    // loads
    double uk[NU] = {0.0, 0.0}, vu[NU] = {0.0, 0.0};
    double xn1[NX], vxn1[NX], pin1[NX];
    double ll[M::R], lu[M::R];
#pragma unroll
    for (int i = 0; i < NX; ++i) { xn1[i] = 0.0; vxn1[i] = 0.0; pin1[i] = 0.0; }
    if (KIND != LAST) {
#pragma unroll
        for (int c = 0; c < NU; ++c) { uk[c] = ldw<KIND>(L, k, M::U + c); vu[c] = ldw<KIND>(L, k, M::V + c); }
#pragma unroll
        for (int i = 0; i < NX; ++i) {
            xn1[i] = COOP ? nxt[i] : ldw<KIND>(L, k + 1, M::X + i);
            vxn1[i] = COOP ? nxt[NX + i] : ldw<KIND>(L, k + 1, M::V + NU + i);
            pin1[i] = COOP ? nxt[2 * NX + i] : ldw<KIND>(L, k + 1, M::QPI + i);
        }
    }
#pragma unroll
    for (int r = 0; r < M::R; ++r) {
        ll[r] = 0.0; lu[r] = 0.0;
        if (!row_present<KIND>(r)) continue;
        ll[r] = ldw<KIND>(L, k, M::LAML + r);
        lu[r] = ldw<KIND>(L, k, M::LAMU + r);
    }
    __builtin_amdgcn_sched_barrier(0);
    if (KIND != LAST) {
#pragma unroll
        for (int c = 0; c < NU; ++c) uk[c] = (first && zero_u) ? 0.0 : uk[c];
#pragma unroll
        for (int i = 0; i < NX; ++i) pin1[i] = first ? 0.0 : pin1[i];
    }
#pragma unroll
    for (int r = 0; r < M::R; ++r) { ll[r] = first ? 0.0 : ll[r]; lu[r] = first ? 0.0 : lu[r]; }
    // SQP step (nothing to add before the first QP; v is uninitialised then)
    AB ab = {};
    if (KIND != LAST) {
#pragma unroll
        for (int c = 0; c < NU; ++c) {
            uk[c] = first ? uk[c] : uk[c] + alpha * vu[c];
            uk[c] = is_last<KIND>(kr) ? 0.0 : uk[c];
            stw<KIND>(L, k, M::U + c, uk[c]);
        }
#pragma unroll
        for (int i = 0; i < NX; ++i) xn1[i] = first ? xn1[i] : xn1[i] + alpha * vxn1[i];
        double xn[NX];
        rk4_sens(L, xk, uk, xn, ab);
        store_ab<NH, KIND>(L, k, ab);
#pragma unroll
        for (int i = 0; i < NX; ++i) {
            double bv = xn[i] - xn1[i];
            stw<KIND>(L, k, M::BV + i, bv);
            a.re = is_last<KIND>(kr) ? a.re : dmax(a.re, dabs(bv));
        }
    }
#pragma unroll
    for (int i = 0; i < NX; ++i) stw<KIND>(L, k, M::X + i, xk[i]);
    // stationarity of the Lagrangian
    double g[NZ], Hd[NZ], gl[NZ];
    cost_terms<NH, KIND>(L, o, pc, kr, xk, uk, g, Hd);
#pragma unroll
    for (int j = 0; j < NZ; ++j) gl[j] = g[j];
    if (KIND != LAST) {
        double gn[NZ];
#pragma unroll
        for (int j = 0; j < NZ; ++j) gn[j] = gl[j];
        add_BAt(L, ab, pin1, gn);
#pragma unroll
        for (int j = 0; j < NZ; ++j) gl[j] = is_last<KIND>(kr) ? gl[j] : gn[j];
    }
    if (KIND != FIRST) {
#pragma unroll
        for (int i = 0; i < NX; ++i) gl[NU + i] = is_first<KIND>(kr) ? gl[NU + i] : gl[NU + i] - pik[i];
    }
    double vq[NZ];
#pragma unroll
    for (int j = 0; j < NZ; ++j) vq[j] = 0.0;
    if (KIND == FIRST || KIND == ANY) {
#pragma unroll
        for (int i = 0; i < NX; ++i) {
            double d0 = pc.get(C_X0 + i) - xk[i];
            vq[NU + i] = is_first<KIND>(kr) ? d0 : 0.0;
            a.ri = is_first<KIND>(kr) ? dmax(a.ri, dabs(d0)) : a.ri;
        }
    }
    // rows: the multipliers of the last QP enter the residuals, then the cold start replaces them
#pragma unroll
    for (int r = 0; r < M::R; ++r) {
        if (!row_present<KIND>(r)) continue;
        const bool pr = row_on<KIND>(kr, r) && ((r < 9) || (r - 9 < L.nobs));
        double c, jx, jy, lo, up;
        row_geom<NH>(pc, r, xk, uk, c, jx, jy);
        row_bounds<NH, KIND>(L, pc, kr, r, lo, up);
        double dl = lu[r] - ll[r];
        if (r < ROW_G0) gl[r] = pr ? gl[r] + dl : gl[r];
        else {
            gl[2] = pr ? fma(jx, dl, gl[2]) : gl[2];
            gl[3] = pr ? fma(jy, dl, gl[3]) : gl[3];
        }
        double sl = c - lo, su = up - c;
        double ri = dmax(dmax(a.ri, -sl), -su);
        double rc = dmax(dmax(a.rc, dabs(ll[r] * sl)), dabs(lu[r] * su));
        a.ri = pr ? ri : a.ri;
        a.rc = pr ? rc : a.rc;
        // cold start of the interior point for this row
        double dlo = lo - c, dup = up - c, tl, tu;
        if (r < ROW_G0) {
            tl = vq[r] - dlo;
            tu = dup - vq[r];
            const bool bl = tl < thr0, bu = tu < thr0;
            const double vn = bl ? (bu ? 0.5 * (dlo + dup) : dlo + thr0) : (bu ? dup - thr0 : vq[r]);
            vq[r] = pr ? vn : vq[r];
            tu = bu ? thr0 : tu;
            tl = bl ? thr0 : tl;
        } else {
            double jv = jdot(r, jx, jy, vq);
            tl = jv - dlo;
            tu = dup - jv;
            tl = (tl < thr0) ? thr0 : tl;
            tu = (tu < thr0) ? thr0 : tu;
        }
        stw<KIND>(L, k, M::TL + r, tl);
        stw<KIND>(L, k, M::TU + r, tu);
        stw<KIND>(L, k, M::LAML + r, mu0 / tl);
        stw<KIND>(L, k, M::LAMU + r, mu0 / tu);
        if (MPCX_STORE_RECIP) {
            stw<KIND>(L, k, M::TIL + r, 1.0 / tl);
            stw<KIND>(L, k, M::TIU + r, 1.0 / tu);
        }
    }
#pragma unroll
    for (int j = 0; j < NZ; ++j) {
        if (var_present<KIND>(j)) a.rs = var_on<KIND>(kr, j) ? dmax(a.rs, dabs(gl[j])) : a.rs;
        stw<KIND>(L, k, M::V + j, vq[j]);
    }
#pragma unroll
    for (int i = 0; i < NX; ++i) stw<KIND>(L, k, M::QPI + i, 0.0);
    // objective: summed per stage, then over the stages
    double ck = 0.0, cke = 0.0;
    if (KIND != LAST) {
#pragma unroll
        for (int j = 0; j < NZ; ++j) {
            double z = (j < NU) ? uk[j] : (xk[j - NU] - pc.get(C_GOAL + j - NU));
            double hw = (0.5 * L.h) * pc.get(C_W + j);
            ck = fma(hw * z, z, ck);
        }
    }
    if (KIND == LAST || KIND == ANY) {
#pragma unroll
        for (int i = 0; i < NX; ++i) {
            double z = xk[i] - pc.get(C_GOAL + i);
            cke = fma((0.5 * pc.get(C_WE + i)) * z, z, cke);
        }
    }
    a.cost = a.cost + (is_last<KIND>(kr) ? cke : ck);
#pragma unroll
    for (int i = 0; i < NX; ++i) {
        xk[i] = xn1[i];
        pik[i] = pin1[i];
    }
}

// The sweeps are separate (non-inlined) functions so that each gets its own register
// allocation; arguments travel by value in registers and results come back through a
// small per-lane LDS area `ret` ([slot][lane] doubles), never through the stack.
template <int NH>
__device__ __attribute__((noinline)) void sweep_L(Lane L, PC<NH> pc, gopts *op, bool first, bool zero_u, double alpha,
                                                  ldouble *ret)
{
    const mpcx_opts ol = opts_in_registers(op);
    const mpcx_opts &o = ol;
    typedef Map<NH> M;
    const int N = L.N;
    LAcc a = {0.0, 0.0, 0.0, 0.0, 0.0};
    double xk[NX], pik[NX];
#pragma unroll
    for (int i = 0; i < NX; ++i) {
        double x0 = ldw<MID>(L, 0, M::X + i), v0 = ldw<MID>(L, 0, M::V + NU + i);
        xk[i] = first ? x0 : x0 + alpha * v0;
        pik[i] = 0.0;
    }
    stage_L<NH, FIRST>(L, o, pc, 0, first, zero_u, alpha, xk, pik, a);
    {
        int k = 1;
        for (; k < N && k <= MPCX_EDGE_STAGES; ++k) stage_L<NH, MIDC>(L, o, pc, k, first, zero_u, alpha, xk, pik, a);
        for (; k < N - MPCX_EDGE_STAGES; ++k) stage_L<NH, MID>(L, o, pc, k, first, zero_u, alpha, xk, pik, a);
        for (; k < N; ++k) stage_L<NH, MIDC>(L, o, pc, k, first, zero_u, alpha, xk, pik, a);
    }
    stage_L<NH, LAST>(L, o, pc, N, first, zero_u, alpha, xk, pik, a);
    ret[0 * 64] = a.rs; ret[1 * 64] = a.re; ret[2 * 64] = a.ri; ret[3 * 64] = a.rc;
    ret[4 * 64] = a.cost;
}

// ---------------------------------------------------------------------------
// sweep M (forward, reads only): the line search's look at w + alpha * d, d = the primal solution
// of the QP just solved (mpcx_opts.globalization = 1): objective, largest dynamics gap, largest
// violation of an inequality row (and of the initial state), and the sum of the absolute values of
// the QP's multipliers.  Sums per stage first, then over the stages in ascending order.
// ---------------------------------------------------------------------------
struct MAcc { double cost, gap, viol, msum; };

template <int NH, int KIND, bool COOP = false>
DEV void stage_M(const Lane &L, const PC<NH> &pc, int k, double alpha, double *xk, MAcc &a,
                 const KR &kr = KR{false, false}, const double *nxt = nullptr, const double *qk = nullptr)
{
    typedef Map<NH> M;
    double uk[NU] = {0.0, 0.0}, xn1[NX], qpi[NX], ll[M::R], lu[M::R];
#pragma unroll
    for (int i = 0; i < NX; ++i) { xn1[i] = 0.0; qpi[i] = 0.0; }
    if (KIND != LAST) {
#pragma unroll
        for (int c = 0; c < NU; ++c) {
            const double u0 = ldw<KIND>(L, k, M::U + c), vu = ldw<KIND>(L, k, M::V + c);
            uk[c] = is_last<KIND>(kr) ? 0.0 : u0 + alpha * vu;
        }
#pragma unroll
        for (int i = 0; i < NX; ++i) {
            const double x1 = COOP ? nxt[i] : ldw<KIND>(L, k + 1, M::X + i);
            const double v1 = COOP ? nxt[NX + i] : ldw<KIND>(L, k + 1, M::V + NU + i);
            xn1[i] = x1 + alpha * v1;
        }
    }
    if (KIND != FIRST) {
#pragma unroll
        for (int i = 0; i < NX; ++i) qpi[i] = COOP ? qk[i] : ldw<KIND>(L, k, M::QPI + i);
    }
#pragma unroll
    for (int r = 0; r < M::R; ++r) {
        ll[r] = 0.0; lu[r] = 0.0;
        if (!row_present<KIND>(r)) continue;
        ll[r] = ldw<KIND>(L, k, M::LAML + r);
        lu[r] = ldw<KIND>(L, k, M::LAMU + r);
    }
    __builtin_amdgcn_sched_barrier(0);
    double ck = 0.0, cke = 0.0;
    if (KIND != LAST) {
        double xn[NX];
        AB ab = {};
        rk4_sens(L, xk, uk, xn, ab);
#pragma unroll
        for (int i = 0; i < NX; ++i) a.gap = is_last<KIND>(kr) ? a.gap : dmax(a.gap, dabs(xn[i] - xn1[i]));
#pragma unroll
        for (int j = 0; j < NZ; ++j) {
            double z = (j < NU) ? uk[j] : (xk[j - NU] - pc.get(C_GOAL + j - NU));
            double hw = (0.5 * L.h) * pc.get(C_W + j);
            ck = fma(hw * z, z, ck);
        }
    }
    if (KIND == LAST || KIND == ANY) {
#pragma unroll
        for (int i = 0; i < NX; ++i) {
            double z = xk[i] - pc.get(C_GOAL + i);
            cke = fma((0.5 * pc.get(C_WE + i)) * z, z, cke);
        }
    }
    a.cost = a.cost + (is_last<KIND>(kr) ? cke : ck);
    if (KIND == FIRST || KIND == ANY) {
#pragma unroll
        for (int i = 0; i < NX; ++i) {
            const double d0 = dabs(pc.get(C_X0 + i) - xk[i]);
            a.viol = is_first<KIND>(kr) ? dmax(a.viol, d0) : a.viol;
        }
    }
    double sk = 0.0;
    if (KIND != FIRST) {
#pragma unroll
        for (int i = 0; i < NX; ++i) sk = is_first<KIND>(kr) ? sk : sk + dabs(qpi[i]);
    }
#pragma unroll
    for (int r = 0; r < M::R; ++r) {
        if (!row_present<KIND>(r)) continue;
        const bool pr = row_on<KIND>(kr, r) && ((r < 9) || (r - 9 < L.nobs));
        double c, jx, jy, lo, up;
        row_geom<NH>(pc, r, xk, uk, c, jx, jy);
        row_bounds<NH, KIND>(L, pc, kr, r, lo, up);
        const double vv = dmax(dmax(a.viol, lo - c), c - up);
        a.viol = pr ? vv : a.viol;
        const double s2 = (sk + dabs(ll[r])) + dabs(lu[r]);
        sk = pr ? s2 : sk;
    }
    a.msum = a.msum + sk;
#pragma unroll
    for (int i = 0; i < NX; ++i) xk[i] = xn1[i];
}

template <int NH>
__device__ __attribute__((noinline)) void sweep_M(Lane L, PC<NH> pc, double alpha, ldouble *ret)
{
    typedef Map<NH> M;
    const int N = L.N;
    MAcc a = {0.0, 0.0, 0.0, 0.0};
    double xk[NX];
#pragma unroll
    for (int i = 0; i < NX; ++i) {
        const double x0 = ldw<MID>(L, 0, M::X + i), v0 = ldw<MID>(L, 0, M::V + NU + i);
        xk[i] = x0 + alpha * v0;
    }
    stage_M<NH, FIRST>(L, pc, 0, alpha, xk, a);
    for (int k = 1; k < N; ++k) stage_M<NH, MID>(L, pc, k, alpha, xk, a);
    stage_M<NH, LAST>(L, pc, N, alpha, xk, a);
    ret[0 * 64] = a.cost; ret[1 * 64] = a.gap; ret[2 * 64] = a.viol; ret[3 * 64] = a.msum;
}

// ---------------------------------------------------------------------------
// backward sweeps
// ---------------------------------------------------------------------------
struct QpRes { double g, b, d, m, mu; };

struct Step {          // the interior-point step waiting to be applied
    bool have;
    int mode;
    double sigma_mu, alpha;
};

// P(k+1) lower triangle, packed row-wise: index of (i, j), i >= j
DEV constexpr int tri(int i, int j) { return i * (i + 1) / 2 + j; }
DEV double psym(const double *P, int i, int j) { return (i >= j) ? P[tri(i, j)] : P[tri(j, i)]; }

// what the row part of a B0 stage hands to the Riccati part (and, for vx / qpi, to stage k-1)
struct B0Out {
    double gt[NZ], rg[NZ], Ht[NZ], hxx, hxy, hyy, rb[NX];
    AB ab;
    double vx[NX], qpi[NX];
    double ng, nb, nd, nm, ssum;
};

// everything B0 reads of a stage, and the stage's primal iterate after the pending step
template <int NH>
struct B0In {
    double x[NX], u[NU], v[NZ], vo[NZ], dv[NZ], dva[NZ], qpi[NX];
    double rgo[NX];      // x-part of the stationarity residual the iteration before left behind
    double ll[Map<NH>::R], lu[Map<NH>::R], tl[Map<NH>::R], tu[Map<NH>::R], til_[Map<NH>::R], tiu_[Map<NH>::R];
    AB ab;
    double bv[NX];
};

// first half: all loads of the stage
template <int NH, int KIND>
DEV void stage_B0_load(const Lane &L, int k, const Step &sp, B0In<NH> &in, const KR &kr = KR{false, false})
{
    typedef Map<NH> M;
    // ---- loads ----
    const int sol = (sp.mode == 2) ? M::SOL2 : 0;      // which of the two stored solutions is the step
    double (&x)[NX] = in.x, (&u)[NU] = in.u, (&vo)[NZ] = in.vo, (&dv)[NZ] = in.dv,
           (&dva)[NZ] = in.dva, (&qpi)[NX] = in.qpi;
    double (&ll)[M::R] = in.ll, (&lu)[M::R] = in.lu, (&tl)[M::R] = in.tl, (&tu)[M::R] = in.tu,
           (&til_)[M::R] = in.til_, (&tiu_)[M::R] = in.tiu_;
    load_xu<NH, KIND>(L, k, kr, x, u);
#pragma unroll
    for (int j = 0; j < NZ; ++j) {
        vo[j] = ldw<KIND>(L, k, M::V + j);
        dv[j] = ldw_lane<KIND>(L, k, M::DV + j, sol);
        dva[j] = ldw<KIND>(L, k, M::DVA + j);
    }
#pragma unroll
    for (int i = 0; i < NX; ++i) {
        qpi[i] = (KIND != FIRST) ? ldw<KIND>(L, k, M::QPI + i) : 0.0;
        in.rgo[i] = (KIND != FIRST) ? ldw<KIND>(L, k, M::RG + i) : 0.0;
    }
#pragma unroll
    for (int r = 0; r < M::R; ++r) {
        ll[r] = lu[r] = tl[r] = tu[r] = til_[r] = tiu_[r] = 1.0;
        if (!row_present<KIND>(r)) continue;
        ll[r] = ldw<KIND>(L, k, M::LAML + r);
        lu[r] = ldw<KIND>(L, k, M::LAMU + r);
        tl[r] = ldw<KIND>(L, k, M::TL + r);
        tu[r] = ldw<KIND>(L, k, M::TU + r);
        if (MPCX_STORE_RECIP) {
            til_[r] = ldw<KIND>(L, k, M::TIL + r);
            tiu_[r] = ldw<KIND>(L, k, M::TIU + r);
        }
    }
    AB &ab = in.ab;
    double (&bv)[NX] = in.bv;
    ab = AB{};
#pragma unroll
    for (int i = 0; i < NX; ++i) bv[i] = 0.0;
    if (KIND != LAST) {
        load_ab<NH, KIND>(L, k, ab);
#pragma unroll
        for (int i = 0; i < NX; ++i) bv[i] = ldw<KIND>(L, k, M::BV + i);
    }
    __builtin_amdgcn_sched_barrier(0);     // every load of the stage is in flight before the first use
}

// the pending step on the primal.  ST = false: nothing is stored, the caller keeps the stage in
// registers (in.v is the updated value either way)
template <int NH, int KIND, bool ST = true>
DEV void stage_B0_apply(const Lane &L, int k, const Step &sp, B0In<NH> &in, const KR &kr = KR{false, false})
{
    typedef Map<NH> M;
    double (&v)[NZ] = in.v, (&vo)[NZ] = in.vo, (&dv)[NZ] = in.dv;
#pragma unroll
    for (int j = 0; j < NZ; ++j) {
        v[j] = vo[j];
        if (var_present<KIND>(j)) {
            v[j] = (sp.have && var_on<KIND>(kr, j)) ? fma(sp.alpha, dv[j], vo[j]) : vo[j];
            if (ST) stw<KIND>(L, k, M::V + j, v[j]);
        }
    }
}

// The pending step on the stage's multipliers and slacks, and the stage's own part of the step of its
// dynamics multiplier.  That step solves the x_k block of the Newton system,
//   Ht_xx dx_k + A_k' dpi_(k+1) - dpi_k = -gt_x,
// backwards from the last stage (dpi_chain below), gt and Ht being the right-hand side and the barrier
// Hessian of the system that produced the step; both are formed again here from the iterate before the
// step and the stationarity residual it left (in.rgo), so neither the cost-to-go Hessian nor its
// gradient has to be kept.  cx = gt_x + Ht_xx dx_k.
// RC = true (stage resident in registers): in.til_ / in.tiu_ hold the reciprocals of the slacks.
template <int NH, int KIND, bool ST = true, bool RC = false>
DEV void stage_B0_step_rows(const Lane &L, const mpcx_opts &o, const PC<NH> &pc, int k, const Step &sp, B0In<NH> &in,
                            double *cx, const KR &kr = KR{false, false})
{
    typedef Map<NH> M;
    double (&x)[NX] = in.x, (&u)[NU] = in.u, (&vo)[NZ] = in.vo, (&dv)[NZ] = in.dv, (&dva)[NZ] = in.dva;
    double (&ll)[M::R] = in.ll, (&lu)[M::R] = in.lu, (&tl)[M::R] = in.tl, (&tu)[M::R] = in.tu,
           (&til_)[M::R] = in.til_, (&tiu_)[M::R] = in.tiu_;
    double g[NZ], Hd[NZ], gto[NX], Hto[NX], oxx = 0.0, oxy = 0.0, oyy = 0.0;
    cost_terms<NH, KIND>(L, o, pc, kr, x, u, g, Hd);
#pragma unroll
    for (int i = 0; i < NX; ++i) { gto[i] = in.rgo[i]; Hto[i] = Hd[NU + i]; }
#pragma unroll
    for (int r = 0; r < M::R; ++r) {
        if (!row_present<KIND>(r)) continue;
        const bool pr = row_on<KIND>(kr, r) && ((r < 9) || (r - 9 < L.nobs));
        double c, jx, jy, lo, up;
        row_geom<NH>(pc, r, x, u, c, jx, jy);
        row_bounds<NH, KIND>(L, pc, kr, r, lo, up);
        const double dlo = lo - c, dup = up - c;
        double l_l = ll[r], l_u = lu[r], t_l = tl[r], t_u = tu[r];
        double jvo = jdot(r, jx, jy, vo);
        double rdl = (dlo - jvo) + t_l;
        double rdu = (jvo - dup) + t_u;
        double til = (MPCX_STORE_RECIP || RC) ? til_[r] : 1.0 / t_l, tiu = (MPCX_STORE_RECIP || RC) ? tiu_[r] : 1.0 / t_u;
        double jda = jdot(r, jx, jy, dva);
        double rml, rmu;
        row_rm(sp.mode, sp.sigma_mu, jda, l_l, l_u, t_l, t_u, til, tiu, rdl, rdu, rml, rmu);
        if (r >= NU && KIND != FIRST) {
            // this row in the right-hand side and the barrier Hessian of the system solved
            const double qo = fma(-l_l, rdl, rml) * til - fma(-l_u, rdu, rmu) * tiu;
            const double wo = l_l * til + l_u * tiu;
            if (r < ROW_G0) {
                gto[r - NU] = pr ? gto[r - NU] + qo : gto[r - NU];
                Hto[r - NU] = pr ? Hto[r - NU] + wo : Hto[r - NU];
            } else {
                gto[0] = pr ? fma(jx, qo, gto[0]) : gto[0];
                gto[1] = pr ? fma(jy, qo, gto[1]) : gto[1];
                const double wx = wo * jx, wy = wo * jy;
                oxx = pr ? fma(wx, jx, oxx) : oxx;
                oxy = pr ? fma(wy, jx, oxy) : oxy;
                oyy = pr ? fma(wy, jy, oyy) : oyy;
            }
        }
        double jd = jdot(r, jx, jy, dv);
        double dtl = jd - rdl;
        double dtu = -jd - rdu;
        double dll = -(fma(l_l, dtl, rml) * til);
        double dlu = -(fma(l_u, dtu, rmu) * tiu);
        double t;
        t = fma(sp.alpha, dll, l_l); t = (t < o.lam_min) ? o.lam_min : t; l_l = sp.have ? t : l_l;
        t = fma(sp.alpha, dlu, l_u); t = (t < o.lam_min) ? o.lam_min : t; l_u = sp.have ? t : l_u;
        t = fma(sp.alpha, dtl, t_l); t = (t < o.t_min) ? o.t_min : t; t_l = sp.have ? t : t_l;
        t = fma(sp.alpha, dtu, t_u); t = (t < o.t_min) ? o.t_min : t; t_u = sp.have ? t : t_u;
        if (ST) {
            stw<KIND>(L, k, M::LAML + r, l_l);
            stw<KIND>(L, k, M::LAMU + r, l_u);
            stw<KIND>(L, k, M::TL + r, t_l);
            stw<KIND>(L, k, M::TU + r, t_u);
        }
        ll[r] = l_l; lu[r] = l_u; tl[r] = t_l; tu[r] = t_u;      // (in.* hold the updated values on return)
    }
    const double *dx = &dv[NU];
    cx[0] = fma(oxy, dx[1], fma(Hto[0] + oxx, dx[0], gto[0]));
    cx[1] = fma(Hto[1] + oyy, dx[1], fma(oxy, dx[0], gto[1]));
#pragma unroll
    for (int i = 2; i < NX; ++i) cx[i] = fma(Hto[i], dx[i], gto[i]);
}

// d = cx + A_k' dpn: the step of the dynamics multiplier of stage k from that of stage k+1
DEV void dpi_chain(const AB &ab, const double *cx, const double *dpn, double *d)
{
    d[0] = cx[0] + dpn[0];
    d[1] = cx[1] + dpn[1];
    d[2] = fma(ab.a12, dpn[1], fma(ab.a02, dpn[0], cx[2])) + dpn[2];
    d[3] = fma(ab.a23, dpn[2], fma(ab.a13, dpn[1], fma(ab.a03, dpn[0], cx[3]))) + dpn[3];
    d[4] = fma(ab.a24, dpn[2], fma(ab.a14, dpn[1], fma(ab.a04, dpn[0], cx[4]))) + dpn[4];
}

// the pending step on the dynamics multiplier, d = its step (in.qpi is the updated value)
template <int NH, int KIND, bool ST = true>
DEV void stage_B0_apply_pi(const Lane &L, int k, const Step &sp, B0In<NH> &in, const double *d, const KR &kr = KR{false, false})
{
    typedef Map<NH> M;
    double (&qpi)[NX] = in.qpi;
    if (KIND != FIRST) {
#pragma unroll
        for (int i = 0; i < NX; ++i) {
            qpi[i] = sp.have ? fma(sp.alpha, d[i], qpi[i]) : qpi[i];
            qpi[i] = is_first<KIND>(kr) ? 0.0 : qpi[i];
            if (ST) stw<KIND>(L, k, M::QPI + i, qpi[i]);
        }
    }
}

// second half (the pending step has been applied to the whole stage): rows, residuals, barrier Hessian,
// affine right-hand side.  vxn / qpin: updated x-part of v and dynamics multiplier of stage k+1.
// RC = true (stage resident in registers): in.til_ / in.tiu_ carry the reciprocals of the slacks from
// sweep to sweep; on return they hold those of the updated slacks
template <int NH, int KIND, bool ST = true, bool RC = false>
DEV void stage_B0_tail(const Lane &L, const mpcx_opts &o, const PC<NH> &pc, int k, const Step &sp, B0In<NH> &in,
                       const double *vxn, const double *qpin, B0Out &out, const KR &kr = KR{false, false})
{
    typedef Map<NH> M;
    out.ng = out.nb = out.nd = out.nm = 0.0;
    double ssum = 0.0;
    double (&x)[NX] = in.x, (&u)[NU] = in.u, (&v)[NZ] = in.v, (&qpi)[NX] = in.qpi;
    double (&ll)[M::R] = in.ll, (&lu)[M::R] = in.lu, (&tl)[M::R] = in.tl, (&tu)[M::R] = in.tu,
           (&til_)[M::R] = in.til_, (&tiu_)[M::R] = in.tiu_;
    AB &ab = in.ab;
    double (&bv)[NX] = in.bv;
    double g[NZ], Ht[NZ], rg[NZ], hxx = 0.0, hxy = 0.0, hyy = 0.0;
    cost_terms<NH, KIND>(L, o, pc, kr, x, u, g, Ht);
#pragma unroll
    for (int j = 0; j < NZ; ++j) rg[j] = fma(Ht[j], v[j], g[j]);
    if (KIND != LAST) {
        double rn[NZ];
#pragma unroll
        for (int j = 0; j < NZ; ++j) rn[j] = rg[j];
        add_BAt(L, ab, qpin, rn);
#pragma unroll
        for (int j = 0; j < NZ; ++j) rg[j] = is_last<KIND>(kr) ? rg[j] : rn[j];
    }
    if (KIND != FIRST) {
#pragma unroll
        for (int i = 0; i < NX; ++i) rg[NU + i] = is_first<KIND>(kr) ? rg[NU + i] : rg[NU + i] - qpi[i];
    }
    // ---- rows ----
    double qrow[M::R], gjx[M::R], gjy[M::R];
#pragma unroll
    for (int r = 0; r < M::R; ++r) {
        qrow[r] = 0.0; gjx[r] = 0.0; gjy[r] = 0.0;
        if (!row_present<KIND>(r)) continue;
        const bool pr = row_on<KIND>(kr, r) && ((r < 9) || (r - 9 < L.nobs));
        double c, jx, jy, lo, up;
        row_geom<NH>(pc, r, x, u, c, jx, jy);
        row_bounds<NH, KIND>(L, pc, kr, r, lo, up);
        const double dlo = lo - c, dup = up - c;
        const double l_l = ll[r], l_u = lu[r], t_l = tl[r], t_u = tu[r];      // (after the pending step)
        double jv = jdot(r, jx, jy, v);
        double dlm = l_u - l_l;
        if (r < ROW_G0) rg[r] = pr ? rg[r] + dlm : rg[r];
        else {
            rg[2] = pr ? fma(jx, dlm, rg[2]) : rg[2];
            rg[3] = pr ? fma(jy, dlm, rg[3]) : rg[3];
        }
        double rdl = (dlo - jv) + t_l;
        double rdu = (jv - dup) + t_u;
        double nd = dmax(dmax(out.nd, dabs(rdl)), dabs(rdu));
        double ml = l_l * t_l, mu_ = l_u * t_u;
        double nm = dmax(dmax(out.nm, dabs(ml)), dabs(mu_));
        double sum = (ssum + ml) + mu_;
        out.nd = pr ? nd : out.nd;
        out.nm = pr ? nm : out.nm;
        ssum = pr ? sum : ssum;
        // affine right-hand side and barrier Hessian
        double til = 1.0 / t_l, tiu = 1.0 / t_u;
        if (MPCX_STORE_RECIP) {
            stw<KIND>(L, k, M::TIL + r, til);
            stw<KIND>(L, k, M::TIU + r, tiu);
        }
        if (RC) { til_[r] = til; tiu_[r] = tiu; }
        double ql = fma(-l_l, rdl, ml) * til;
        double qu = fma(-l_u, rdu, mu_) * tiu;
        qrow[r] = ql - qu;
        gjx[r] = jx;
        gjy[r] = jy;
        double w = l_l * til + l_u * tiu;
        if (r < ROW_G0) Ht[r] = pr ? Ht[r] + w : Ht[r];
        else {
            double wx = w * jx, wy = w * jy;
            hxx = pr ? fma(wx, jx, hxx) : hxx;
            hxy = pr ? fma(wy, jx, hxy) : hxy;
            hyy = pr ? fma(wy, jy, hyy) : hyy;
        }
    }
    out.ssum = ssum;
    double gt[NZ];
#pragma unroll
    for (int j = 0; j < NZ; ++j) {
        if (ST && j >= NU) stw<KIND>(L, k, M::RG + j - NU, rg[j]);
        out.rg[j] = rg[j];
        if (var_present<KIND>(j)) out.ng = var_on<KIND>(kr, j) ? dmax(out.ng, dabs(rg[j])) : out.ng;
        gt[j] = rg[j];
    }
#pragma unroll
    for (int r = 0; r < M::R; ++r) {
        if (!row_present<KIND>(r)) continue;
        const bool pr = row_on<KIND>(kr, r) && ((r < 9) || (r - 9 < L.nobs));
        if (r < ROW_G0) gt[r] = pr ? gt[r] + qrow[r] : gt[r];
        else {
            gt[2] = pr ? fma(gjx[r], qrow[r], gt[2]) : gt[2];
            gt[3] = pr ? fma(gjy[r], qrow[r], gt[3]) : gt[3];
        }
    }
#pragma unroll
    for (int j = 0; j < NZ; ++j) { out.gt[j] = gt[j]; out.Ht[j] = Ht[j]; }
    out.hxx = hxx; out.hxy = hxy; out.hyy = hyy;
    out.ab = ab;
#pragma unroll
    for (int i = 0; i < NX; ++i) { out.rb[i] = 0.0; out.vx[i] = v[NU + i]; out.qpi[i] = qpi[i]; }
    if (KIND != LAST) {
        // dynamics residual
        mul_AB(L, ab, &v[NU], &v[0], bv, out.rb);
    #pragma unroll
        for (int i = 0; i < NX; ++i) {
            out.rb[i] = out.rb[i] - vxn[i];
            if (ST) stw<KIND>(L, k, M::RB + i, out.rb[i]);
            out.nb = is_last<KIND>(kr) ? out.nb : dmax(out.nb, dabs(out.rb[i]));
        }
    }
}

// B0, row part of stage k: applies the pending step to the stage's iterate, then residuals,
// barrier Hessian and affine right-hand side.  dpn: step of the dynamics multiplier of stage k+1 on
// entry, of stage k on return.
template <int NH, int KIND>
DEV void stage_B0_rows(const Lane &L, const mpcx_opts &o, const PC<NH> &pc, int k, const Step &sp,
                       const double *vxn, const double *qpin, double *dpn, B0Out &out)
{
    B0In<NH> in;
    stage_B0_load<NH, KIND>(L, k, sp, in);
    stage_B0_apply<NH, KIND>(L, k, sp, in);
    double cx[NX];
    stage_B0_step_rows<NH, KIND>(L, o, pc, k, sp, in, cx);
    if (KIND != FIRST) {
        double d[NX];
        if (KIND == LAST) {
#pragma unroll
            for (int i = 0; i < NX; ++i) d[i] = cx[i];
        } else {
            dpi_chain(in.ab, cx, dpn, d);
        }
        stage_B0_apply_pi<NH, KIND>(L, k, sp, in, d);
#pragma unroll
        for (int i = 0; i < NX; ++i) dpn[i] = d[i];
    }
    stage_B0_tail<NH, KIND>(L, o, pc, k, sp, in, vxn, qpin, out);
}

// B0, Riccati part of stage k: factorisation and backward substitution of the affine
// right-hand side; Pn, pn = P(k+1), p(k+1); Po, po receive P(k), p(k) (may be the same arrays:
// every read of the inputs precedes the first write of the outputs)
struct RicOut { double prb[NX], l10, d0, d1, K[NU][NX], kv[NU]; };

// feed-forward of a stage from the input part (m0, m1) of its right-hand side: -(L D L')^-1 m
DEV void feed_forward(double l10, double d0, double d1, double m0, double m1, double *kv)
{
    double y0 = m0;
    double y1 = fma(-l10, y0, m1);
    double z0 = y0 * d0, z1 = y1 * d1;
    double s0 = fma(-l10, z1, z0);
    kv[0] = -s0;
    kv[1] = -z1;
}

// prb = P(k+1) rb of a stage
DEV void ric_prb(const B0Out &in, const double *Pn, double *prb)
{
#pragma unroll
    for (int i = 0; i < NX; ++i) {
        double acc = psym(Pn, i, 0) * in.rb[0];
#pragma unroll
        for (int j = 1; j < NX; ++j) acc = fma(psym(Pn, i, j), in.rb[j], acc);
        prb[i] = acc;
    }
}

// FF = false: kv receives (m0, m1) instead of the feed-forward, which the caller forms with
// feed_forward() when it needs it (the lane pipelines need it once, after their last step).
// PART = 0: the whole stage.  The two recursions of a stage do not depend on each other's results of
// the same step, so a lane pipeline may run them one after the other: PART = 1 is the factorisation alone
// (K, the pivots, P(k) from P(k+1)), PART = 2 the backward substitution alone (p(k) and the input part of
// the right-hand side from p(k+1), with ro.prb = P(k+1) rb and ro.K already in place).
template <int KIND, bool FF = true, int PART = 0>
DEV void stage_B0_ric(const Lane &L, const B0Out &in, const double *Pn, const double *pn, double *Po, double *po,
                      RicOut &ro)
{
    if (KIND == LAST) {
        // P_N = barrier Hessian of the terminal state, p_N = its right-hand side
#pragma unroll
        for (int i = 0; i < NX; ++i)
#pragma unroll
            for (int j = 0; j <= i; ++j) {
                double e = 0.0;
                if (i == j) e = (i == 0) ? in.Ht[2] + in.hxx : (i == 1) ? in.Ht[3] + in.hyy : in.Ht[NU + i];
                else if (i == 1 && j == 0) e = in.hxy;
                Po[tri(i, j)] = e;
            }
#pragma unroll
        for (int i = 0; i < NX; ++i) po[i] = in.gt[NU + i];
    } else {
        // w = p(k+1) + P(k+1) rb
        double w[NX];
        if (PART == 0) ric_prb(in, Pn, ro.prb);
        if (PART != 1) {
#pragma unroll
            for (int i = 0; i < NX; ++i) w[i] = pn[i] + ro.prb[i];
        }
        double Kg[NU][NX];
        if (PART != 2) {
            // PBA = P(k+1) [B A]; columns 0,1 = B, 2..6 = A
            double PBA[NX][NZ];
#pragma unroll
            for (int i = 0; i < NX; ++i) {
                const double p0 = psym(Pn, i, 0), p1 = psym(Pn, i, 1), p2 = psym(Pn, i, 2),
                             p3 = psym(Pn, i, 3), p4 = psym(Pn, i, 4);
                PBA[i][0] = fma(p3, L.hB, fma(p2, in.ab.b20, fma(p1, in.ab.b10, p0 * in.ab.b00)));
                PBA[i][1] = fma(p4, L.hB, fma(p2, in.ab.b21, fma(p1, in.ab.b11, p0 * in.ab.b01)));
                PBA[i][2] = p0;
                PBA[i][3] = p1;
                PBA[i][4] = fma(p1, in.ab.a12, p0 * in.ab.a02) + p2;
                PBA[i][5] = fma(p2, in.ab.a23, fma(p1, in.ab.a13, p0 * in.ab.a03)) + p3;
                PBA[i][6] = fma(p2, in.ab.a24, fma(p1, in.ab.a14, p0 * in.ab.a04)) + p4;
            }
            // M = Ht + [B A]' PBA, lower triangle
            double Mm[NZ][NZ];
#pragma unroll
            for (int c = 0; c < NZ; ++c) {
#pragma unroll
                for (int r = c; r < NZ; ++r) {
                    double acc = 0.0;
                    if (r == c) acc = (r == 2) ? in.Ht[2] + in.hxx : (r == 3) ? in.Ht[3] + in.hyy : in.Ht[r];
                    else if (r == 3 && c == 2) acc = in.hxy;
                    if (r == 0)
                        acc = fma(L.hB, PBA[3][c], fma(in.ab.b20, PBA[2][c], fma(in.ab.b10, PBA[1][c], fma(in.ab.b00, PBA[0][c], acc))));
                    else if (r == 1)
                        acc = fma(L.hB, PBA[4][c], fma(in.ab.b21, PBA[2][c], fma(in.ab.b11, PBA[1][c], fma(in.ab.b01, PBA[0][c], acc))));
                    else if (r == 2) acc = acc + PBA[0][c];
                    else if (r == 3) acc = acc + PBA[1][c];
                    else if (r == 4) acc = fma(in.ab.a12, PBA[1][c], fma(in.ab.a02, PBA[0][c], acc)) + PBA[2][c];
                    else if (r == 5) acc = fma(in.ab.a23, PBA[2][c], fma(in.ab.a13, PBA[1][c], fma(in.ab.a03, PBA[0][c], acc))) + PBA[3][c];
                    else acc = fma(in.ab.a24, PBA[2][c], fma(in.ab.a14, PBA[1][c], fma(in.ab.a04, PBA[0][c], acc))) + PBA[4][c];
                    Mm[r][c] = acc;
                }
            }
            const double d0 = 1.0 / Mm[0][0];                          // reciprocal pivots
            const double l10 = Mm[1][0] * d0;
            const double d1 = 1.0 / fma(-l10, Mm[1][0], Mm[1][1]);
            ro.l10 = l10; ro.d0 = d0; ro.d1 = d1;
#pragma unroll
            for (int i = 0; i < NX; ++i) {
                double y0 = Mm[NU + i][0];
                double y1 = fma(-l10, y0, Mm[NU + i][1]);
                double z0 = y0 * d0, z1 = y1 * d1;
                double s0 = fma(-l10, z1, z0);
                Kg[0][i] = -s0;
                Kg[1][i] = -z1;
                ro.K[0][i] = Kg[0][i];
                ro.K[1][i] = Kg[1][i];
            }
            // P(k) = M_xx + M_xu K
#pragma unroll
            for (int i = 0; i < NX; ++i)
#pragma unroll
                for (int j = 0; j <= i; ++j) {
                    double e = fma(Mm[NU + i][0], Kg[0][j], Mm[NU + i][NU + j]);
                    Po[tri(i, j)] = fma(Mm[NU + i][1], Kg[1][j], e);
                }
        } else {
#pragma unroll
            for (int i = 0; i < NX; ++i) { Kg[0][i] = ro.K[0][i]; Kg[1][i] = ro.K[1][i]; }
        }
        if (PART != 1) {
            // right-hand side of this stage and feed-forward
            double m[NZ];
#pragma unroll
            for (int j = 0; j < NZ; ++j) m[j] = in.gt[j];
            add_BAt(L, in.ab, w, m);
            if (FF) feed_forward(ro.l10, ro.d0, ro.d1, m[0], m[1], ro.kv);
            else { ro.kv[0] = m[0]; ro.kv[1] = m[1]; }
#pragma unroll
            for (int i = 0; i < NX; ++i) po[i] = fma(Kg[1][i], m[1], fma(Kg[0][i], m[0], m[NU + i]));
        }
    }
}

// the factors of stage k go to the workspace
template <int NH, int KIND>
DEV void store_ric(const Lane &L, int k, bool factors, const RicOut &ro)
{
    typedef Map<NH> M;
    if (factors) {
#pragma unroll
        for (int i = 0; i < NX; ++i) {
            stw<KIND>(L, k, M::K + i, ro.K[0][i]);
            stw<KIND>(L, k, M::K + NX + i, ro.K[1][i]);
        }
        stw<KIND>(L, k, M::LDL + 0, ro.l10);
        stw<KIND>(L, k, M::LDL + 1, ro.d0);
        stw<KIND>(L, k, M::LDL + 2, ro.d1);
        stw<KIND>(L, k, M::KV + 0, ro.kv[0]);
        stw<KIND>(L, k, M::KV + 1, ro.kv[1]);
    }
}

template <int NH>
__device__ __attribute__((noinline)) void sweep_B0(Lane L, PC<NH> pc, gopts *op, bool sp_have, int sp_mode,
                                                   double sp_sigma_mu, double sp_alpha, int m_total, ldouble *ret)
{
    const mpcx_opts ol = opts_in_registers(op);
    const mpcx_opts &o = ol;
    Step sp;
    sp.have = sp_have; sp.mode = sp_mode; sp.sigma_mu = sp_sigma_mu; sp.alpha = sp_alpha;
    const int N = L.N;
    double Pn[15], pn[NX], vxn[NX], qpin[NX], dpn[NX];
#pragma unroll
    for (int i = 0; i < NX; ++i) { pn[i] = 0.0; vxn[i] = 0.0; qpin[i] = 0.0; dpn[i] = 0.0; }
#pragma unroll
    for (int i = 0; i < 15; ++i) Pn[i] = 0.0;
    double ng = 0.0, nb = 0.0, nd = 0.0, nm = 0.0, sum = 0.0;
#define B0_STAGE(KIND, kk)                                                             \
    {                                                                                  \
        B0Out t;                                                                       \
        stage_B0_rows<NH, KIND>(L, o, pc, kk, sp, vxn, qpin, dpn, t);                  \
        RicOut ro;                                                                     \
        stage_B0_ric<KIND>(L, t, Pn, pn, Pn, pn, ro);                                  \
        store_ric<NH, KIND>(L, kk, KIND != LAST, ro);                                  \
        ng = dmax(ng, t.ng); nb = dmax(nb, t.nb); nd = dmax(nd, t.nd); nm = dmax(nm, t.nm); \
        sum = sum + t.ssum;                                                            \
        _Pragma("unroll") for (int i = 0; i < NX; ++i) { vxn[i] = t.vx[i]; qpin[i] = t.qpi[i]; } \
    }
    B0_STAGE(LAST, N)
    {
        int k = N - 1;
        for (; k >= 1 && k >= N - MPCX_EDGE_STAGES; --k) B0_STAGE(MIDC, k)
        for (; k > MPCX_EDGE_STAGES; --k) B0_STAGE(MID, k)
        for (; k >= 1; --k) B0_STAGE(MIDC, k)
    }
    B0_STAGE(FIRST, 0)
#undef B0_STAGE
    ret[0 * 64] = ng; ret[1 * 64] = nb; ret[2 * 64] = nd; ret[3 * 64] = nm;
    ret[4 * 64] = sum / (double)m_total;
}

// Bs: right-hand sides of the corrector (1) and of the centring-only step (2), both
// substituted backwards with the stored factors in the same pass over the data
struct BsOut {
    double gt1[NZ], gt2[NZ], k0[NX], k1[NX], l10, d0, d1;
    AB ab;
};

// what Bs reads of a stage besides the factors that go straight into BsOut: the two per-variable sums the
// affine forward sweep left (stage_F_compute, NS = 1)
struct BsIn {
    double ga[NZ], gb[NZ];
};

// Bs, loads of stage k: the two sums (in) and the stored factors of the stage (out)
template <int NH, int KIND>
DEV void stage_Bs_load(const Lane &L, int k, BsIn &in, BsOut &out, const KR &kr = KR{false, false})
{
    typedef Map<NH> M;
#pragma unroll
    for (int j = 0; j < NZ; ++j) {
        in.ga[j] = ldw<KIND>(L, k, M::GA + j);
        in.gb[j] = ldw<KIND>(L, k, M::GB + j);
    }
    out.ab = AB{};
    out.l10 = 0.0; out.d0 = 1.0; out.d1 = 1.0;
#pragma unroll
    for (int i = 0; i < NX; ++i) { out.k0[i] = 0.0; out.k1[i] = 0.0; }
    if (KIND != LAST) {
        load_ab<NH, KIND>(L, k, out.ab);
#pragma unroll
        for (int i = 0; i < NX; ++i) {
            out.k0[i] = ldw<KIND>(L, k, M::K + i);
            out.k1[i] = ldw<KIND>(L, k, M::K + NX + i);
        }
        out.l10 = ldw<KIND>(L, k, M::LDL + 0);
        out.d0 = ldw<KIND>(L, k, M::LDL + 1);
        out.d1 = ldw<KIND>(L, k, M::LDL + 2);
    }
    __builtin_amdgcn_sched_barrier(0);
}

// Bs, right-hand sides of the two corrections to the affine step.  The corrector and the centring-only step
// solve the affine step's system with other complementarity targets, so each is the affine step plus the
// solution for the difference of the right-hand sides (no stationarity residual, no dynamics residual, no
// initial-state defect).  That difference is the rows' J' (c_l / t_l - c_u / t_u) with c = dt_a dlam_a -
// sigma*mu (corrector) or -sigma*mu (centring only): ga - sigma*mu gb and -sigma*mu gb.
DEV void stage_Bs_compute(double sigma_mu, const BsIn &in, BsOut &out)
{
#pragma unroll
    for (int j = 0; j < NZ; ++j) {
        out.gt1[j] = fma(-sigma_mu, in.gb[j], in.ga[j]);
        out.gt2[j] = (-sigma_mu) * in.gb[j];
    }
}

// The two sums formed from a stage resident in registers (the bursts, which would otherwise have to carry
// them from the affine forward sweep to here): the operations of stage_F_compute (NS = 1), on the affine
// step dva.  in.til_ / in.tiu_ hold the reciprocals of the slacks.
template <int NH, int KIND>
DEV void stage_Bs_sums(const Lane &L, const PC<NH> &pc, const B0In<NH> &in, BsIn &sums, const KR &kr)
{
    typedef Map<NH> M;
    double (&ga)[NZ] = sums.ga, (&gb)[NZ] = sums.gb;
#pragma unroll
    for (int j = 0; j < NZ; ++j) { ga[j] = 0.0; gb[j] = 0.0; }
#pragma unroll
    for (int r = 0; r < M::R; ++r) {
        if (!row_present<KIND>(r)) continue;
        const bool pr = row_on<KIND>(kr, r) && ((r < 9) || (r - 9 < L.nobs));
        double c, jx, jy, lo, up;
        row_geom<NH>(pc, r, in.x, in.u, c, jx, jy);
        row_bounds<NH, KIND>(L, pc, kr, r, lo, up);
        const double dlo = lo - c, dup = up - c;
        const double jv = jdot(r, jx, jy, in.v);
        const double rdl = (dlo - jv) + in.tl[r];
        const double rdu = (jv - dup) + in.tu[r];
        const double til = in.til_[r], tiu = in.tiu_[r];
        const double ml = in.ll[r] * in.tl[r], mu_ = in.lu[r] * in.tu[r];
        const double jd = jdot(r, jx, jy, in.dva);
        const double dtl = jd - rdl, dtu = -jd - rdu;
        const double dll = -(fma(in.ll[r], dtl, ml) * til);
        const double dlu = -(fma(in.lu[r], dtu, mu_) * tiu);
        const double ar = (dtl * dll) * til - (dtu * dlu) * tiu;
        const double br = til - tiu;
        if (r < ROW_G0) {
            ga[r] = pr ? ga[r] + ar : ga[r];
            gb[r] = pr ? gb[r] + br : gb[r];
        } else {
            ga[2] = pr ? fma(jx, ar, ga[2]) : ga[2];
            ga[3] = pr ? fma(jy, ar, ga[3]) : ga[3];
            gb[2] = pr ? fma(jx, br, gb[2]) : gb[2];
            gb[3] = pr ? fma(jy, br, gb[3]) : gb[3];
        }
    }
}

template <int NH, int KIND>
DEV void stage_Bs_rows(const Lane &L, int k, double sigma_mu, BsOut &out, const KR &kr = KR{false, false})
{
    BsIn in;
    stage_Bs_load<NH, KIND>(L, k, in, out, kr);
    stage_Bs_compute(sigma_mu, in, out);
}

// Bs, recursion part of a stage: pn1 / pn2 hold p(k+1) of the two systems on entry, p(k) on
// return; kvo[s] = feed-forward of system s
template <int KIND, bool FF = true>
DEV void stage_Bs_rec(const Lane &L, const BsOut &in, const double *pi1, const double *pi2, double *pn1, double *pn2,
                      double (*kvo)[NU])
{
    // (pi1 / pi2 may be pn1 / pn2: every read of an input precedes the first write of the outputs)
    if (KIND == LAST) {
#pragma unroll
        for (int i = 0; i < NX; ++i) { pn1[i] = in.gt1[NU + i]; pn2[i] = in.gt2[NU + i]; }
    } else {
#pragma unroll
        for (int sidx = 0; sidx < 2; ++sidx) {
            double *pn = sidx ? pn2 : pn1;
            const double *pin = sidx ? pi2 : pi1;
            const double *gt = sidx ? in.gt2 : in.gt1;
            double w[NX], m[NZ];
#pragma unroll
            for (int i = 0; i < NX; ++i) w[i] = pin[i];
#pragma unroll
            for (int j = 0; j < NZ; ++j) m[j] = gt[j];
            add_BAt(L, in.ab, w, m);
            if (FF) feed_forward(in.l10, in.d0, in.d1, m[0], m[1], kvo[sidx]);
            else { kvo[sidx][0] = m[0]; kvo[sidx][1] = m[1]; }
#pragma unroll
            for (int i = 0; i < NX; ++i) pn[i] = fma(in.k1[i], m[1], fma(in.k0[i], m[0], m[NU + i]));
        }
    }
}

template <int NH, int KIND>
DEV void store_rec(const Lane &L, int k, bool ff, const double (*kvo)[NU])
{
    typedef Map<NH> M;
    if (ff) {
        stw<KIND>(L, k, M::KV + 0, kvo[0][0]);  stw<KIND>(L, k, M::KV + 1, kvo[0][1]);
        stw<KIND>(L, k, M::KV2 + 0, kvo[1][0]); stw<KIND>(L, k, M::KV2 + 1, kvo[1][1]);
    }
}

template <int NH, int KIND>
DEV void stage_Bs(const Lane &L, const mpcx_opts &o, const PC<NH> &pc, int k, double sigma_mu,
                  double *pn1, double *pn2)
{
    BsOut t;
    stage_Bs_rows<NH, KIND>(L, k, sigma_mu, t);
    double kvo[2][NU];
    stage_Bs_rec<KIND>(L, t, pn1, pn2, pn1, pn2, kvo);
    store_rec<NH, KIND>(L, k, KIND != LAST, kvo);
}

template <int NH>
__device__ __attribute__((noinline)) void sweep_Bs(Lane L, PC<NH> pc, gopts *op, double sigma_mu)
{
    const mpcx_opts ol = opts_in_registers(op);
    const mpcx_opts &o = ol;
    const int N = L.N;
    double pn1[NX], pn2[NX];
#pragma unroll
    for (int i = 0; i < NX; ++i) { pn1[i] = 0.0; pn2[i] = 0.0; }
    stage_Bs<NH, LAST>(L, o, pc, N, sigma_mu, pn1, pn2);
    {
        int k = N - 1;
        for (; k >= 1 && k >= N - MPCX_EDGE_STAGES; --k) stage_Bs<NH, MIDC>(L, o, pc, k, sigma_mu, pn1, pn2);
        for (; k > MPCX_EDGE_STAGES; --k) stage_Bs<NH, MID>(L, o, pc, k, sigma_mu, pn1, pn2);
        for (; k >= 1; --k) stage_Bs<NH, MIDC>(L, o, pc, k, sigma_mu, pn1, pn2);
    }
    stage_Bs<NH, FIRST>(L, o, pc, 0, sigma_mu, pn1, pn2);
}

// ---------------------------------------------------------------------------
// forward sweeps.  NS = 1: the affine (predictor) step: stores it (DVA), returns
// its longest feasible length and the centring sums.  NS = 2: the corrector step
// and the centring-only step side by side: stores both (DV and DV2)
// and returns both lengths.
// ---------------------------------------------------------------------------
// the longest feasible step of a stage is num/den: the smallest ratio met in the stage, kept
// as a pair and compared by cross multiplication (one division per stage); the step of the
// sweep is the minimum over the stages, which does not depend on the order of the stages
struct FAcc { double alpha[2], s1, s2; };

DEV void ratio_min(double val, double step, double &num, double &den)
{
    const bool c = (step < 0.0) && (val * den < num * (-step));
    num = c ? val : num;
    den = c ? -step : den;
}

// A copy of x the compiler cannot see through.  The lane pipelines below run some hundred steps on a
// few operands while the rest of the stage waits in the accumulator registers; taking the operands
// through this just before the loop gives them live ranges of their own, which the register allocator
// then keeps in vector registers for the length of the loop (instead of reading them back every step).
DEV double fresh(double x)
{
    asm volatile("" : "+v"(x));
    return x;
}
DEV void fresh_ab(AB &ab)
{
    ab.a02 = fresh(ab.a02); ab.a03 = fresh(ab.a03); ab.a04 = fresh(ab.a04); ab.a12 = fresh(ab.a12);
    ab.a13 = fresh(ab.a13); ab.a14 = fresh(ab.a14); ab.a23 = fresh(ab.a23); ab.a24 = fresh(ab.a24);
    ab.b00 = fresh(ab.b00); ab.b10 = fresh(ab.b10); ab.b20 = fresh(ab.b20); ab.b01 = fresh(ab.b01);
    ab.b11 = fresh(ab.b11); ab.b21 = fresh(ab.b21);
}

// forward recursion of one stage: du = kv + K dx, dxn = A dx + B du + rb
DEV void F_rec(const Lane &L, const AB &ab, const double (*kg)[NX], const double *kv, const double *rb,
               const double *dx, double *du, double *dxn)
{
#pragma unroll
    for (int c = 0; c < NU; ++c) {
        double acc = kv[c];
#pragma unroll
        for (int i = 0; i < NX; ++i) acc = fma(kg[c][i], dx[i], acc);
        du[c] = acc;
    }
    mul_AB(L, ab, dx, du, rb, dxn);
}

// everything a forward sweep reads of a stage
template <int NH, int NS>
struct FIn {
    double x[NX], u[NU], v[NZ], dva[NZ];
    double ll[Map<NH>::R], lu[Map<NH>::R], tl[Map<NH>::R], tu[Map<NH>::R], til_[Map<NH>::R], tiu_[Map<NH>::R];
    AB ab;
    double rb[NX], kv[NS][NU], kg[NU][NX];
};

template <int NH, int KIND, int NS>
DEV void stage_F_load(const Lane &L, int k, FIn<NH, NS> &in, const KR &kr = KR{false, false})
{
    typedef Map<NH> M;
    double (&x)[NX] = in.x, (&u)[NU] = in.u, (&v)[NZ] = in.v, (&dva)[NZ] = in.dva;
    double (&ll)[M::R] = in.ll, (&lu)[M::R] = in.lu, (&tl)[M::R] = in.tl, (&tu)[M::R] = in.tu,
           (&til_)[M::R] = in.til_, (&tiu_)[M::R] = in.tiu_;
    load_xu<NH, KIND>(L, k, kr, x, u);
#pragma unroll
    for (int j = 0; j < NZ; ++j) {
        v[j] = ldw<KIND>(L, k, M::V + j);
        dva[j] = (NS == 1) ? 0.0 : ldw<KIND>(L, k, M::DVA + j);
    }
#pragma unroll
    for (int r = 0; r < M::R; ++r) {
        ll[r] = lu[r] = tl[r] = tu[r] = til_[r] = tiu_[r] = 1.0;
        if (!row_present<KIND>(r)) continue;
        ll[r] = ldw<KIND>(L, k, M::LAML + r);
        lu[r] = ldw<KIND>(L, k, M::LAMU + r);
        tl[r] = ldw<KIND>(L, k, M::TL + r);
        tu[r] = ldw<KIND>(L, k, M::TU + r);
        if (MPCX_STORE_RECIP) {
            til_[r] = ldw<KIND>(L, k, M::TIL + r);
            tiu_[r] = ldw<KIND>(L, k, M::TIU + r);
        }
    }
    AB &ab = in.ab;
    double (&rb)[NX] = in.rb, (&kv)[NS][NU] = in.kv, (&kg)[NU][NX] = in.kg;
    ab = AB{};
#pragma unroll
    for (int i = 0; i < NX; ++i) { rb[i] = 0.0; kg[0][i] = 0.0; kg[1][i] = 0.0; }
#pragma unroll
    for (int q = 0; q < NS; ++q) { kv[q][0] = 0.0; kv[q][1] = 0.0; }
    if (KIND != LAST) {
        load_ab<NH, KIND>(L, k, ab);
#pragma unroll
        for (int i = 0; i < NX; ++i) {
            rb[i] = (NS == 1) ? ldw<KIND>(L, k, M::RB + i) : 0.0;      // (the corrections have no dynamics residual)
            kg[0][i] = ldw<KIND>(L, k, M::K + i);
            kg[1][i] = ldw<KIND>(L, k, M::K + NX + i);
        }
#pragma unroll
        for (int q = 0; q < NS; ++q) {
            kv[q][0] = ldw<KIND>(L, k, M::KV + q * M::SOL2 + 0);
            kv[q][1] = ldw<KIND>(L, k, M::KV + q * M::SOL2 + 1);
        }
    }
    __builtin_amdgcn_sched_barrier(0);
}

// ST = false: nothing is stored; the steps of the stage come back in dvo[q][0..6] and (NS = 1) the two
// sums for sweep Bs in gao, gbo
template <int NH, int KIND, int NS, bool ST = true, bool RC = false, bool SUMS = true>
DEV void stage_F_compute(const Lane &L, const mpcx_opts &o, const PC<NH> &pc, int k, double sigma_mu,
                         FIn<NH, NS> &in, double (*dx)[NX], FAcc &a, const KR &kr = KR{false, false},
                         double (*dvo)[NZ] = nullptr, double *gao = nullptr, double *gbo = nullptr)
{
    typedef Map<NH> M;
    double (&x)[NX] = in.x, (&u)[NU] = in.u, (&v)[NZ] = in.v, (&dva)[NZ] = in.dva;
    double (&ll)[M::R] = in.ll, (&lu)[M::R] = in.lu, (&tl)[M::R] = in.tl, (&tu)[M::R] = in.tu,
           (&til_)[M::R] = in.til_, (&tiu_)[M::R] = in.tiu_;
    AB &ab = in.ab;
    double (&rb)[NX] = in.rb, (&kv)[NS][NU] = in.kv, (&kg)[NU][NX] = in.kg;
    double dvk[NS][NZ], dxn[NS][NX];
#pragma unroll
    for (int q = 0; q < NS; ++q) {
#pragma unroll
        for (int i = 0; i < NX; ++i) { dvk[q][NU + i] = dx[q][i]; dxn[q][i] = 0.0; }
        dvk[q][0] = dvk[q][1] = 0.0;
        if (KIND != LAST) {
            F_rec(L, ab, kg, kv[q], rb, dx[q], dvk[q], dxn[q]);
            dvk[q][0] = is_last<KIND>(kr) ? 0.0 : dvk[q][0];
            dvk[q][1] = is_last<KIND>(kr) ? 0.0 : dvk[q][1];
        }
        if (NS == 2) {      // the step = the affine step + the correction just formed (dx[q] carries the correction)
#pragma unroll
            for (int j = 0; j < NZ; ++j) dvk[q][j] = dva[j] + dvk[q][j];
        }
#pragma unroll
        for (int j = 0; j < NZ; ++j) {
            if (ST) stw<KIND>(L, k, (NS == 1 ? M::DVA : M::DV + q * M::SOL2) + j, dvk[q][j]);
            else dvo[q][j] = dvk[q][j];
        }
    }
    double s1k = 0.0, s2k = 0.0, knum[NS], kden[NS], ga[NZ], gb[NZ];
#pragma unroll
    for (int j = 0; j < NZ; ++j) { ga[j] = 0.0; gb[j] = 0.0; }
#pragma unroll
    for (int q = 0; q < NS; ++q) { knum[q] = 1.0; kden[q] = 1.0; }
#pragma unroll
    for (int r = 0; r < M::R; ++r) {
        if (!row_present<KIND>(r)) continue;
        const bool pr = row_on<KIND>(kr, r) && ((r < 9) || (r - 9 < L.nobs));
        double c, jx, jy, lo, up;
        row_geom<NH>(pc, r, x, u, c, jx, jy);
        row_bounds<NH, KIND>(L, pc, kr, r, lo, up);
        const double dlo = lo - c, dup = up - c;
        double jv = jdot(r, jx, jy, v);
        double rdl = (dlo - jv) + tl[r];
        double rdu = (jv - dup) + tu[r];
        double til = (MPCX_STORE_RECIP || RC) ? til_[r] : 1.0 / tl[r], tiu = (MPCX_STORE_RECIP || RC) ? tiu_[r] : 1.0 / tu[r];
        const double ml = ll[r] * tl[r], mu_ = lu[r] * tu[r];
        double rml[NS], rmu[NS];
        if (NS == 1) { rml[0] = ml; rmu[0] = mu_; }
        else {
            double jda = jdot(r, jx, jy, dva);
            const double dtla = jda - rdl, dtua = -jda - rdu;
            const double dlla = -(fma(ll[r], dtla, ml) * til);
            const double dlua = -(fma(lu[r], dtua, mu_) * tiu);
            rml[0] = fma(dtla, dlla, ml) - sigma_mu;
            rmu[0] = fma(dtua, dlua, mu_) - sigma_mu;
            rml[NS - 1] = ml - sigma_mu;
            rmu[NS - 1] = mu_ - sigma_mu;
        }
#pragma unroll
        for (int q = 0; q < NS; ++q) {
            double jd = jdot(r, jx, jy, dvk[q]);
            double dtl = jd - rdl;
            double dtu = -jd - rdu;
            double dll = -(fma(ll[r], dtl, rml[q]) * til);
            double dlu = -(fma(lu[r], dtu, rmu[q]) * tiu);
            double num = knum[q], den = kden[q];
            ratio_min(ll[r], dll, num, den);
            ratio_min(tl[r], dtl, num, den);
            ratio_min(lu[r], dlu, num, den);
            ratio_min(tu[r], dtu, num, den);
            knum[q] = pr ? num : knum[q];
            kden[q] = pr ? den : kden[q];
            if (NS == 1) {
                double s1 = s1k, s2 = s2k;
                s1 = fma(ll[r], dtl, s1);
                s1 = fma(tl[r], dll, s1);
                s1 = fma(lu[r], dtu, s1);
                s1 = fma(tu[r], dlu, s1);
                s2 = fma(dtl, dll, s2);
                s2 = fma(dtu, dlu, s2);
                s1k = pr ? s1 : s1k;
                s2k = pr ? s2 : s2k;
                // the rows' part of the right-hand sides of the corrections Bs solves for, per variable,
                // up to the factor sigma*mu that is only known once this sweep has ended
                // (SUMS = false: the caller forms them later, stage_Bs_sums)
                if (!SUMS) continue;
                const double ar = (dtl * dll) * til - (dtu * dlu) * tiu;
                const double br = til - tiu;
                if (r < ROW_G0) {
                    ga[r] = pr ? ga[r] + ar : ga[r];
                    gb[r] = pr ? gb[r] + br : gb[r];
                } else {
                    ga[2] = pr ? fma(jx, ar, ga[2]) : ga[2];
                    ga[3] = pr ? fma(jy, ar, ga[3]) : ga[3];
                    gb[2] = pr ? fma(jx, br, gb[2]) : gb[2];
                    gb[3] = pr ? fma(jy, br, gb[3]) : gb[3];
                }
            }
        }
    }
    if (NS == 1) {
        a.s1 = a.s1 + s1k; a.s2 = a.s2 + s2k;
        if (SUMS) {
#pragma unroll
            for (int j = 0; j < NZ; ++j) {
                if (ST) { stw<KIND>(L, k, M::GA + j, ga[j]); stw<KIND>(L, k, M::GB + j, gb[j]); }
                else { gao[j] = ga[j]; gbo[j] = gb[j]; }
            }
        }
    }
#pragma unroll
    for (int q = 0; q < NS; ++q) {
        const double ak = knum[q] / kden[q];       // longest step of this stage; the step is the min over stages
        a.alpha[q] = (ak < a.alpha[q]) ? ak : a.alpha[q];
    }
#pragma unroll
    for (int q = 0; q < NS; ++q)
#pragma unroll
        for (int i = 0; i < NX; ++i) dx[q][i] = dxn[q][i];
}

template <int NH, int KIND, int NS>
DEV void stage_F(const Lane &L, const mpcx_opts &o, const PC<NH> &pc, int k, double sigma_mu,
                 double (*dx)[NX], FAcc &a)
{
    FIn<NH, NS> in;
    stage_F_load<NH, KIND, NS>(L, k, in);
    stage_F_compute<NH, KIND, NS>(L, o, pc, k, sigma_mu, in, dx, a);
}

// NS = 1: returns the affine step length, ret[0], ret[1] = S1, S2.
// NS = 2: returns the corrector step length, ret[0] = the centring-only one.
template <int NH, int NS>
__device__ __attribute__((noinline)) double sweep_F(Lane L, PC<NH> pc, gopts *op, double sigma_mu, ldouble *ret)
{
    const mpcx_opts ol = opts_in_registers(op);
    const mpcx_opts &o = ol;
    const int N = L.N;
    FAcc a = {{1.0, 1.0}, 0.0, 0.0};
    double dx[NS][NX];
#pragma unroll
    for (int q = 0; q < NS; ++q)
#pragma unroll
        for (int i = 0; i < NX; ++i) dx[q][i] = 0.0;
    stage_F<NH, FIRST, NS>(L, o, pc, 0, sigma_mu, dx, a);
    {
        int k = 1;
        for (; k < N && k <= MPCX_EDGE_STAGES; ++k) stage_F<NH, MIDC, NS>(L, o, pc, k, sigma_mu, dx, a);
        for (; k < N - MPCX_EDGE_STAGES; ++k) stage_F<NH, MID, NS>(L, o, pc, k, sigma_mu, dx, a);
        for (; k < N; ++k) stage_F<NH, MIDC, NS>(L, o, pc, k, sigma_mu, dx, a);
    }
    stage_F<NH, LAST, NS>(L, o, pc, N, sigma_mu, dx, a);
    if (NS == 1) { ret[0 * 64] = a.s1; ret[1 * 64] = a.s2; }
    else ret[0 * 64] = a.alpha[1];
    return a.alpha[0];
}

// Diagnostic build only (-DMPCX_STAMPS): per-wavefront cycle sums of each phase
// of the tick loop, read back by mpcx_debug_read_stamps().  No stamp code is
// compiled into the shipped library.
#ifdef MPCX_STAMPS
#define N_STAMP 64
__device__ unsigned long long g_stamps[2048][N_STAMP];
#define STAMP_DECL() unsigned long long st_t0 = 0
#define STAMP_BEGIN() st_t0 = __builtin_amdgcn_s_memtime()
#define STAMP(slot, pred)                                                              \
    do {                                                                               \
        unsigned long long st_t1 = __builtin_amdgcn_s_memtime();                       \
        unsigned long long st_n = __popcll(__ballot(pred));                            \
        if (threadIdx.x == 0) {                                                        \
            g_stamps[blockIdx.x][2 * (slot)] += st_t1 - st_t0;                         \
            g_stamps[blockIdx.x][2 * (slot) + 1] += st_n;                              \
        }                                                                              \
        st_t0 = __builtin_amdgcn_s_memtime();                                          \
    } while (0)
// wall-clock marks (100 MHz counter): the slot keeps the time of the first call
#define STAMP_N(slot, n)                                                               \
    do {                                                                               \
        unsigned long long st_t1 = __builtin_amdgcn_s_memtime();                       \
        if (threadIdx.x == 0) {                                                        \
            g_stamps[blockIdx.x][2 * (slot)] += st_t1 - st_t0;                         \
            g_stamps[blockIdx.x][2 * (slot) + 1] += (n);                               \
        }                                                                              \
        st_t0 = __builtin_amdgcn_s_memtime();                                          \
    } while (0)
#define MARK(slot)                                                                     \
    do {                                                                               \
        if (g_stamps[blockIdx.x][2 * (slot)] == 0) g_stamps[blockIdx.x][2 * (slot)] = wall_clock64(); \
    } while (0)
#define MARK_LAST(slot) do { g_stamps[blockIdx.x][2 * (slot)] = wall_clock64(); } while (0)
// time and age when the wavefront first starts a burst of a problem at least `thr` iterations old (two words)
#define MARK_AGE(word, word_age, thr, age)                                                 \
    do {                                                                                   \
        if ((age) >= (thr) && g_stamps[blockIdx.x][word] == 0) {                           \
            g_stamps[blockIdx.x][word] = wall_clock64();                                   \
            g_stamps[blockIdx.x][word_age] = (unsigned long long)(age);                    \
        }                                                                                  \
    } while (0)
// cycles of a part of a non-inlined function (slots 16..19)
#define SUB_BEGIN() unsigned long long sub_t0 = __builtin_amdgcn_s_memtime()
#define SUB_END(slot)                                                                  \
    do {                                                                               \
        unsigned long long sub_t1 = __builtin_amdgcn_s_memtime();                      \
        if (threadIdx.x == 0) g_stamps[blockIdx.x][2 * (slot)] += sub_t1 - sub_t0;     \
        sub_t0 = __builtin_amdgcn_s_memtime();                                         \
    } while (0)
#else
#define MARK(slot) ((void)0)
#define MARK_LAST(slot) ((void)0)
#define MARK_AGE(word, word_age, thr, age) ((void)0)
#define STAMP_DECL() ((void)0)
#define STAMP_BEGIN() ((void)0)
#define STAMP(slot, pred) ((void)0)
#define STAMP_N(slot, n) ((void)0)
#define SUB_BEGIN() ((void)0)
#define SUB_END(slot) ((void)0)
#endif

// ---------------------------------------------------------------------------
// single-problem mode.  When a wavefront has only a few problems left (the end of
// a batch, or a call with a handful of problems) one lane per problem leaves the
// machine waiting on the memory latency of one dependency chain.  The sweeps
// below work on ONE problem, 64 consecutive stages at a time with lane j on the j-th
// of them: the row part of the 64 stages runs at once, the recursion over them runs
// once, on values broadcast from the lane that owns the stage, and the reductions
// visit the stages in the order the one-lane-per-problem sweeps do.  Every number is produced by
// the same operations in the same order, so which mode ran leaves no trace in
// the results.
// ---------------------------------------------------------------------------
DEV double bcast(double v, int src)
{
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), src);
    const int hi = __builtin_amdgcn_readlane(__double2hiint(v), src);
    return __hiloint2double(hi, lo);
}
DEV double uniform(double v)
{
    const int lo = __builtin_amdgcn_readfirstlane(__double2loint(v));
    const int hi = __builtin_amdgcn_readfirstlane(__double2hiint(v));
    return __hiloint2double(hi, lo);
}

// stores of one lane become visible to the loads of the others (same wavefront, same CU),
// and loads issued so far have their data before any later store is issued.  The sweeps of
// one interior-point iteration need none between them: in all of them a lane loads and stores
// fields of its own stage only (the lanes talk through DPP moves), and a thread sees its own
// stores.  One is needed where the mapping of lanes to data changes: before the first of the
// sweeps, inside sweep L (which reads the next stage), and when going back to one lane per problem.
DEV void coop_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
    __builtin_amdgcn_wave_barrier();
}

// C: the problem's Lane with C.lane = its column in the workspace; stage j of it is
// addressed as stage 0 of a Lane whose "lane" is moved by j stages
DEV Lane stage_lane(const Lane &C, unsigned j)
{
    Lane S = C;
    // (a wavefront's workspace is far below 4 GiB)
    if (C.region == 0) S.lane = j * (unsigned)(C.F * 512) + C.lane;
    else S.lane = C.lane + (j >> 6) * (unsigned)(C.F * 512) + (j & 63u) * 8u;
    return S;
}

DEV unsigned region_bytes(const Lane &L) { return (unsigned)(((L.N + 64) / 64) * L.F * 512); }

// C = L addressed at the data of the problem of lane p: its column, or its region
DEV void problem_home(const Lane &L, int p, Lane &C)
{
    C = L;
    const unsigned r = (unsigned)__builtin_amdgcn_readlane((int)L.roff, p);
    if (r != 0) { C.lane = r; C.region = region_bytes(L); }
    else C.lane = (L.region == 0) ? 8u * (unsigned)p : L.region * (unsigned)p;
    C.roff = 0;
}

// the problem of lane p as wave-uniform values: C.lane = p is its column in the workspace
template <int NH>
DEV void problem_of_lane(const Lane &L, const PC<NH> &pc, ldouble *pc_lds, int p, Lane &C, PC<NH> &pcc)
{
    problem_home(L, p, C);
    C.b = (unsigned)__builtin_amdgcn_readlane((int)L.b, p);
    C.nobs = __builtin_amdgcn_readlane(L.nobs, p);
    C.h = bcast(L.h, p); C.hh = bcast(L.hh, p); C.h6 = bcast(L.h6, p);
    C.hB = bcast(L.hB, p); C.invL = bcast(L.invL, p);
    pcc.p = pc_lds + p;
    pcc.lmk = bcast(pc.lmk, p);
}

// Values handed from lane to lane in the recursions: lane j takes the value of lane j+1
// (or j-1); the lane at the end of the wavefront takes `edge`.
DEV double from_lane_above(double v, double edge)
{
    const int lo = __builtin_amdgcn_update_dpp(__double2loint(edge), __double2loint(v), 0x130, 0xf, 0xf, false);
    const int hi = __builtin_amdgcn_update_dpp(__double2hiint(edge), __double2hiint(v), 0x130, 0xf, 0xf, false);
    return __hiloint2double(hi, lo);      // wave_shl:1
}
DEV double from_lane_below(double v, double edge)
{
    const int lo = __builtin_amdgcn_update_dpp(__double2loint(edge), __double2loint(v), 0x138, 0xf, 0xf, false);
    const int hi = __builtin_amdgcn_update_dpp(__double2hiint(edge), __double2hiint(v), 0x138, 0xf, 0xf, false);
    return __hiloint2double(hi, lo);      // wave_shr:1
}

// The same with zero as the edge value (all a sweep over a single block of stages needs): the
// move then writes every lane, which saves the copy of `edge` into the destination beforehand.
template <bool ZERO>
DEV double lane_above(double v, double edge)
{
    if (!ZERO) return from_lane_above(v, edge);
    const int lo = __builtin_amdgcn_mov_dpp(__double2loint(v), 0x130, 0xf, 0xf, true);
    const int hi = __builtin_amdgcn_mov_dpp(__double2hiint(v), 0x130, 0xf, 0xf, true);
    return __hiloint2double(hi, lo);
}
template <bool ZERO>
DEV double lane_below(double v, double edge)
{
    if (!ZERO) return from_lane_below(v, edge);
    const int lo = __builtin_amdgcn_mov_dpp(__double2loint(v), 0x138, 0xf, 0xf, true);
    const int hi = __builtin_amdgcn_mov_dpp(__double2hiint(v), 0x138, 0xf, 0xf, true);
    return __hiloint2double(hi, lo);
}

// The same for sweeps over several blocks of stages, where the lane at the end of the wavefront takes the
// value the neighbouring block handed over.  `prev` is the destination of the previous step of the same
// pipeline, started from the block's edge value: the move leaves the end lane of its destination alone, so
// that lane holds the edge value throughout and the destination needs no copy of it each step.
template <bool ZERO>
DEV double lane_above_keep(double prev, double v)
{
    if (ZERO) return lane_above<true>(v, 0.0);
    const int lo = __builtin_amdgcn_update_dpp(__double2loint(prev), __double2loint(v), 0x130, 0xf, 0xf, false);
    const int hi = __builtin_amdgcn_update_dpp(__double2hiint(prev), __double2hiint(v), 0x130, 0xf, 0xf, false);
    return __hiloint2double(hi, lo);
}
template <bool ZERO>
DEV double lane_below_keep(double prev, double v)
{
    if (ZERO) return lane_below<true>(v, 0.0);
    const int lo = __builtin_amdgcn_update_dpp(__double2loint(prev), __double2loint(v), 0x138, 0xf, 0xf, false);
    const int hi = __builtin_amdgcn_update_dpp(__double2hiint(prev), __double2hiint(v), 0x138, 0xf, 0xf, false);
    return __hiloint2double(hi, lo);
}

// The maxima and minima over the stages do not depend on the order (the per-stage values are
// never NaN: each starts from the identity and is updated by a comparison that a NaN fails), so
// they are reduced by a DPP tree; lanes without a source take the identity.  The sums stay chains.
template <int CTRL>
DEV double dpp_take(double v, double idv)
{
    const int lo = __builtin_amdgcn_update_dpp(__double2loint(idv), __double2loint(v), CTRL, 0xf, 0xf, false);
    const int hi = __builtin_amdgcn_update_dpp(__double2hiint(idv), __double2hiint(v), CTRL, 0xf, 0xf, false);
    return __hiloint2double(hi, lo);
}
DEV double dmin(double a, double b) { return (b < a) ? b : a; }
// v >= 0 in every lane; the largest, in all lanes
DEV double wave_max(double v)
{
    v = dmax(v, dpp_take<0x111>(v, 0.0));      // row_shr:1, 2, 4, 8: lane 15 of a row has the row's
    v = dmax(v, dpp_take<0x112>(v, 0.0));
    v = dmax(v, dpp_take<0x114>(v, 0.0));
    v = dmax(v, dpp_take<0x118>(v, 0.0));
    v = dmax(v, dpp_take<0x142>(v, 0.0));      // row_bcast:15, then :31: lane 63 has the wavefront's
    v = dmax(v, dpp_take<0x143>(v, 0.0));
    return bcast(v, 63);
}
// v <= 1 in every lane; the smallest, in all lanes
DEV double wave_min(double v)
{
    v = dmin(v, dpp_take<0x111>(v, 1.0));
    v = dmin(v, dpp_take<0x112>(v, 1.0));
    v = dmin(v, dpp_take<0x114>(v, 1.0));
    v = dmin(v, dpp_take<0x118>(v, 1.0));
    v = dmin(v, dpp_take<0x142>(v, 1.0));
    v = dmin(v, dpp_take<0x143>(v, 1.0));
    return bcast(v, 63);
}

// The recursions run as a pipeline over the lanes.  Every lane keeps the operands of its own
// stage and, each step, recomputes its stage from what the neighbouring lane produced in the
// step before.  The lane at the start of the chain has its true input from the first step
// on, so after s steps the first s stages hold their true results, and a stage whose input
// is true reproduces the same result in every later step.  After as many steps as there are
// stages all of them are true; what the other lanes computed meanwhile is never used.
// LDS words of one wavefront behind the result slots of the sweeps: what a block of stages
// hands to the next one (wave-uniform values, parked there so that they do not occupy
// registers while the rows of the next block are computed)
enum { CY_P = 0, CY_PV = 15, CY_VX = 20, CY_QPI = 25, CY_NORM = 30, CY_DPI = 35, CY_N = 40 };

DEV void lds_order()
{
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

#define SPR_REC 38
// Doubles of LDS per wavefront between the result slots and the carry words, and the stages staged at a time:
// what is left of 40 KB -- four workgroups on a compute unit -- beside the problem constants, which grow with the
// obstacle rows; with four obstacles three workgroups fit anyway.  The staging area starts at the result slots
// (free while a recursion runs); a burst, which needs the carry words only when it ends, lets it run over them.
template <int NH> struct Spr {
    static constexpr int EXTRA = (NH == 0 || NH == 4) ? 704 : 5120 - (PCN<NH>::N + 6) * 64;
    static constexpr int CH_BURST = ((6 * 64 + EXTRA) / SPR_REC > 25) ? 25 : (6 * 64 + EXTRA) / SPR_REC;
    static constexpr int CH_BLOCK = ((5 * 64 + EXTRA) / SPR_REC > 25) ? 25 : (5 * 64 + EXTRA) / SPR_REC;
    static constexpr int CARRY = 5 * 64 + EXTRA;      // the carry words, in doubles behind the first result slot
    static_assert(EXTRA >= 0 && CH_BURST >= 8 && CH_BLOCK >= 8, "staging area");
};

// ---------------------------------------------------------------------------
// The backward Riccati recursion of a block of stages held one per lane (a burst's whole horizon, or one of
// the 64-stage blocks of a long one) -- factors and the affine backward substitution -- with the algebra of
// ONE stage spread over the lanes instead of a lane pipeline over the stages.
//
// The lane pipelines execute a stage's ~390 instructions in every lane for one useful lane a step.  Here the
// stages are taken in turn, N down to 0, and lane 8 r + c, r < 7, owns entry (r, c) of
// M = Ht + [B A]' P(k+1) [B A] for c < 7 and entry r of the right-hand side m = gt + [B A]' (p(k+1) + P(k+1) rb)
// for c = 7: P(k+1) travels as 15 wave-uniform values (v_readlane), p(k+1) reaches the right-hand-side lanes by
// ds_bpermute (the other lanes read -0 from lane 63 in the same instruction), every lane forms its column of
// P(k+1) [B A] (the right-hand-side lanes: P(k+1) rb, then + p(k+1)) and its entry with five multiply-adds
// each, every lane factorises the 2 x 2 input block itself, and one multiply-add pair gives entry (r, c) of
// P(k) = M_xx + M_xu K in the entry lanes and entry r of p(k) = m_x + K' m_u in the right-hand-side lanes (the
// same two instructions on other exchange partners).  Every number is produced by the operations of
// stage_B0_ric() in their order: the dense chains here contain the pattern chains there, and a multiplication
// by a structural 1 or an addition of a structural 0 product leaves a finite number as it is.
//
// The operands of a stage are uniform per step, so they come from LDS: the stage's own lane writes a record
// of SPR_REC doubles (word 37 takes the stores of lanes that have no result; [0,14) the pattern entries of A and B; 14 hB; 15 zero; 16 one; [17,24) the diagonal of the
// barrier Hessian, the (x, y) block's added; 24 its off-diagonal entry; [25,30) rb; [30,37) gt), CH stages at
// a time (CH), and takes the results back from it ([0,5) K row 0; [5,10) K row 1; 10 l10; 11, 12 the reciprocal
// pivots; 13, 14 the input part of the right-hand side).  `stg` is the wavefront's staging area of
// CH * SPR_REC doubles.  The lanes 0 .. nst-1 hold the stages to be done, the highest first.  P and p of the
// stage above them come from lane `jinit` if that lane holds the last stage of the horizon (its barrier Hessian
// and gradient), else (jinit < 0) from cP, cp, what the block above left; on return `pnew` holds P and p of the
// block's lowest stage, entry (i, j) of P in lane 8 (NU+i) + NU+j, entry i of p in lane 8 (NU+i) + 7.
// SUM: the complementarity sum rides along, one pipeline step per stage.
// ---------------------------------------------------------------------------
DEV int spr_word(int c, int j)      // word of entry (j, c) of [B A | rb] in a staging record
{
    switch (c) {
    case 0: return (j == 0) ? 8 : (j == 1) ? 9 : (j == 2) ? 10 : (j == 3) ? 14 : 15;
    case 1: return (j == 0) ? 11 : (j == 1) ? 12 : (j == 2) ? 13 : (j == 3) ? 15 : 14;
    case 2: return (j == 0) ? 16 : 15;
    case 3: return (j == 1) ? 16 : 15;
    case 4: return (j == 0) ? 0 : (j == 1) ? 3 : (j == 2) ? 16 : 15;
    case 5: return (j == 0) ? 1 : (j == 1) ? 4 : (j == 2) ? 6 : (j == 3) ? 16 : 15;
    case 6: return (j == 0) ? 2 : (j == 1) ? 5 : (j == 2) ? 7 : (j == 3) ? 15 : 16;
    default: return 25 + j;
    }
}
DEV double spr_perm(int addr, double v)
{
    const int lo = __builtin_amdgcn_ds_bpermute(addr, __double2loint(v));
    const int hi = __builtin_amdgcn_ds_bpermute(addr, __double2hiint(v));
    return __hiloint2double(hi, lo);
}
typedef __attribute__((address_space(3))) char lchar;

template <int CH, bool SUM>
DEV void ric_spread(const B0Out &t, double hB, int nst, int jinit, const double *cP, const double *cp, ldouble *stg,
                    RicOut &ro, double &sum, double &pnew)
{
    const int lane = (int)threadIdx.x;
    const int r = lane >> 3, c = lane & 7;
    const bool ent = r < NZ;
    const bool rhs = ent && c == 7;
    const int rr = ent ? r : 0, cc = ent ? c : 0;      // (lanes without an entry run along on entry (0, 0))
    unsigned oc[NX], orr[NX];
#pragma unroll
    for (int j = 0; j < NX; ++j) { oc[j] = 8u * (unsigned)spr_word(cc, j); orr[j] = 8u * (unsigned)spr_word(rr, j); }
    const unsigned off_h = (cc == 7) ? 8u * (unsigned)(30 + rr)
                                     : 8u * (unsigned)((rr == cc) ? 17 + rr : (rr == 3 && cc == 2) ? 24 : 15);
    // where a lane's two results of a step go in the record: the diagonal lanes (NU+j, NU+j) have column j of K,
    // lanes 0 and 1 the pivots, the lanes (0, 7), (1, 7) the input part of the right-hand side; the rest word 37
    const bool diag = ent && cc < 7 && rr == cc && rr >= NU;
    const bool kvl = rhs && rr < NU;
    const unsigned oA = 8u * (unsigned)(diag ? (cc - NU) : (lane == 0) ? 10 : kvl ? 13 + rr : 37);
    const unsigned oB = 8u * (unsigned)(diag ? (5 + cc - NU) : (lane == 0) ? 11 : (lane == 1) ? 12 : 37);
    // exchange partners.  Entry lanes: M[r][0], M[r][1] of the row and M[c][0], M[c][1] of the column.
    // Right-hand-side lanes: m[0], m[1], and M[r][0], M[r][1] of the row
    const int src_r0 = 4 * (rhs ? 8 * 0 + 7 : 8 * rr + 0), src_r1 = 4 * (rhs ? 8 * 1 + 7 : 8 * rr + 1);
    const int src_c0 = 4 * (rhs ? 8 * rr + 0 : 8 * cc + 0), src_c1 = 4 * (rhs ? 8 * rr + 1 : 8 * cc + 1);
    // p(k+1)[i] comes from its lane for the right-hand-side lanes and from lane 63, which holds -0, for the others
    int src_p[NX];
#pragma unroll
    for (int i = 0; i < NX; ++i) src_p[i] = 4 * (rhs ? 8 * (NU + i) + 7 : 63);
    // P and p above the block's highest stage
    pnew = (lane == 63) ? -0.0 : 0.0;
    if (jinit >= 0) {
#pragma unroll
        for (int i = 0; i < NX; ++i) {
#pragma unroll
            for (int j = 0; j <= i; ++j) {
                double e = 0.0;
                if (i == j) e = (i == 0) ? t.Ht[2] + t.hxx : (i == 1) ? t.Ht[3] + t.hyy : t.Ht[NU + i];
                else if (i == 1 && j == 0) e = t.hxy;
                const double v = bcast(e, jinit);
                pnew = (rr == NU + i && cc == NU + j) ? v : pnew;
            }
            const double g = bcast(t.gt[NU + i], jinit);
            pnew = (rr == NU + i && cc == 7) ? g : pnew;
        }
    } else {
#pragma unroll
        for (int i = 0; i < NX; ++i) {
#pragma unroll
            for (int j = 0; j <= i; ++j) pnew = (rr == NU + i && cc == NU + j) ? cP[tri(i, j)] : pnew;
            pnew = (rr == NU + i && cc == 7) ? cp[i] : pnew;
        }
    }
    for (int khi = nst - 1; khi >= 0; khi -= CH) {
        const int klo = (khi - CH + 1 > 0) ? khi - CH + 1 : 0;
        lds_order();
        if (lane >= klo && lane <= khi) {
            ldouble *w = stg + (lane - klo) * SPR_REC;
            w[0] = t.ab.a02; w[1] = t.ab.a03; w[2] = t.ab.a04; w[3] = t.ab.a12; w[4] = t.ab.a13; w[5] = t.ab.a14;
            w[6] = t.ab.a23; w[7] = t.ab.a24; w[8] = t.ab.b00; w[9] = t.ab.b10; w[10] = t.ab.b20; w[11] = t.ab.b01;
            w[12] = t.ab.b11; w[13] = t.ab.b21;
            w[14] = hB; w[15] = 0.0; w[16] = 1.0;
#pragma unroll
            for (int q = 0; q < NZ; ++q) w[17 + q] = (q == 2) ? t.Ht[2] + t.hxx : (q == 3) ? t.Ht[3] + t.hyy : t.Ht[q];
            w[24] = t.hxy;
#pragma unroll
            for (int i = 0; i < NX; ++i) w[25 + i] = t.rb[i];
#pragma unroll
            for (int q = 0; q < NZ; ++q) w[30 + q] = t.gt[q];
        }
        lds_order();
        lchar *base = (lchar *)stg + (khi - klo) * SPR_REC * 8;
        double bc[NX], br[NX], ht;
#pragma unroll
        for (int i = 0; i < NX; ++i) { bc[i] = *(const ldouble *)(base + oc[i]); br[i] = *(const ldouble *)(base + orr[i]); }
        ht = *(const ldouble *)(base + off_h);
        for (int k = khi; k >= klo; --k) {
            // the operands of the next stage are fetched while this one computes (reading them at the top of
            // their own step instead, behind the broadcast of P, measured 2 % of an iteration slower)
            double nbc[NX], nbr[NX], nht;
            const lchar *nb = (k > klo) ? base - SPR_REC * 8 : base;
#pragma unroll
            for (int i = 0; i < NX; ++i) { nbc[i] = *(const ldouble *)(nb + oc[i]); nbr[i] = *(const ldouble *)(nb + orr[i]); }
            nht = *(const ldouble *)(nb + off_h);
            if (SUM) sum = lane_above<true>(sum, 0.0) + t.ssum;
            double pn[NX];
#pragma unroll
            for (int i = 0; i < NX; ++i) pn[i] = spr_perm(src_p[i], pnew);
            double P[15];
#pragma unroll
            for (int i = 0; i < NX; ++i)
#pragma unroll
                for (int j = 0; j <= i; ++j) P[tri(i, j)] = bcast(pnew, 8 * (NU + i) + (NU + j));
            double pba[NX];
#pragma unroll
            for (int i = 0; i < NX; ++i) {
                double acc = psym(P, i, 0) * bc[0];
#pragma unroll
                for (int j = 1; j < NX; ++j) acc = fma(psym(P, i, j), bc[j], acc);
                pba[i] = pn[i] + acc;      // (adding -0 changes no number)
            }
            double m = ht;
#pragma unroll
            for (int i = 0; i < NX; ++i) m = fma(br[i], pba[i], m);
            const double m00 = bcast(m, 0), m10 = bcast(m, 8), m11 = bcast(m, 9);
            const double mr0 = spr_perm(src_r0, m), mr1 = spr_perm(src_r1, m), mc0 = spr_perm(src_c0, m),
                         mc1 = spr_perm(src_c1, m);
            __builtin_amdgcn_sched_barrier(0);      // (the exchanges are under way while the pivots are inverted)
            const double d0 = 1.0 / m00;
            const double l10 = m10 * d0;
            const double d1 = 1.0 / fma(-l10, m10, m11);
            const double y1 = fma(-l10, mc0, mc1);
            const double z0 = mc0 * d0, z1 = y1 * d1;
            const double s0 = fma(-l10, z1, z0);
            const double k0j = -s0, k1j = -z1;
            const double e = fma(mr0, k0j, m);
            const double pk = fma(mr1, k1j, e);
            pnew = (lane == 63) ? -0.0 : pk;
            *(ldouble *)(base + oA) = diag ? k0j : kvl ? m : l10;
            *(ldouble *)(base + oB) = diag ? k1j : (lane == 1) ? d1 : d0;
#pragma unroll
            for (int i = 0; i < NX; ++i) { bc[i] = nbc[i]; br[i] = nbr[i]; }
            ht = nht;
            base -= SPR_REC * 8;
        }
        lds_order();
        if (lane >= klo && lane <= khi) {
            const ldouble *w = stg + (lane - klo) * SPR_REC;
#pragma unroll
            for (int i = 0; i < NX; ++i) { ro.K[0][i] = w[i]; ro.K[1][i] = w[5 + i]; }
            ro.l10 = w[10]; ro.d0 = w[11]; ro.d1 = w[12]; ro.kv[0] = w[13]; ro.kv[1] = w[14];
        }
    }
    lds_order();
}


// one block of 64 consecutive stages (k0 .. k0+63) of sweep B0; what the block above handed
// down, and the running reductions, are read from and left in the carry words
template <int NH, bool ONE>
__device__ __attribute__((noinline)) void coop_B0_block(COOP_PARAMS, gopts *op, bool sp_have, int sp_mode,
                                                        double sp_sigma_mu, double sp_alpha, int k0_spread, ldouble *ret)
{
    COOP_UNPACK(NH);
    const mpcx_opts ol = opts_in_registers(op);
    const mpcx_opts &o = ol;
    const int k0 = k0_spread & 0x3fffffff;
    const bool spread = ((k0_spread >> 30) & 1) != 0;      // (bit 30: the recursion spread over the lanes, not as a pipeline)
    Step sp;
    sp.have = sp_have; sp.mode = sp_mode; sp.sigma_mu = sp_sigma_mu; sp.alpha = sp_alpha;
    const int N = __builtin_amdgcn_readfirstlane(C.N);      // (the same in every lane: loop counts in scalar registers)
    const int j = (int)threadIdx.x;
    ldouble *carry = ret - j + Spr<NH>::CARRY;      // same address in every lane
    const int kj = k0 + j;
    const Lane S = stage_lane(C, (unsigned)kj);
    // the stage above supplies its updated state and multiplier: the lane above, or the
    // lowest stage of the block above
    const KR kr = {kj == 0, kj == N};
    B0In<NH> ld = {};
    SUB_BEGIN();
    double cx[NX], d[NX];
#pragma unroll
    for (int i = 0; i < NX; ++i) { cx[i] = 0.0; d[i] = 0.0; }
    if (kj <= N) {
        stage_B0_load<NH, ANY>(S, 0, sp, ld, kr);
        stage_B0_apply<NH, ANY>(S, 0, sp, ld, kr);
        stage_B0_step_rows<NH, ANY>(S, o, pc, 0, sp, ld, cx, kr);
    }
    const int steps = (N - k0 < 64) ? N - k0 : 64;       // stages below N in this block
    if (__builtin_amdgcn_readfirstlane((int)sp.have) != 0) {
        // the step of the dynamics multipliers: a chain down the stages, run as a pipeline over the lanes
        // (the lane of stage N starts it; the end lane takes what the block above handed down)
        double dn[NX];
#pragma unroll
        for (int i = 0; i < NX; ++i) { dn[i] = carry[CY_DPI + i]; d[i] = (kj == N) ? cx[i] : 0.0; }
        lds_order();
        for (int s = 0; s < steps; ++s) {
#pragma unroll
            for (int i = 0; i < NX; ++i) dn[i] = lane_above_keep<ONE>(dn[i], d[i]);
            if (kj < N) dpi_chain(ld.ab, cx, dn, d);
        }
        if (j == 0) {
#pragma unroll
            for (int i = 0; i < NX; ++i) carry[CY_DPI + i] = d[i];
        }
    }
    if (kj <= N) stage_B0_apply_pi<NH, ANY>(S, 0, sp, ld, d, kr);
    double vxn[NX], qpin[NX];
#pragma unroll
    for (int i = 0; i < NX; ++i) {
        vxn[i] = lane_above<ONE>(ld.v[NU + i], carry[CY_VX + i]);
        qpin[i] = lane_above<ONE>(ld.qpi[i], carry[CY_QPI + i]);
    }
    lds_order();
#pragma unroll
    for (int i = 0; i < NX; ++i) {
        const double a = bcast(ld.v[NU + i], 0), b = bcast(ld.qpi[i], 0);
        if (j == 0) { carry[CY_VX + i] = a; carry[CY_QPI + i] = b; }
    }
    B0Out t = {};
    if (kj <= N) stage_B0_tail<NH, ANY>(S, o, pc, 0, sp, ld, vxn, qpin, t, kr);
    SUB_END(16);
    // reductions over the stages of the block, downwards
    const int kend = (N - k0 < 63) ? N - k0 : 63;
    lds_order();
    double ng = carry[CY_NORM + 0], nb = carry[CY_NORM + 1], nd = carry[CY_NORM + 2], nm = carry[CY_NORM + 3],
           sum = carry[CY_NORM + 4];
    ng = dmax(ng, wave_max(t.ng)); nb = dmax(nb, wave_max(t.nb));       // (zero in lanes without a stage)
    nd = dmax(nd, wave_max(t.nd)); nm = dmax(nm, wave_max(t.nm));
    {
        // the sum is a chain from stage kend down to the block's first: the same pipeline as
        // the recursions (lanes above kend keep handing down the value the chain starts from)
        // (they hold s = +0, and no partial sum is ever -0, so adding their zero changes nothing)
        double in = sum;
        for (int s = 0; s <= kend; ++s) {
            in = lane_above_keep<ONE>(in, sum);
            sum = in + t.ssum;
        }
        sum = bcast(sum, 0);
    }
    SUB_END(17);
    // Riccati recursion down the lanes; cP, cp: P and p of the lowest stage of the block above
    double cP[15], cp[NX];
#pragma unroll
    for (int i = 0; i < 15; ++i) cP[i] = carry[CY_P + i];
#pragma unroll
    for (int i = 0; i < NX; ++i) cp[i] = carry[CY_PV + i];
    lds_order();
    if (j == 0) {
        carry[CY_NORM + 0] = ng; carry[CY_NORM + 1] = nb; carry[CY_NORM + 2] = nd; carry[CY_NORM + 3] = nm;
        carry[CY_NORM + 4] = sum;
    }
    RicOut ro = {};
    if (spread) {
        // (the staging area is the result slots and what follows them: nothing of this sweep is there yet)
        double pk, none = 0.0;
        ric_spread<Spr<NH>::CH_BLOCK, false>(t, S.hB, steps, (N - k0 < 64) ? N - k0 : -1, cP, cp, ret - j, ro, none, pk);
        if (kj < N) feed_forward(ro.l10, ro.d0, ro.d1, ro.kv[0], ro.kv[1], ro.kv);
        SUB_END(18);
        // P and p of the block's lowest stage, each entry from the lane that holds it
        int cw = -1;
#pragma unroll
        for (int i = 0; i < NX; ++i) {
#pragma unroll
            for (int jj = 0; jj <= i; ++jj) cw = (j == 8 * (NU + i) + NU + jj) ? CY_P + tri(i, jj) : cw;
            cw = (j == 8 * (NU + i) + 7) ? CY_PV + i : cw;
        }
        if (cw >= 0) carry[cw] = pk;
    } else {
    double Po[15], po[NX];
#pragma unroll
    for (int i = 0; i < NX; ++i) po[i] = 0.0;
#pragma unroll
    for (int i = 0; i < 15; ++i) Po[i] = 0.0;
    if (kj == N) stage_B0_ric<LAST>(S, t, Po, po, Po, po, ro);
    double Pi[15], pi[NX];
#pragma unroll
    for (int i = 0; i < 15; ++i) Pi[i] = cP[i];
#pragma unroll
    for (int i = 0; i < NX; ++i) pi[i] = cp[i];
    for (int s = 0; s < steps; ++s) {
#pragma unroll
        for (int i = 0; i < 15; ++i) Pi[i] = lane_above_keep<ONE>(Pi[i], Po[i]);
#pragma unroll
        for (int i = 0; i < NX; ++i) pi[i] = lane_above_keep<ONE>(pi[i], po[i]);
        if (kj < N) stage_B0_ric<MID, false>(S, t, Pi, pi, Po, po, ro);
    }
    if (kj < N) feed_forward(ro.l10, ro.d0, ro.d1, ro.kv[0], ro.kv[1], ro.kv);
    SUB_END(18);
    if (j == 0) {
#pragma unroll
        for (int i = 0; i < 15; ++i) carry[CY_P + i] = Po[i];
#pragma unroll
        for (int i = 0; i < NX; ++i) carry[CY_PV + i] = po[i];
    }
    }
    lds_order();
    if (kj < N) store_ric<NH, ANY>(S, 0, true, ro);
    SUB_END(19);
}

template <int NH>
DEV void coop_B0(const Lane &C, const PC<NH> &pc, gopts *op, const Step &sp, int m_total, int spread_on, ldouble *ret)
{
    const int sb = spread_on ? (1 << 30) : 0;
    const int N = __builtin_amdgcn_readfirstlane(C.N);      // (the same in every lane: loop counts in scalar registers)
    ldouble *carry = ret - threadIdx.x + Spr<NH>::CARRY;
    if (threadIdx.x < CY_N) carry[threadIdx.x] = 0.0;
    lds_order();
    // blocks of 64 consecutive stages, the last block first
    if (N < 64) {      // a single block: everything handed down from above is zero
        coop_B0_block<NH, true>(COOP_PASS(C, pc), op, sp.have, sp.mode, sp.sigma_mu, sp.alpha, sb, ret);
    } else {
        for (int k0 = (N / 64) * 64; k0 >= 0; k0 -= 64)
            coop_B0_block<NH, false>(COOP_PASS(C, pc), op, sp.have, sp.mode, sp.sigma_mu, sp.alpha, k0 | sb, ret);
    }
    lds_order();
    const double ng = carry[CY_NORM + 0], nb = carry[CY_NORM + 1], nd = carry[CY_NORM + 2], nm = carry[CY_NORM + 3],
                 sum = carry[CY_NORM + 4];
    ret[0 * 64] = ng; ret[1 * 64] = nb; ret[2 * 64] = nd; ret[3 * 64] = nm;
    ret[4 * 64] = sum / (double)m_total;
}

template <int NH, bool ONE>
__device__ __attribute__((noinline)) void coop_Bs(COOP_PARAMS, gopts *op, double sigma_mu)
{
    COOP_UNPACK(NH);
    (void)op;
    const int N = __builtin_amdgcn_readfirstlane(C.N);      // (the same in every lane: loop counts in scalar registers)
    const int j = (int)threadIdx.x;
    double c1[NX], c2[NX];          // p of the lowest stage of the block above, both systems
#pragma unroll
    for (int i = 0; i < NX; ++i) { c1[i] = 0.0; c2[i] = 0.0; }
    for (int k0 = ONE ? 0 : (N / 64) * 64; k0 >= 0; k0 -= 64) {
        const int kj = k0 + j;
        const Lane S = stage_lane(C, (unsigned)kj);
        BsOut t = {};
        const KR kr = {kj == 0, kj == N};
        SUB_BEGIN();
        if (kj <= N) stage_Bs_rows<NH, ANY>(S, 0, sigma_mu, t, kr);
        SUB_END(24);
        double o1[NX], o2[NX], kvo[2][NU] = {{0.0, 0.0}, {0.0, 0.0}};
#pragma unroll
        for (int i = 0; i < NX; ++i) { o1[i] = 0.0; o2[i] = 0.0; }
        if (kj == N) stage_Bs_rec<LAST>(S, t, o1, o2, o1, o2, kvo);
        const int steps = (N - k0 < 64) ? N - k0 : 64;
        double i1[NX], i2[NX];
#pragma unroll
        for (int i = 0; i < NX; ++i) { i1[i] = c1[i]; i2[i] = c2[i]; }
        for (int s = 0; s < steps; ++s) {
#pragma unroll
            for (int i = 0; i < NX; ++i) {
                i1[i] = lane_above_keep<ONE>(i1[i], o1[i]);
                i2[i] = lane_above_keep<ONE>(i2[i], o2[i]);
            }
            if (kj < N) stage_Bs_rec<MID, false>(S, t, i1, i2, o1, o2, kvo);
        }
        if (kj < N) {
            feed_forward(t.l10, t.d0, t.d1, kvo[0][0], kvo[0][1], kvo[0]);
            feed_forward(t.l10, t.d0, t.d1, kvo[1][0], kvo[1][1], kvo[1]);
        }
#pragma unroll
        for (int i = 0; i < NX; ++i) { c1[i] = bcast(o1[i], 0); c2[i] = bcast(o2[i], 0); }
        SUB_END(25);
        if (kj < N) store_rec<NH, ANY>(S, 0, true, kvo);
    }
}

template <int NH, int NS, bool ONE>
__device__ __attribute__((noinline)) double coop_F(COOP_PARAMS, gopts *op, double sigma_mu, ldouble *ret)
{
    COOP_UNPACK(NH);
    const mpcx_opts ol = opts_in_registers(op);
    const mpcx_opts &o = ol;
    const int N = __builtin_amdgcn_readfirstlane(C.N);      // (the same in every lane: loop counts in scalar registers)
    const int j = (int)threadIdx.x;
    double cdx[NS][NX];          // state step that enters the first stage of the block
#pragma unroll
    for (int q = 0; q < NS; ++q)
#pragma unroll
        for (int i = 0; i < NX; ++i) cdx[q][i] = 0.0;
    double alpha[2] = {1.0, 1.0}, s1 = 0.0, s2 = 0.0;
    for (int k0 = 0; ONE ? (k0 == 0) : (k0 <= N); k0 += 64) {
        const int kj = k0 + j;
        const Lane S = stage_lane(C, (unsigned)kj);
        // everything the sweep reads of this lane's stage, in one round trip
        const KR kr = {kj == 0, kj == N};
        FIn<NH, NS> ld = {};
        SUB_BEGIN();
        if (kj <= N) stage_F_load<NH, ANY, NS>(S, 0, ld, kr);
        // the recursion up the lanes: dxi = state step entering this lane's stage, dxo = leaving it
        const int kend = (N - k0 < 63) ? N - k0 : 63;
        double dxi[NS][NX], dxo[NS][NX];
#pragma unroll
        for (int q = 0; q < NS; ++q)
#pragma unroll
            for (int i = 0; i < NX; ++i) { dxi[q][i] = cdx[q][i]; dxo[q][i] = 0.0; }
        for (int s = 0; s <= kend; ++s) {
#pragma unroll
            for (int q = 0; q < NS; ++q) {
#pragma unroll
                for (int i = 0; i < NX; ++i) dxi[q][i] = lane_below_keep<ONE>(dxi[q][i], dxo[q][i]);
                if (kj < N) {
                    double du[NU];
                    F_rec(C, ld.ab, ld.kg, ld.kv[q], ld.rb, dxi[q], du, dxo[q]);
                }
            }
        }
#pragma unroll
        for (int q = 0; q < NS; ++q)
#pragma unroll
            for (int i = 0; i < NX; ++i) cdx[q][i] = bcast(dxo[q][i], 63);
        SUB_END(26);
        FAcc a = {{1.0, 1.0}, 0.0, 0.0};
        if (kj <= N) stage_F_compute<NH, ANY, NS>(S, o, pc, 0, sigma_mu, ld, dxi, a, kr);
#pragma unroll
        for (int q = 0; q < NS; ++q) alpha[q] = dmin(alpha[q], wave_min(a.alpha[q]));      // (1 in lanes without a stage)
        if (NS == 1) {      // chains from the block's first stage up to stage kend
            // (lanes without a stage add +0; no partial sum is ever -0, so that changes nothing)
            double i1 = s1, i2 = s2;
            for (int s = 0; s <= kend; ++s) {
                i1 = lane_below_keep<ONE>(i1, s1); i2 = lane_below_keep<ONE>(i2, s2);
                s1 = i1 + a.s1;
                s2 = i2 + a.s2;
            }
            s1 = bcast(s1, kend); s2 = bcast(s2, kend);
        }
        SUB_END(27);
    }
    if (NS == 1) { ret[0 * 64] = s1; ret[1 * 64] = s2; }
    else ret[0 * 64] = alpha[1];
    return alpha[0];
}

// ---------------------------------------------------------------------------
// A BURST of interior-point iterations of one problem (N < 64), the wavefront with a lane per
// stage and the stage RESIDENT in the lane's registers: the iterate, the linearisation, the
// multipliers and the pending step are loaded once, every iteration of the burst hands its
// results to the next in registers (the four sweeps of an iteration are one instruction stream
// here, their row parts and lane pipelines the same code as in coop_B0_block / coop_F / coop_Bs),
// and the stage goes back to the workspace once at the end.  Between the first load and the
// last store a burst touches no memory but the problem constants in LDS.  The operations and
// their order per problem are those of the separate sweeps.
// Results (wave-uniform) are left in the carry words behind the sweeps' result slots.
// ---------------------------------------------------------------------------
// what a lane of the solve kernel is doing (see the kernel below)
enum LaneState {
    LS_FETCH = 0,   // needs a problem
    LS_LIN = 1,     // needs sweep L (new SQP iteration)
    LS_QP = 2,      // inside the interior point: one iteration per tick
    LS_OUT = 3,     // problem ended, results not written yet
    LS_DONE = 4     // no problems left
};

enum { BR_STATE = 0, BR_STATUS, BR_ADD_TOTAL, BR_ADD_SQP, BR_QIT, BR_HAVE, BR_MODE, BR_SIGMA_MU, BR_ALPHA, BR_DONE, BR_N };
static_assert(BR_N <= CY_N, "burst results fit the carry words");
static_assert(MPCX_STORE_RECIP == 0, "the resident burst keeps the slack reciprocals in registers itself");

template <int NH>
__device__ __attribute__((noinline)) void coop_burst(COOP_PARAMS, gopts *op, int sp_have_mode,
                                                     double sp_sigma_mu, double sp_alpha, int m_total, int qit_max,
                                                     ldouble *ret)
{
    // (two small integers per argument: the arguments then fit the registers of the calling convention)
    const bool sp_have = ((sp_have_mode >> 8) & 1) != 0;
    const int sp_mode = sp_have_mode & 255;
    int qit = qit_max & 0xffff;
    const int max_iters = (qit_max >> 16) & 0x3fff;
    const bool spread = ((qit_max >> 30) & 1) != 0;      // (the sign bit: the kernel has the staging area and the caller wants it used)
    typedef Map<NH> M;
    COOP_UNPACK(NH);
    const mpcx_opts ol = opts_in_registers(op);
    const mpcx_opts &o = ol;
    const int N = __builtin_amdgcn_readfirstlane(C.N);      // (the same in every lane: loop counts in scalar registers)
    const int j = (int)threadIdx.x;
    const int kj = j;
    const bool on = kj <= N;
    const Lane S = stage_lane(C, (unsigned)kj);
    const KR kr = {kj == 0, kj == N};
    ldouble *carry = ret - j + Spr<NH>::CARRY;      // same address in every lane
    Step sp;
    sp.have = sp_have; sp.mode = sp_mode; sp.sigma_mu = sp_sigma_mu; sp.alpha = sp_alpha;
    // ---- this lane's stage ----
    B0In<NH> R = {};
    if (on) stage_B0_load<NH, ANY>(S, 0, sp, R, kr);
    // the reciprocals of the slacks stay in registers with the stage (the sweeps recompute them where needed)
#pragma unroll
    for (int r = 0; r < M::R; ++r) { R.til_[r] = 1.0 / R.tl[r]; R.tiu_[r] = 1.0 / R.tu[r]; }
    int n_state = LS_QP, n_status = -1, add_total = 0, add_sqp = 0, done = 0;
    SUB_BEGIN();
    SUB_END(27);
    for (;;) {
        // ================= B0: pending step, residuals, barrier Hessian, factorisation =================
        if (on) stage_B0_apply<NH, ANY, false>(S, 0, sp, R, kr);
        {
            double cx[NX], d[NX];
#pragma unroll
            for (int i = 0; i < NX; ++i) { cx[i] = 0.0; d[i] = 0.0; }
            if (on) stage_B0_step_rows<NH, ANY, false, true>(S, o, pc, 0, sp, R, cx, kr);
            if (__builtin_amdgcn_readfirstlane((int)sp.have) != 0) {
                // the step of the dynamics multipliers: a chain from stage N down, as a pipeline over the lanes
                AB ac = R.ab;
                fresh_ab(ac);
#pragma unroll
                for (int i = 0; i < NX; ++i) { cx[i] = fresh(cx[i]); d[i] = (kj == N) ? cx[i] : 0.0; }
                for (int s = 0; s < N; ++s) {
                    double dn[NX];
#pragma unroll
                    for (int i = 0; i < NX; ++i) dn[i] = lane_above<true>(d[i], 0.0);
                    if (kj < N) dpi_chain(ac, cx, dn, d);
                }
            }
            if (on) stage_B0_apply_pi<NH, ANY, false>(S, 0, sp, R, d, kr);
        }
        double vxn[NX], qpin[NX];
#pragma unroll
        for (int i = 0; i < NX; ++i) {
            vxn[i] = lane_above<true>(R.v[NU + i], 0.0);
            qpin[i] = lane_above<true>(R.qpi[i], 0.0);
        }
        B0Out t = {};
        if (on) stage_B0_tail<NH, ANY, false, true>(S, o, pc, 0, sp, R, vxn, qpin, t, kr);
#pragma unroll
        for (int jj = 0; jj < NZ; ++jj) R.vo[jj] = R.v[jj];      // the iterate the next pending step starts from
#pragma unroll
        for (int i = 0; i < NX; ++i) R.rgo[i] = t.rg[NU + i];    // ... and the stationarity residual it leaves
        const int kend = N;
        const double Rg = uniform(wave_max(t.ng)), Rb = uniform(wave_max(t.nb)), Rd = uniform(wave_max(t.nd)),
                     Rm = uniform(wave_max(t.nm));
        // The complementarity sum is a chain from stage N down to stage 0, 0 + s(N) + s(N-1) + ...; it rides
        // along with the Riccati pipeline below, which hands values down the lanes in the same direction.
        // (lanes without a stage hold s = +0 and hand down +0; no partial sum is ever -0, so adding their
        // zero changes nothing)
        double sum = 0.0 + t.ssum;
        SUB_END(16);
        double Po[15], po[NX];
        RicOut ro = {};
#pragma unroll
        for (int i = 0; i < NX; ++i) po[i] = 0.0;
#pragma unroll
        for (int i = 0; i < 15; ++i) Po[i] = 0.0;
        // ================= end of the QP? =================
        int qs = -1;
        if (!(finite_d(Rg) && finite_d(Rb) && finite_d(Rd) && finite_d(Rm))) qs = 3;
        else if (Rg < o.qp_tol_stat && Rb < o.qp_tol_eq && Rd < o.qp_tol_ineq && Rm < o.qp_tol_comp) qs = 0;
        else if (qit >= o.max_qp_iter) qs = 1;
        if (qs >= 0) {
            add_total = qit;
            add_sqp = 1;
            if (qs == 3) { n_status = MPCX_QP_FAILURE; n_state = LS_OUT; }
            else n_state = LS_LIN;
            break;
        }
        // ================= Riccati factorisation and the affine backward substitution =================
        if (spread) {
            double pn0;
            ric_spread<Spr<NH>::CH_BURST, true>(t, S.hB, N, N, nullptr, nullptr, ret - j, ro, sum, pn0);
        } else {
        if (kj == N) stage_B0_ric<LAST>(S, t, Po, po, Po, po, ro);
        // the factorisation down the lanes ...
        for (int s = 0; s < N; ++s) {
            double Pi[15];
#pragma unroll
            for (int i = 0; i < 15; ++i) Pi[i] = lane_above<true>(Po[i], 0.0);
            sum = lane_above<true>(sum, 0.0) + t.ssum;
            if (kj < N) stage_B0_ric<MID, false, 1>(S, t, Pi, nullptr, Po, nullptr, ro);
        }
        // ... then, with P(k+1) rb formed once from the final P(k+1), the affine right-hand side
        {
            double Pi[15];
#pragma unroll
            for (int i = 0; i < 15; ++i) Pi[i] = lane_above<true>(Po[i], 0.0);
            if (kj < N) ric_prb(t, Pi, ro.prb);
            B0Out tp = t;
            fresh_ab(tp.ab);
#pragma unroll
            for (int jj = 0; jj < NZ; ++jj) tp.gt[jj] = fresh(tp.gt[jj]);
            RicOut rp = ro;
#pragma unroll
            for (int i = 0; i < NX; ++i) { rp.prb[i] = fresh(rp.prb[i]); rp.K[0][i] = fresh(rp.K[0][i]); rp.K[1][i] = fresh(rp.K[1][i]); }
            for (int s = 0; s < N; ++s) {
                double pi[NX];
#pragma unroll
                for (int i = 0; i < NX; ++i) pi[i] = lane_above<true>(po[i], 0.0);
                if (kj < N) stage_B0_ric<MID, false, 2>(S, tp, nullptr, pi, nullptr, po, rp);
            }
            ro.kv[0] = rp.kv[0]; ro.kv[1] = rp.kv[1];
        }
        }
        if (kj < N) feed_forward(ro.l10, ro.d0, ro.d1, ro.kv[0], ro.kv[1], ro.kv);
        const double mu = uniform(bcast(sum, 0) / (double)m_total);
        SUB_END(17);
        // ================= F (affine): predictor step, its length, centring sums =================
        double a_aff, S1, S2;
        {
            FIn<NH, 1> ld = {};
#pragma unroll
            for (int i = 0; i < NX; ++i) { ld.x[i] = R.x[i]; ld.rb[i] = t.rb[i]; ld.kg[0][i] = ro.K[0][i]; ld.kg[1][i] = ro.K[1][i]; }
            ld.u[0] = R.u[0]; ld.u[1] = R.u[1];
            ld.kv[0][0] = ro.kv[0]; ld.kv[0][1] = ro.kv[1];
#pragma unroll
            for (int jj = 0; jj < NZ; ++jj) { ld.v[jj] = R.v[jj]; ld.dva[jj] = 0.0; }
#pragma unroll
            for (int r = 0; r < M::R; ++r) {
                ld.ll[r] = R.ll[r]; ld.lu[r] = R.lu[r]; ld.tl[r] = R.tl[r]; ld.tu[r] = R.tu[r];
                ld.til_[r] = R.til_[r]; ld.tiu_[r] = R.tiu_[r];
            }
            ld.ab = R.ab;
            double dxi[1][NX], dxo[1][NX];
#pragma unroll
            for (int i = 0; i < NX; ++i) { dxi[0][i] = 0.0; dxo[0][i] = 0.0; }
            fresh_ab(ld.ab);
#pragma unroll
            for (int i = 0; i < NX; ++i) { ld.kg[0][i] = fresh(ld.kg[0][i]); ld.kg[1][i] = fresh(ld.kg[1][i]); ld.rb[i] = fresh(ld.rb[i]); }
            ld.kv[0][0] = fresh(ld.kv[0][0]); ld.kv[0][1] = fresh(ld.kv[0][1]);
            for (int s = 0; s <= kend; ++s) {
#pragma unroll
                for (int i = 0; i < NX; ++i) dxi[0][i] = lane_below<true>(dxo[0][i], 0.0);
                {      // (every lane: what stage N and the lanes without a stage produce is never used)
                    double du[NU];
                    F_rec(C, ld.ab, ld.kg, ld.kv[0], ld.rb, dxi[0], du, dxo[0]);
                }
            }
            FAcc a = {{1.0, 1.0}, 0.0, 0.0};
            double dvo[1][NZ];
#pragma unroll
            for (int jj = 0; jj < NZ; ++jj) dvo[0][jj] = 0.0;
            if (on) stage_F_compute<NH, ANY, 1, false, true, false>(S, o, pc, 0, 0.0, ld, dxi, a, kr, dvo);
#pragma unroll
            for (int jj = 0; jj < NZ; ++jj) R.dva[jj] = dvo[0][jj];
            a_aff = uniform(wave_min(a.alpha[0]));
            double s1 = 0.0, s2 = 0.0;
            for (int s = 0; s <= kend; ++s) {      // chains from stage 0 up to stage N
                s1 = lane_below<true>(s1, 0.0) + a.s1;
                s2 = lane_below<true>(s2, 0.0) + a.s2;
            }
            S1 = uniform(bcast(s1, kend));
            S2 = uniform(bcast(s2, kend));
        }
        SUB_END(24);
        double sigma_mu;
        {
            double S0 = mu * (double)m_total;
            double mu_aff = fma(a_aff, fma(a_aff, S2, S1), S0) / (double)m_total;
            if (mu_aff < 0.0) mu_aff = 0.0;
            double ratio = mu_aff / mu;
            double sigma = ratio * ratio * ratio;
            sigma_mu = sigma * mu;
        }
        // ================= Bs: corrector and centring-only right-hand sides, substituted backwards =================
        double o1[NX], o2[NX], kvo[2][NU] = {{0.0, 0.0}, {0.0, 0.0}};
        {
            BsIn bin;
            BsOut t2 = {};
#pragma unroll
            for (int i = 0; i < NX; ++i) { t2.k0[i] = ro.K[0][i]; t2.k1[i] = ro.K[1][i]; }
#pragma unroll
            for (int jj = 0; jj < NZ; ++jj) { bin.ga[jj] = 0.0; bin.gb[jj] = 0.0; }
            if (on) stage_Bs_sums<NH, ANY>(S, pc, R, bin, kr);
            t2.ab = R.ab;
            t2.l10 = (kj < N) ? ro.l10 : 0.0; t2.d0 = (kj < N) ? ro.d0 : 1.0; t2.d1 = (kj < N) ? ro.d1 : 1.0;
            stage_Bs_compute(sigma_mu, bin, t2);
#pragma unroll
            for (int i = 0; i < NX; ++i) { o1[i] = 0.0; o2[i] = 0.0; }
            if (kj == N) stage_Bs_rec<LAST>(S, t2, o1, o2, o1, o2, kvo);
            fresh_ab(t2.ab);
#pragma unroll
            for (int i = 0; i < NX; ++i) { t2.k0[i] = fresh(t2.k0[i]); t2.k1[i] = fresh(t2.k1[i]); }
#pragma unroll
            for (int jj = 0; jj < NZ; ++jj) { t2.gt1[jj] = fresh(t2.gt1[jj]); t2.gt2[jj] = fresh(t2.gt2[jj]); }
            t2.l10 = fresh(t2.l10);
            for (int s = 0; s < N; ++s) {
                double i1[NX], i2[NX];
#pragma unroll
                for (int i = 0; i < NX; ++i) {
                    i1[i] = lane_above<true>(o1[i], 0.0);
                    i2[i] = lane_above<true>(o2[i], 0.0);
                }
                if (kj < N) stage_Bs_rec<MID, false>(S, t2, i1, i2, o1, o2, kvo);
            }
            if (kj < N) {
                feed_forward(t2.l10, t2.d0, t2.d1, kvo[0][0], kvo[0][1], kvo[0]);
                feed_forward(t2.l10, t2.d0, t2.d1, kvo[1][0], kvo[1][1], kvo[1]);
            }
        }
        SUB_END(25);
        // ================= F (both): corrector and centring-only steps, their lengths =================
        double alpha1, alpha2;
        double dvo[2][NZ];
        {
            FIn<NH, 2> ld = {};
#pragma unroll
            for (int i = 0; i < NX; ++i) {
                ld.x[i] = R.x[i]; ld.rb[i] = 0.0; ld.kg[0][i] = ro.K[0][i]; ld.kg[1][i] = ro.K[1][i];
            }
            ld.u[0] = R.u[0]; ld.u[1] = R.u[1];
            ld.kv[0][0] = kvo[0][0]; ld.kv[0][1] = kvo[0][1]; ld.kv[1][0] = kvo[1][0]; ld.kv[1][1] = kvo[1][1];
#pragma unroll
            for (int jj = 0; jj < NZ; ++jj) { ld.v[jj] = R.v[jj]; ld.dva[jj] = R.dva[jj]; }
#pragma unroll
            for (int r = 0; r < M::R; ++r) {
                ld.ll[r] = R.ll[r]; ld.lu[r] = R.lu[r]; ld.tl[r] = R.tl[r]; ld.tu[r] = R.tu[r];
                ld.til_[r] = R.til_[r]; ld.tiu_[r] = R.tiu_[r];
            }
            ld.ab = R.ab;
            double dxi[2][NX], dxo[2][NX];
#pragma unroll
            for (int q = 0; q < 2; ++q)
#pragma unroll
                for (int i = 0; i < NX; ++i) { dxi[q][i] = 0.0; dxo[q][i] = 0.0; }
            fresh_ab(ld.ab);
#pragma unroll
            for (int i = 0; i < NX; ++i) { ld.kg[0][i] = fresh(ld.kg[0][i]); ld.kg[1][i] = fresh(ld.kg[1][i]); ld.rb[i] = fresh(ld.rb[i]); }
#pragma unroll
            for (int q = 0; q < 2; ++q) { ld.kv[q][0] = fresh(ld.kv[q][0]); ld.kv[q][1] = fresh(ld.kv[q][1]); }
            for (int s = 0; s <= kend; ++s) {
#pragma unroll
                for (int q = 0; q < 2; ++q) {
#pragma unroll
                    for (int i = 0; i < NX; ++i) dxi[q][i] = lane_below<true>(dxo[q][i], 0.0);
                    {      // (every lane, as in the affine sweep)
                        double du[NU];
                        F_rec(C, ld.ab, ld.kg, ld.kv[q], ld.rb, dxi[q], du, dxo[q]);
                    }
                }
            }
            FAcc a = {{1.0, 1.0}, 0.0, 0.0};
#pragma unroll
            for (int q = 0; q < 2; ++q) {
#pragma unroll
                for (int jj = 0; jj < NZ; ++jj) dvo[q][jj] = 0.0;
            }
            if (on) stage_F_compute<NH, ANY, 2, false, true>(S, o, pc, 0, sigma_mu, ld, dxi, a, kr, dvo);
            alpha1 = uniform(wave_min(a.alpha[0]));
            alpha2 = uniform(wave_min(a.alpha[1]));
        }
        SUB_END(26);
        // ================= the step taken =================
        const bool use2 = o.cond_pred_corr && (alpha1 < a_aff);
        const double alpha = use2 ? alpha2 : alpha1;
        ++qit;
        ++done;
#pragma unroll
        for (int jj = 0; jj < NZ; ++jj) R.dv[jj] = use2 ? dvo[1][jj] : dvo[0][jj];
        if (!(alpha > o.alpha_min)) {      // minimum step or NaN
            add_total = qit;
            add_sqp = 1;
            n_status = MPCX_QP_FAILURE;
            n_state = LS_OUT;
            break;
        }
        sp.have = true;
        sp.mode = use2 ? 2 : 1;
        sp.sigma_mu = sigma_mu;
        sp.alpha = alpha * fma(alpha, 0.9999999, (1.0 - alpha) * 0.99);
        if (done >= max_iters) break;
    }
    // ---- the stage goes back to the workspace: the iterate and the pending step, where the sweeps expect them ----
    SUB_END(27);
    if (on) {
        const int sol = (sp.mode == 2) ? M::SOL2 : 0;
#pragma unroll
        for (int jj = 0; jj < NZ; ++jj) {
            stw<ANY>(S, 0, M::V + jj, R.v[jj]);
            stw<ANY>(S, 0, M::DVA + jj, R.dva[jj]);
            *wsp(S, 0, M::DV + jj, 512u * (unsigned)sol) = R.dv[jj];
        }
#pragma unroll
        for (int i = 0; i < NX; ++i) {
            stw<ANY>(S, 0, M::QPI + i, R.qpi[i]);
            stw<ANY>(S, 0, M::RG + i, R.rgo[i]);
        }
#pragma unroll
        for (int r = 0; r < M::R; ++r) {
            stw<ANY>(S, 0, M::LAML + r, R.ll[r]);
            stw<ANY>(S, 0, M::LAMU + r, R.lu[r]);
            stw<ANY>(S, 0, M::TL + r, R.tl[r]);
            stw<ANY>(S, 0, M::TU + r, R.tu[r]);
        }
    }
    lds_order();
    if (j == 0) {
        carry[BR_STATE] = (double)n_state; carry[BR_STATUS] = (double)n_status; carry[BR_ADD_TOTAL] = (double)add_total;
        carry[BR_ADD_SQP] = (double)add_sqp; carry[BR_QIT] = (double)qit; carry[BR_HAVE] = sp.have ? 1.0 : 0.0;
        carry[BR_MODE] = (double)sp.mode; carry[BR_SIGMA_MU] = sp.sigma_mu; carry[BR_ALPHA] = sp.alpha;
        carry[BR_DONE] = (double)done;
    }
    lds_order();
    SUB_END(30);
}

// sweep L has no recursion: every stage is linearised at its own (new) state, which its lane
// forms from the stage's own iterate and step exactly as the stage before would have
template <int NH, bool ONE>
__device__ __attribute__((noinline)) void coop_L(COOP_PARAMS, gopts *op, bool first, bool zero_u, double alpha, ldouble *ret)
{
    typedef Map<NH> M;
    COOP_UNPACK(NH);
    const mpcx_opts ol = opts_in_registers(op);
    const mpcx_opts &o = ol;
    const int N = __builtin_amdgcn_readfirstlane(C.N);      // (the same in every lane: loop counts in scalar registers)
    const int j = (int)threadIdx.x;
    coop_sync();
    LAcc tot = {0.0, 0.0, 0.0, 0.0, 0.0};
    for (int k0 = 0; ONE ? (k0 == 0) : (k0 <= N); k0 += 64) {
        const int kj = k0 + j;
        const Lane S = stage_lane(C, (unsigned)kj);
        // everything a lane reads of its own and of the next stage, before any lane stores
        double xk[NX], pik[NX], nxt[3 * NX];
#pragma unroll
        for (int i = 0; i < NX; ++i) { xk[i] = 0.0; pik[i] = 0.0; nxt[i] = nxt[NX + i] = nxt[2 * NX + i] = 0.0; }
        if (kj <= N) {
#pragma unroll
            for (int i = 0; i < NX; ++i) {
                const double x0 = ldw<ANY>(S, 0, M::X + i), v0 = ldw<ANY>(S, 0, M::V + NU + i), q0 = ldw<ANY>(S, 0, M::QPI + i);
                xk[i] = first ? x0 : x0 + alpha * v0;
                pik[i] = (kj > 0 && !first) ? q0 : 0.0;
            }
        }
        if (kj < N) {
#pragma unroll
            for (int i = 0; i < NX; ++i) {
                const Lane S1 = stage_lane(C, (unsigned)kj + 1u);
                nxt[i] = ldw<ANY>(S1, 0, M::X + i);
                nxt[NX + i] = ldw<ANY>(S1, 0, M::V + NU + i);
                nxt[2 * NX + i] = ldw<ANY>(S1, 0, M::QPI + i);
            }
        }
        coop_sync();
        LAcc a = {0.0, 0.0, 0.0, 0.0, 0.0};
        const KR kr = {kj == 0, kj == N};
        if (kj <= N) stage_L<NH, ANY, true>(S, o, pc, 0, first, zero_u, alpha, xk, pik, a, kr, nxt);
        const int kend = (N - k0 < 63) ? N - k0 : 63;
        tot.rs = dmax(tot.rs, wave_max(a.rs)); tot.re = dmax(tot.re, wave_max(a.re));
        tot.ri = dmax(tot.ri, wave_max(a.ri)); tot.rc = dmax(tot.rc, wave_max(a.rc));
        {
            const double start = tot.cost;
            double c = start;
            for (int s = 0; s <= kend; ++s) {
                const double in = lane_below<ONE>(c, start);
                c = (j <= kend) ? in + a.cost : in;
            }
            tot.cost = bcast(c, kend);
        }
    }
    ret[0 * 64] = tot.rs; ret[1 * 64] = tot.re; ret[2 * 64] = tot.ri; ret[3 * 64] = tot.rc;
    ret[4 * 64] = tot.cost;
}

// sweep M of one problem, a lane per stage (the line search's look at a trial point, see stage_M)
template <int NH, bool ONE>
__device__ __attribute__((noinline)) void coop_M(COOP_PARAMS, double alpha, ldouble *ret)
{
    typedef Map<NH> M;
    COOP_UNPACK(NH);
    const int N = __builtin_amdgcn_readfirstlane(C.N);      // (the same in every lane: loop counts in scalar registers)
    const int j = (int)threadIdx.x;
    coop_sync();
    MAcc tot = {0.0, 0.0, 0.0, 0.0};
    for (int k0 = 0; ONE ? (k0 == 0) : (k0 <= N); k0 += 64) {
        const int kj = k0 + j;
        const Lane S = stage_lane(C, (unsigned)kj);
        double xk[NX], qk[NX], nxt[2 * NX];
#pragma unroll
        for (int i = 0; i < NX; ++i) { xk[i] = 0.0; qk[i] = 0.0; nxt[i] = nxt[NX + i] = 0.0; }
        if (kj <= N) {
#pragma unroll
            for (int i = 0; i < NX; ++i) {
                const double x0 = ldw<ANY>(S, 0, M::X + i), v0 = ldw<ANY>(S, 0, M::V + NU + i);
                xk[i] = x0 + alpha * v0;
                qk[i] = ldw<ANY>(S, 0, M::QPI + i);
            }
        }
        if (kj < N) {
            const Lane S1 = stage_lane(C, (unsigned)kj + 1u);
#pragma unroll
            for (int i = 0; i < NX; ++i) {
                nxt[i] = ldw<ANY>(S1, 0, M::X + i);
                nxt[NX + i] = ldw<ANY>(S1, 0, M::V + NU + i);
            }
        }
        MAcc a = {0.0, 0.0, 0.0, 0.0};
        const KR kr = {kj == 0, kj == N};
        if (kj <= N) stage_M<NH, ANY, true>(S, pc, 0, alpha, xk, a, kr, nxt, qk);
        const int kend = (N - k0 < 63) ? N - k0 : 63;
        tot.gap = dmax(tot.gap, wave_max(a.gap));
        tot.viol = dmax(tot.viol, wave_max(a.viol));
        {
            // the two sums: chains from the block's first stage up to stage kend
            const double s1 = tot.cost, s2 = tot.msum;
            double c1 = s1, c2 = s2;
            for (int s = 0; s <= kend; ++s) {
                const double i1 = lane_below<ONE>(c1, s1), i2 = lane_below<ONE>(c2, s2);
                c1 = (j <= kend) ? i1 + a.cost : i1;
                c2 = (j <= kend) ? i2 + a.msum : i2;
            }
            tot.cost = bcast(c1, kend);
            tot.msum = bcast(c2, kend);
        }
    }
    ret[0 * 64] = tot.cost; ret[1 * 64] = tot.gap; ret[2 * 64] = tot.viol; ret[3 * 64] = tot.msum;
}

// ---------------------------------------------------------------------------
// the fused solve.  Every lane runs a small state machine over its own problem;
// the wavefront executes the sweeps in a fixed order each TICK and a lane joins
// the sweeps its state calls for.  A lane whose problem ends takes the next
// unsolved problem from a global counter, so lanes stay busy although problems
// need very different numbers of iterations.  Which lane solves which problem
// does not influence any result.
// ---------------------------------------------------------------------------


// Write-out, fetch and sweep L ("service") run with few lanes each, so they are held
// back until this many lanes wait for them (or no lane has interior-point work).
#ifndef MPCX_SERVICE_MIN
#define MPCX_SERVICE_MIN 8
#endif
// (coop_max: with this many problems or fewer in progress the wavefront takes them one at a
// time in single-problem mode.)
// (disable_tail_calls: a call that is marked as a possible tail call loses the treatment of local
// functions whose callers are all known, in which no register is callee-saved; the sweeps
// would then save and restore over a hundred registers through scratch memory on every call)
template <int NH>
__global__ void __launch_bounds__(64) __attribute__((disable_tail_calls))
mpcx_sqp_kernel(mpcx_batch in, const mpcx_opts *op_generic, mpcx_result out, double *workspace, int *counter, int lanes, int coop_max, int lin_max, int svc_min,
                int boost_min, int boost_div, int boost_cap, int boost_lanes, int coop_layout, int slots, unsigned deal, int burst_few,
                double hold_alpha, int hold_extra, int boost_mul, int spread_on, unsigned char *ho, int n_workers, int ho_limit, int ho_late)
{
    const mpcx_opts &o = *op_generic;
    gopts *op = (gopts *)op_generic;      // what the sweeps are handed
    typedef Map<NH> M;
    const int N = in.N;
    const unsigned B = (unsigned)in.B;
    // stage records of the [stage][field][lane] layout, then `slots` single-problem regions
    const size_t wave_doubles = (size_t)((N + 1) + slots * ((N + 64) / 64)) * M::F * 64;

    // constants, 5 result slots, the rest of the staging area of ric_spread(), carry words
    __shared__ double pc_lds[(PCN<NH>::N + 6) * 64 + Spr<NH>::EXTRA];
    PC<NH> pc;
    pc.p = (ldouble *)pc_lds + threadIdx.x;
    pc.lmk = 0.0;
    ldouble *ret = (ldouble *)pc_lds + PCN<NH>::N * 64 + threadIdx.x;     // results of the sweeps

    Lane L;
    L.wave = (gdouble *)workspace + (size_t)blockIdx.x * wave_doubles;
    L.pdata = (const gdouble *)in.pdata;
    L.B = (size_t)in.B;
    L.lane = 8u * threadIdx.x;
    L.region = coop_layout ? (unsigned)(((N + 64) / 64) * M::F * 512) : 0u;
    L.roff = 0;
    L.b = 0;
    L.N = N;
    L.F = M::F;
    L.nobs = 0;
    L.h = L.hh = L.h6 = L.hB = L.invL = 0.0;

    const bool zero_u = (in.uguess == nullptr);      // no input guess: a problem's first linearisation takes u = 0
    int state = ((int)threadIdx.x < lanes) ? LS_FETCH : LS_DONE;
    int status = MPCX_MAXITER, sqp_it = 0, qp_it = 0, qp_total = 0, m_total = 0;
    bool first = true, fresh = false;
    double res[4] = {0.0, 0.0, 0.0, 0.0};
    double cost = 0.0;
    double ls_rho = 0.0;      // line search (mpcx_opts.globalization): penalty weight of this lane's problem
    Step sp;
    sp.have = false; sp.mode = 0; sp.sigma_mu = 0.0; sp.alpha = 0.0;

    STAMP_DECL();
#ifdef MPCX_STAMPS
    unsigned long long t_fetch = 0;
#endif
    if (threadIdx.x == 0) MARK(21);
    // what a lane decides after sweep L of its problem (not in real-time-iteration mode's second call)
    auto after_L = [&]() {
        first = false;
        state = LS_OUT;
        if (!(finite_d(res[0]) && finite_d(res[1]) && finite_d(res[2]) && finite_d(res[3]))) {
            status = MPCX_NAN;
        } else if (!o.rti && res[0] < o.tol_stat && res[1] < o.tol_eq && res[2] < o.tol_ineq &&
                   res[3] < o.tol_comp) {
            status = MPCX_SUCCESS;
        } else if (sqp_it >= o.max_sqp_iter) {
            status = MPCX_MAXITER;
        } else {
            state = LS_QP;
            sp.have = false;
            qp_it = 0;
        }
    };
    // Line search (globalization = 1), one trial: (c, re, ri, ms) = sweep M at step length alpha; rho, phi0
    // are set by the first trial.  Returns true when alpha is taken, else alpha has been shortened.
    auto ls_trial = [&](int trial, double c, double re, double ri, double ms, double cost0, double viol0,
                        double &rho, double &phi0, double &alpha) -> bool {
        if (trial == 0) {
            const double want = o.ls_rho_factor * ms;
            if (want > rho) rho = want;
            phi0 = fma(rho, viol0, cost0);
        }
        const double phi = fma(rho, dmax(re, ri), c);
        if (phi <= phi0 - ((o.ls_c1 * alpha) * rho) * viol0) return true;
        if (alpha * o.ls_shrink < o.ls_alpha_min) return true;
        alpha = alpha * o.ls_shrink;
        return false;
    };
    // sweep L of the problem of lane p, the wavefront with a lane per stage; results into lane p
    auto single_L = [&](int p) {
        Lane C;
        PC<NH> pcc;
        problem_of_lane<NH>(L, pc, (ldouble *)pc_lds, p, C, pcc);
        const bool fp = __builtin_amdgcn_readlane((int)first, p) != 0;
        double alpha = 1.0;
        if (o.globalization && !o.rti && !fp) {
            // the step of the QP just solved is tried at shorter and shorter lengths
            double rho = bcast(ls_rho, p), phi0 = 0.0;
            const double cost0 = bcast(cost, p), viol0 = dmax(bcast(res[1], p), bcast(res[2], p));
            for (int trial = 0;; ++trial) {
                if (N < 64) coop_M<NH, true>(COOP_PASS(C, pcc), alpha, ret);
                else coop_M<NH, false>(COOP_PASS(C, pcc), alpha, ret);
                const double c = uniform(ret[0 * 64]), re = uniform(ret[1 * 64]), ri = uniform(ret[2 * 64]),
                             ms = uniform(ret[3 * 64]);
                if (ls_trial(trial, c, re, ri, ms, cost0, viol0, rho, phi0, alpha)) break;
            }
            if ((int)threadIdx.x == p) ls_rho = rho;
        }
        SUB_BEGIN();
        if (N < 64) coop_L<NH, true>(COOP_PASS(C, pcc), op, fp, zero_u, alpha, ret);
        else coop_L<NH, false>(COOP_PASS(C, pcc), op, fp, zero_u, alpha, ret);
        SUB_END(28);
        const double r0 = uniform(ret[0 * 64]), r1 = uniform(ret[1 * 64]), r2 = uniform(ret[2 * 64]),
                     r3 = uniform(ret[3 * 64]), r4 = uniform(ret[4 * 64]);
        // (the second sweep L of a real-time iteration only applies the step and reports the objective:
        // the residuals handed back are those of the test before the QP)
        if ((int)threadIdx.x == p) {
            if (!(o.rti && !fp)) { res[0] = r0; res[1] = r1; res[2] = r2; res[3] = r3; }
            cost = r4;
        }
    };
    // one interior-point iteration of the problem of lane p, the wavefront with a lane per stage
    auto single_qp = [&](int p) {
        SUB_BEGIN();
        coop_sync();
        Lane C;
        PC<NH> pcc;
        problem_of_lane<NH>(L, pc, (ldouble *)pc_lds, p, C, pcc);
        Step s;
        s.have = __builtin_amdgcn_readlane((int)sp.have, p) != 0;
        s.mode = __builtin_amdgcn_readlane(sp.mode, p);
        s.sigma_mu = bcast(sp.sigma_mu, p);
        s.alpha = bcast(sp.alpha, p);
        const int mt = __builtin_amdgcn_readlane(m_total, p);
        int qit = __builtin_amdgcn_readlane(qp_it, p);
        int n_state = LS_QP, n_status = -1, add_total = 0, add_sqp = 0;
        coop_B0<NH>(C, pcc, op, s, mt, spread_on, ret);
        SUB_END(31);
        STAMP(12, true);
        const double Rg = uniform(ret[0 * 64]), Rb = uniform(ret[1 * 64]), Rd = uniform(ret[2 * 64]),
                     Rm = uniform(ret[3 * 64]), mu = uniform(ret[4 * 64]);
        int qs = -1;
        if (!(finite_d(Rg) && finite_d(Rb) && finite_d(Rd) && finite_d(Rm))) qs = 3;
        else if (Rg < o.qp_tol_stat && Rb < o.qp_tol_eq && Rd < o.qp_tol_ineq && Rm < o.qp_tol_comp) qs = 0;
        else if (qit >= o.max_qp_iter) qs = 1;
        if (qs >= 0) {
            add_total = qit;
            add_sqp = 1;
            if (qs == 3) { n_status = MPCX_QP_FAILURE; n_state = LS_OUT; }
            else n_state = LS_LIN;
        } else {
            const double a_aff = uniform((N < 64) ? coop_F<NH, 1, true>(COOP_PASS(C, pcc), op, 0.0, ret)
                                                  : coop_F<NH, 1, false>(COOP_PASS(C, pcc), op, 0.0, ret));
            STAMP(13, true);
            const double S1 = uniform(ret[0 * 64]), S2 = uniform(ret[1 * 64]);
            double S0 = mu * (double)mt;
            double mu_aff = fma(a_aff, fma(a_aff, S2, S1), S0) / (double)mt;
            if (mu_aff < 0.0) mu_aff = 0.0;
            double ratio = mu_aff / mu;
            double sigma = ratio * ratio * ratio;
            const double sigma_mu = sigma * mu;
            if (N < 64) coop_Bs<NH, true>(COOP_PASS(C, pcc), op, sigma_mu);
            else coop_Bs<NH, false>(COOP_PASS(C, pcc), op, sigma_mu);
            STAMP(14, true);
            const double alpha1 = uniform((N < 64) ? coop_F<NH, 2, true>(COOP_PASS(C, pcc), op, sigma_mu, ret)
                                                   : coop_F<NH, 2, false>(COOP_PASS(C, pcc), op, sigma_mu, ret));
            const double alpha2 = uniform(ret[0 * 64]);
            STAMP(15, true);
            const bool use2 = o.cond_pred_corr && (alpha1 < a_aff);
            const double alpha = use2 ? alpha2 : alpha1;
            ++qit;
            if (!(alpha > o.alpha_min)) {
                add_total = qit;
                add_sqp = 1;
                n_status = MPCX_QP_FAILURE;
                n_state = LS_OUT;
            } else {
                s.have = true;
                s.mode = use2 ? 2 : 1;
                s.sigma_mu = sigma_mu;
                s.alpha = alpha * fma(alpha, 0.9999999, (1.0 - alpha) * 0.99);
            }
        }
        if ((int)threadIdx.x == p) {
            state = n_state;
            if (n_status >= 0) status = n_status;
            qp_total += add_total;
            sqp_it += add_sqp;
            qp_it = qit;
            sp = s;
        }
    };

    // up to max_iters interior-point iterations of the problem of lane p in one go, its stages resident
    // in the lanes' registers (N < 64); returns the iterations done
    auto single_burst = [&](int p, int max_iters) -> int {
        coop_sync();
        Lane C;
        PC<NH> pcc;
        problem_of_lane<NH>(L, pc, (ldouble *)pc_lds, p, C, pcc);
        const int have = __builtin_amdgcn_readlane((int)sp.have, p) != 0;
        const int mode = __builtin_amdgcn_readlane(sp.mode, p);
        const double sgm = bcast(sp.sigma_mu, p), alp = bcast(sp.alpha, p);
        const int mt = __builtin_amdgcn_readlane(m_total, p);
        const int qit0 = __builtin_amdgcn_readlane(qp_it, p);
        if (max_iters > 0x3fff) max_iters = 0x3fff;
        if (max_iters < 1) max_iters = 1;
        coop_burst<NH>(COOP_PASS(C, pcc), op, (have << 8) | (mode & 255), sgm, alp, mt,
                       (int)(((unsigned)spread_on << 30) | ((unsigned)max_iters << 16) | ((unsigned)qit0 & 0xffffu)), ret);
        ldouble *carry = ret - threadIdx.x + Spr<NH>::CARRY;
        const int n_state = (int)uniform(carry[BR_STATE]), n_status = (int)uniform(carry[BR_STATUS]),
                  add_total = (int)uniform(carry[BR_ADD_TOTAL]), add_sqp = (int)uniform(carry[BR_ADD_SQP]),
                  qit = (int)uniform(carry[BR_QIT]), done = (int)uniform(carry[BR_DONE]);
        Step s;
        s.have = uniform(carry[BR_HAVE]) != 0.0;
        s.mode = (int)uniform(carry[BR_MODE]);
        s.sigma_mu = uniform(carry[BR_SIGMA_MU]);
        s.alpha = uniform(carry[BR_ALPHA]);
        if ((int)threadIdx.x == p) {
            state = n_state;
            if (n_status >= 0) status = n_status;
            qp_total += add_total;
            sqp_it += add_sqp;
            qp_it = qit;
            sp = s;
        }
        STAMP_N(12, 64 * (done + (n_state != LS_QP ? 1 : 0)));
        return done;
    };

    // Regions: `slots` per wavefront, behind the stage records.  A problem that the wavefront will
    // only work on in single-problem mode from now on is moved to a free one (`solo` lane): the
    // last problems once the queue is empty, and old problems that get extra iterations.
    bool moved = false;          // the queue is empty and the last problems have been moved
    // Hand-over of old problems (ho != nullptr).  A wavefront that finds one of its problems old moves it to one
    // of its regions as before, but instead of giving it bursts itself it writes what the lane knows of the
    // problem into a descriptor and queues the descriptor's number; the lane is free for the next problem.
    // Worker wavefronts -- the last n_workers of the grid from the start, every other wavefront once its own
    // problems are done -- take descriptors from the queue and run each problem to its end in bursts, on the
    // region where it lies (the workspace pointer of the wavefront is that of the problem's first owner
    // meanwhile).  So no problem waits for a neighbour's bursts, old problems are balanced over the wavefronts
    // whatever wavefront fetched them, and the longest problem has a wavefront of its own from its 16th
    // iteration on.  The workers of the first kind leave when the queue is empty and no wavefront that could still
    // hand a problem over is left (a count of the wavefronts that have problems of their own, in the hand-over
    // area); the others when they find the queue empty.
    unsigned *const ho_head = (unsigned *)ho, *const ho_tail = (unsigned *)(ho ? ho + 128 : nullptr),
                   *const ho_active = (unsigned *)(ho ? ho + 256 : nullptr), *const ho_ring = (unsigned *)(ho ? ho + MPCX_HO_RING : nullptr),
                   *const ho_busy = (unsigned *)(ho ? ho + MPCX_HO_BUSY : nullptr);
    double *const ho_desc = (double *)(ho ? ho + MPCX_HO_DESC : nullptr);
    const bool born_worker = ho != nullptr && (int)blockIdx.x >= (int)gridDim.x - n_workers;
    bool worker = born_worker;
    int job = -1;                // worker: the descriptor it is working on
    unsigned given = 0;          // slots of this wavefront whose problems are with a worker (same in all lanes)
    // A batch where most problems grow old is better served by the wavefronts that fetched them (every hand-over
    // is a write-back of the L2 at both ends): a wavefront hands over only while at most a quarter of the
    // problems it has fetched have grown old, and a wavefront that was a worker from the start and has found the
    // queue empty for 60 ms while there are still problems to fetch takes problems of its own.
    unsigned n_got = 0, n_old = 0;
    unsigned long long idle_since = 0;
    gdouble *const own_wave = L.wave;
    if (worker) state = LS_DONE;
    bool solo = false;           // this lane's problem lives in a region; the one-lane-per-problem sweeps skip it
    unsigned slot_used = 0;      // same in all lanes
    const unsigned rec_bytes = (unsigned)(M::F * 512), nblk = (unsigned)((N + 64) / 64);
    auto move_to_region = [&](int p) -> bool {
        if (__builtin_amdgcn_readlane((int)L.roff, p) != 0) return true;
        const unsigned free_slots = ~slot_used & ((slots >= 32) ? 0xffffffffu : ((1u << slots) - 1u));
        if (free_slots == 0) return false;
        const int slot = __ffs((int)free_slots) - 1;
        slot_used |= 1u << slot;
        Lane C;
        problem_home(L, p, C);
        Lane R = C;
        R.region = region_bytes(L);
        R.lane = ((unsigned)(N + 1) + (unsigned)slot * nblk) * rec_bytes;
        SUB_BEGIN();
        coop_sync();
        for (int k0 = 0; k0 <= N; k0 += 64) {
            const int kj = k0 + (int)threadIdx.x;
            if (kj > N) continue;
            const Lane S = stage_lane(C, (unsigned)kj), D = stage_lane(R, (unsigned)kj);
            // what is alive between two interior-point iterations: the linearisation, the iterate, the
            // stationarity residual and the pending step (fields X .. RG and DV .. the end); the other
            // residuals, factors and right-hand sides are formed anew by the next iteration's first sweep
            // before anything reads them.  In the
            // [stage][field][lane] layout every word of a problem sits in a cache line of its own, so a
            // word not copied is a line not fetched.
            for (int part = 0; part < 2; ++part) {
                const int lo = part ? M::DV : M::X, hi = part ? M::F : M::RG + NX;
                for (int f0 = lo; f0 < hi; f0 += 16) {
                    double tmp[16];
#pragma unroll
                    for (int f = 0; f < 16; ++f) tmp[f] = (f0 + f < hi) ? ldw<ANY>(S, 0, f0 + f) : 0.0;
#pragma unroll
                    for (int f = 0; f < 16; ++f)
                        if (f0 + f < hi) stw<ANY>(D, 0, f0 + f, tmp[f]);
                }
            }
        }
        if ((int)threadIdx.x == p) { L.roff = R.lane; solo = true; }
        coop_sync();
        SUB_END(29);
        return true;
    };
    for (;;) {
        STAMP_BEGIN();
        // ---- the queue is empty and few problems are left: from here on the wavefront stays in
        // single-problem mode, so each problem is moved to a region where its stages are contiguous
        if (!moved && slots > 0 && L.region == 0 && !o.rti &&
            __hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= (int)B) {
            const bool active = (state == LS_QP) || (state == LS_LIN) || (state == LS_OUT);
            const unsigned long long amask = __ballot(active);
            if (__popcll(amask) <= slots && __popcll(amask) <= coop_max) {
                moved = true;
                for (unsigned long long rest = amask; rest != 0; rest &= rest - 1)
                    (void)move_to_region(__ffsll((long long)rest) - 1);
            }
        }
        // ---- service: results out, next problem in, new SQP iteration ---------------------
        const bool wants = (state == LS_OUT) || (state == LS_FETCH) || (state == LS_LIN);
        const int n_wants = __popcll(__ballot(wants));
        const int n_qp = __popcll(__ballot(state == LS_QP));
        const bool few = n_qp <= coop_max;     // single-problem mode ahead
        if (n_wants >= svc_min || n_qp == 0 || (few && n_wants > 0)) {
            [[maybe_unused]] const bool st_e = (state == LS_OUT);
            // the trajectory of each finished problem is copied out by the whole wavefront, a lane
            // per stage: one round trip to memory instead of one per stage
            const unsigned long long out_mask = __ballot(state == LS_OUT);
            if (out_mask != 0) coop_sync();
            for (unsigned long long rest = out_mask; rest != 0; rest &= rest - 1) {
                const int p = __ffsll((long long)rest) - 1;
                const size_t b = (unsigned)__builtin_amdgcn_readlane((int)L.b, p);
                Lane C;
                problem_home(L, p, C);
                for (int k0 = 0; k0 <= N; k0 += 64) {
                    const int kj = k0 + (int)threadIdx.x;
                    if (kj > N) continue;
                    const Lane S = stage_lane(C, (unsigned)kj);
                    double xo[NX], uo[NU] = {0.0, 0.0};
#pragma unroll
                    for (int i = 0; i < NX; ++i) xo[i] = ldw<ANY>(S, 0, M::X + i);
                    if (kj < N) { uo[0] = ldw<ANY>(S, 0, M::U + 0); uo[1] = ldw<ANY>(S, 0, M::U + 1); }
#pragma unroll
                    for (int i = 0; i < NX; ++i) out.X[(b * (size_t)(N + 1) + kj) * NX + i] = xo[i];
                    if (kj < N) {
                        out.U[(b * (size_t)N + kj) * NU + 0] = uo[0];
                        out.U[(b * (size_t)N + kj) * NU + 1] = uo[1];
                    }
                }
            }
            {
                // regions of the problems just written out are free again
                const unsigned long long freed = __ballot(state == LS_OUT && L.roff != 0);
                for (unsigned long long rest = freed; rest != 0; rest &= rest - 1) {
                    const unsigned r = (unsigned)__builtin_amdgcn_readlane((int)L.roff, __ffsll((long long)rest) - 1);
                    slot_used &= ~(1u << ((r / rec_bytes - (unsigned)(N + 1)) / nblk));
                }
            }
            if (state == LS_OUT) {
                const size_t b = L.b;
#ifdef MPCX_STAMPS
                if (qp_total >= 5000 && g_stamps[blockIdx.x][41] == 0) {      // the end of a very long problem, and when it was fetched
                    g_stamps[blockIdx.x][41] = wall_clock64();
                    g_stamps[blockIdx.x][43] = (unsigned long long)qp_total;
                    g_stamps[blockIdx.x][45] = t_fetch;
                    g_stamps[blockIdx.x][47] = (unsigned long long)b;
                }
                // the last problem the wavefront writes out: when, how long, fetched when
                g_stamps[blockIdx.x][49] = wall_clock64();
                g_stamps[blockIdx.x][51] = (unsigned long long)qp_total;
                g_stamps[blockIdx.x][53] = t_fetch;
#endif
                L.roff = 0;
                solo = false;
                out.status[b] = status;
                out.sqp_iter[b] = sqp_it;
                out.qp_iter[b] = qp_total;
                out.cost[b] = cost;
#pragma unroll
                for (int i = 0; i < 4; ++i) out.res[(size_t)i * L.B + b] = res[i];
                state = LS_FETCH;
            }
            if (ho != nullptr && out_mask != 0) {
                if (job >= 0) {      // a worker's problem has been written out: its first owner gets the region back
                    if (threadIdx.x == 0) state = LS_DONE;
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
                    if (threadIdx.x == 0) __hip_atomic_store(ho_busy + job, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    job = -1;
                    L.wave = own_wave;
                }
            }
            STAMP(8, st_e);
            [[maybe_unused]] const bool st_a = (state == LS_FETCH);
            if (state == LS_FETCH) {
                // Fetch order.  The batch is taken in the caller's order in blocks of 65536 problems;
                // inside a block the c-th fetch takes problem (c * g) mod (block size), g coprime to the
                // size and near its golden-section point, so consecutive fetches lie far apart: problems
                // that are neighbours in the caller's order (and often alike in length) end up in
                // different wavefronts.  Results are written at the problem's own index.
                // A caller's order (in.order) replaces all of this.  It is a priority list: its front is taken
                // first, and since a wavefront takes `lanes` consecutive tickets at a time the list is dealt so
                // that consecutive entries go to different wavefronts: ticket c takes entry (c mod lanes) * R +
                // c div lanes, R = B div lanes (the B mod lanes entries at the very end keep their place).  The
                // front of the list, where a caller puts the long problems, is then spread one per wavefront
                // instead of filling the first few wavefronts with problems that wait for each other.
                const unsigned c = (unsigned)atomicAdd(counter, 1);
                unsigned b = c;
                if (c < B) {
                    if (in.order != nullptr) {
                        const unsigned per = (unsigned)lanes, R = B / per;
                        const unsigned e = (c < R * per) ? (c % per) * R + c / per : c;
                        b = (unsigned)in.order[e];
                    } else if (deal != 0) {
                        const unsigned full = B & ~65535u;
                        if (c < full) b = (c & ~65535u) | ((c * 40503u) & 65535u);
                        else b = full + (unsigned)(((unsigned long long)(c - full) * deal) % (B - full));
                    }
                }
                if (c >= B) {
                    MARK(20);
                    state = LS_DONE;
                } else if (b >= B) {
                    // an entry of the caller's order that names no problem: the lane asks again next time
                } else {
                    L.b = b;
                    L.roff = 0;
#ifdef MPCX_STAMPS
                    t_fetch = wall_clock64();
#endif
                    int nobs = in.n_obs ? in.n_obs[b] : 0;
                    // (the host validates it; stay memory safe)
                    if (nobs < 0 || nobs > NH) {
                        out.status[b] = MPCX_NAN;
                        out.sqp_iter[b] = 0;
                        out.qp_iter[b] = 0;
                        out.cost[b] = 0.0;
#pragma unroll
                        for (int i = 0; i < 4; ++i) out.res[(size_t)i * L.B + b] = 0.0;
                    } else {
                        L.nobs = nobs;
                        L.h = ldp(L, MPCX_F_H);
                        L.hh = 0.5 * L.h;
                        L.h6 = L.h / 6.0;
                        L.hB = L.h6 * 6.0;
                        L.invL = 1.0 / ldp(L, MPCX_F_L);
                        fill_pc<NH>(L, o, pc);
                        // row sides: stage 0 has the 2 input rows, stages 1..N-1 all rows, stage N 7 rows
                        m_total = 2 * (NU + (N - 1) * (9 + nobs) + 7);
                        status = MPCX_MAXITER;
                        sqp_it = 0;
                        qp_total = 0;
                        ls_rho = 0.0;
                        first = true;
                        state = LS_LIN;
                        fresh = true;
                    }
                }
            }
            // the initial iterate of each new problem is laid down by the whole wavefront, a lane per stage
            const unsigned long long new_mask = __ballot(fresh);
            n_got += (unsigned)__popcll(new_mask);
            fresh = false;
            for (unsigned long long rest = new_mask; rest != 0; rest &= rest - 1) {
                const int p = __ffsll((long long)rest) - 1;
                const size_t b = (unsigned)__builtin_amdgcn_readlane((int)L.b, p);
                Lane C;
                problem_home(L, p, C);
                GuessData G;
                if (!in.xguess)      // no guess handed over: computed here
                    guess_data(in.pdata, nullptr, L.B, b, __builtin_amdgcn_readlane(L.nobs, p), G);
                for (int k0 = 0; k0 <= N; k0 += 64) {
                    const int kj = k0 + (int)threadIdx.x;
                    if (kj > N) continue;
                    const Lane S = stage_lane(C, (unsigned)kj);
                    double xg[NX];
                    if (in.xguess) {
#pragma unroll
                        for (int i = 0; i < NX; ++i) xg[i] = in.xguess[((size_t)kj * NX + i) * L.B + b];
                    } else {
                        guess_node(G, kj, N, xg);
                    }
                    // (the state guess, and the input guess if there is one; the first linearisation takes the
                    // multipliers, and the inputs otherwise, as zero without reading them)
#pragma unroll
                    for (int i = 0; i < NX; ++i) stw<ANY>(S, 0, M::X + i, xg[i]);
                    if (in.uguess) {
                        double ug[NU] = {0.0, 0.0};
                        if (kj < N) {
                            ug[0] = in.uguess[((size_t)kj * NU + 0) * L.B + b];
                            ug[1] = in.uguess[((size_t)kj * NU + 1) * L.B + b];
                        }
                        stw<ANY>(S, 0, M::U + 0, ug[0]);
                        stw<ANY>(S, 0, M::U + 1, ug[1]);
                    }
                }
            }
            if (new_mask != 0) coop_sync();
            STAMP(0, st_a);
            [[maybe_unused]] const bool st_b = (state == LS_LIN);
            // few problems in the wavefront: sweep L of each with a lane per stage
            const unsigned long long lin_mask = __ballot(state == LS_LIN);
            const bool coop_lin = (L.region != 0) ||
                                  (!o.rti && ((__popcll(lin_mask) + n_qp <= coop_max) || (__popcll(lin_mask) <= lin_max)));
            const unsigned long long single_mask = coop_lin ? lin_mask : (lin_mask & __ballot(solo));
            if (single_mask != 0) {
                for (unsigned long long rest = single_mask; rest != 0; rest &= rest - 1) single_L(__ffsll((long long)rest) - 1);
                coop_sync();      // what the stage lanes stored is visible to the problems' own lanes
            }
            double step_alpha = 1.0;
            if (o.globalization && !o.rti) {
                // line search of the lanes whose sweep L runs one lane per problem: sweep M reads only,
                // so the lanes still searching repeat it while the others sit out
                bool searching = (state == LS_LIN) && !first && !coop_lin && !solo;
                double rho = ls_rho, phi0 = 0.0;
                const double cost0 = cost, viol0 = dmax(res[1], res[2]);
                for (int trial = 0; __any(searching); ++trial) {
                    if (searching) {
                        sweep_M<NH>(L, pc, step_alpha, ret);
                        if (ls_trial(trial, ret[0 * 64], ret[1 * 64], ret[2 * 64], ret[3 * 64], cost0, viol0, rho, phi0,
                                     step_alpha)) {
                            searching = false;
                            ls_rho = rho;
                        }
                    }
                }
            }
            if (state == LS_LIN) {
                if (o.rti && !first) {
                    // real-time iteration: apply the one step, report the new objective
                    if (!coop_lin && !solo) {
                        sweep_L<NH>(L, pc, op, false, zero_u, 1.0, ret);
                        cost = ret[4 * 64];
                    }
                    status = MPCX_SUCCESS;
                    state = LS_OUT;
                } else {
                    if (!coop_lin && !solo) {
                        sweep_L<NH>(L, pc, op, first, zero_u, step_alpha, ret);
#pragma unroll
                        for (int i = 0; i < 4; ++i) res[i] = ret[i * 64];
                        cost = ret[4 * 64];
                    }
                    after_L();
                }
            }
            STAMP(1, st_b);
        }
        if (!__any(state != LS_DONE)) {
            if (ho == nullptr) {
                if (threadIdx.x == 0) MARK_LAST(22);
                break;
            }
            // nothing of its own left (and nothing more to hand over): the wavefront works on handed-over problems
            if (!worker) {
                worker = true;
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
                if (threadIdx.x == 0) atomicSub(ho_active, 1u);
            }
            int got = -1;
            if (threadIdx.x == 0) {
                const unsigned h = __hip_atomic_load(ho_head, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                const unsigned t = __hip_atomic_load(ho_tail, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if ((int)(t - h) > 0) {
                    unsigned expect = h;
                    if (__hip_atomic_compare_exchange_strong(ho_head, &expect, h + 1, __ATOMIC_RELAXED, __ATOMIC_RELAXED,
                                                             __HIP_MEMORY_SCOPE_AGENT)) {
                        // (the ring word follows the tail at once; a bounded wait all the same)
                        unsigned v = 0;
                        for (int spin = 0; spin < (1 << 20) && v == 0; ++spin) {
                            v = __hip_atomic_load(ho_ring + (h & (MPCX_HANDOVER_SLOTS - 1)), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                            if (v == 0) __builtin_amdgcn_s_sleep(8);
                        }
                        if (v != 0) {
                            __hip_atomic_store(ho_ring + (h & (MPCX_HANDOVER_SLOTS - 1)), 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                            got = (int)v - 1;
                        }
                    }
                }
            }
            got = __builtin_amdgcn_readfirstlane(got);
            if (got < 0) {
                // The queue is empty.  A wavefront that was a worker from the start stays while any wavefront is left
                // that could hand a problem over (the count is read first); the others have done their share and
                // leave their place to whatever the device has waiting.
                int over = 0;
                if (threadIdx.x == 0) {
                    const unsigned a = __hip_atomic_load(ho_active, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT);
                    const unsigned h = __hip_atomic_load(ho_head, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    const unsigned t = __hip_atomic_load(ho_tail, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    over = ((a == 0u || !born_worker) && h == t) ? 1 : 0;
                }
                if (__builtin_amdgcn_readfirstlane(over) != 0) {
                    if (threadIdx.x == 0) MARK_LAST(22);
                    break;
                }
                if (born_worker && __hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (int)B) {
                    const unsigned long long now = wall_clock64();
                    if (idle_since == 0) idle_since = now;
                    else if (now - idle_since > 6000000ull) {      // 60 ms of the 100 MHz clock
                        idle_since = 0;
                        worker = false;
                        if (threadIdx.x == 0) atomicAdd(ho_active, 1u);
                        state = ((int)threadIdx.x < lanes) ? LS_FETCH : LS_DONE;
                        continue;
                    }
                }
                __builtin_amdgcn_s_sleep(127);
                continue;
            }
            idle_since = 0;
            // the problem moves into lane 0
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            const double *d = ho_desc + (size_t)got * 32;
            const int w = (int)d[20];
            L.wave = (gdouble *)workspace + (size_t)w * wave_doubles;
            job = got;
            if (threadIdx.x == 0) {
                L.b = (unsigned)d[0];
                L.nobs = (int)d[1];
                state = (int)d[2];
                status = (int)d[3];
                sqp_it = (int)d[4];
                qp_it = (int)d[5];
                qp_total = (int)d[6];
                m_total = (int)d[7];
                first = d[8] != 0.0;
                sp.have = d[9] != 0.0;
                sp.mode = (int)d[10];
                sp.sigma_mu = d[11];
                sp.alpha = d[12];
                res[0] = d[13]; res[1] = d[14]; res[2] = d[15]; res[3] = d[16];
                cost = d[17];
                ls_rho = d[18];
                L.roff = (unsigned)d[19];
                solo = true;
                fresh = false;
#ifdef MPCX_STAMPS
                t_fetch = 0;
#endif
                L.h = ldp(L, MPCX_F_H);
                L.hh = 0.5 * L.h;
                L.h6 = L.h / 6.0;
                L.hB = L.h6 * 6.0;
                L.invL = 1.0 / ldp(L, MPCX_F_L);
                fill_pc<NH>(L, o, pc);
            }
            coop_sync();
            continue;
        }

        // ---- few problems left: one interior-point iteration of each, the wavefront on one problem
        const unsigned long long qp_mask = __ballot(state == LS_QP);
        if (__popcll(qp_mask) <= coop_max || L.region != 0) {
            for (unsigned long long rest = qp_mask; rest != 0; rest &= rest - 1) {
                if (N < 64 && burst_few > 0) (void)single_burst(__ffsll((long long)rest) - 1, (job >= 0) ? 4096 : burst_few);
                else single_qp(__ffsll((long long)rest) - 1);
            }
            coop_sync();      // every single-problem sweep starts with one; this one is for the other mode
            STAMP(11, state != LS_DONE);
            continue;
        }

        // ---- one interior-point iteration ---------------------------------------------------
        // The sweeps run with ALL lanes of the wavefront once any lane iterates: a store of 64 lanes
        // covers whole 128-byte lines, while one with a few lanes switched off leaves lines covered in
        // part, and those cost the memory system a read-modify-write each (with 2 to 3 lanes of 64
        // sitting out, as is usual here, sweep B0 took half as long again).  A lane that sits out
        // (waiting for service, or its problem moved to a region of its own) runs the same instructions
        // on its own column of the workspace with no step pending: that rewrites its iterate and its
        // multipliers unchanged -- what sweep L and the write-out read later -- and leaves unspecified
        // values in fields that are dead until the next QP's first sweep writes them again.
        [[maybe_unused]] const bool st_c = (state == LS_QP) && !solo;
        const bool act = (state == LS_QP) && !solo;
        bool iterate = false;
        if (__any(act)) sweep_B0<NH>(L, pc, op, act && sp.have, sp.mode, sp.sigma_mu, sp.alpha, m_total, ret);
        if (act) {
            const double Rg = ret[0 * 64], Rb = ret[1 * 64], Rd = ret[2 * 64], Rm = ret[3 * 64];
            int qs = -1;     // HPIPM-style: 0 solved, 1 iteration limit, 3 NaN
            if (!(finite_d(Rg) && finite_d(Rb) && finite_d(Rd) && finite_d(Rm))) qs = 3;
            else if (Rg < o.qp_tol_stat && Rb < o.qp_tol_eq && Rd < o.qp_tol_ineq && Rm < o.qp_tol_comp) qs = 0;
            else if (qp_it >= o.max_qp_iter) qs = 1;
            if (qs >= 0) {
                qp_total += qp_it;
                ++sqp_it;
                if (qs == 3) { status = MPCX_QP_FAILURE; state = LS_OUT; }
                else state = LS_LIN;          // the next sweep L applies the step
            } else {
                iterate = true;
            }
        }
        STAMP(2, st_c);
        double a_aff = 0.0, sigma_mu = 0.0;
        const bool any_it = __any(iterate);
        if (any_it) {
            const double mu = ret[4 * 64];
            a_aff = sweep_F<NH, 1>(L, pc, op, 0.0, ret);
            const double S1 = ret[0 * 64], S2 = ret[1 * 64];
            double S0 = mu * (double)m_total;
            double mu_aff = fma(a_aff, fma(a_aff, S2, S1), S0) / (double)m_total;
            if (mu_aff < 0.0) mu_aff = 0.0;
            double ratio = mu_aff / mu;
            double sigma = ratio * ratio * ratio;
            sigma_mu = sigma * mu;
        }
        STAMP(3, iterate);
        if (any_it) sweep_Bs<NH>(L, pc, op, sigma_mu);
        STAMP(4, iterate);
        double alpha1_all = 0.0;
        if (any_it) alpha1_all = sweep_F<NH, 2>(L, pc, op, sigma_mu, ret);
        if (iterate) {
            // the corrector step, or the centring-only one where the corrector is the shorter
            const double alpha1 = alpha1_all;
            const double alpha2 = ret[0 * 64];
            const bool use2 = o.cond_pred_corr && (alpha1 < a_aff);
            const double alpha = use2 ? alpha2 : alpha1;
            ++qp_it;
            if (!(alpha > o.alpha_min)) {      // minimum step or NaN
                qp_total += qp_it;
                ++sqp_it;
                status = MPCX_QP_FAILURE;
                state = LS_OUT;
            } else {
                sp.have = true;
                sp.mode = use2 ? 2 : 1;
                sp.sigma_mu = sigma_mu;
                sp.alpha = alpha * fma(alpha, 0.9999999, (1.0 - alpha) * 0.99);
            }
        }
        STAMP(5, iterate);
        // ---- old problems get extra iterations, the wavefront on one problem at a time -------
        // A problem that has already taken many iterations is likely to take many more, and the
        // longest problem of a batch bounds its run time, so the extra iterations grow with the
        // age: age / boost_div per tick of the wavefront, oldest first, at most boost_lanes problems
        // and boost_cap iterations in total.
        // (On a busy GPU a region does not stay in the L2 between two sweeps -- 128 wavefronts per
        // XCD share 4 MB -- so these iterations move about as many bytes as an iteration in the other
        // mode; what they avoid is its latency: a lone lane needs ~0.9 ms per iteration.)
        if (boost_div > 0 && !o.rti) {
            int budget = boost_cap, n_new = 0;
            unsigned long long seen = 0;
            // hand-over: is there room in the queue; and the hand-over itself of lane p's problem, which lies in
            // region `slot` of this wavefront
            auto queue_has_room = [&]() -> bool {
                int waiting = 0;
                if (threadIdx.x == 0)
                    waiting = (int)(__hip_atomic_load(ho_tail, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) -
                                    __hip_atomic_load(ho_head, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
                return __builtin_amdgcn_readfirstlane(waiting) < ho_limit;
            };
            auto give_away = [&](int p, int slot) {
                const int id = (int)blockIdx.x * slots + slot;
                if ((int)threadIdx.x == p) {
                    double *d = ho_desc + (size_t)id * 32;
                    d[0] = (double)L.b; d[1] = (double)L.nobs; d[2] = (double)state; d[3] = (double)status;
                    d[4] = (double)sqp_it; d[5] = (double)qp_it; d[6] = (double)qp_total; d[7] = (double)m_total;
                    d[8] = first ? 1.0 : 0.0; d[9] = sp.have ? 1.0 : 0.0; d[10] = (double)sp.mode;
                    d[11] = sp.sigma_mu; d[12] = sp.alpha;
                    d[13] = res[0]; d[14] = res[1]; d[15] = res[2]; d[16] = res[3];
                    d[17] = cost; d[18] = ls_rho; d[19] = (double)L.roff; d[20] = (double)blockIdx.x;
                    __hip_atomic_store(ho_busy + id, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
                if ((int)threadIdx.x == p) {
                    const unsigned t = atomicAdd(ho_tail, 1u);
                    __hip_atomic_store(ho_ring + (t & (MPCX_HANDOVER_SLOTS - 1)), (unsigned)id + 1u, __ATOMIC_RELEASE,
                                       __HIP_MEMORY_SCOPE_AGENT);
                    // the lane is free; the region stays taken until the worker is done with it
                    state = LS_FETCH;
                    solo = false;
                    L.roff = 0;
                }
                given |= 1u << slot;
            };
            // A problem the wavefront kept for itself (the queue was full, or most of its problems grow old) is
            // offered again: once it is ho_late iterations old -- its bursts, which grow with its age, hold up the
            // wavefront's other problems -- and, when there is nothing left to fetch, whenever the wavefront has
            // more than one such problem: the wavefronts that are done are workers by then.
            const bool late_on = ho != nullptr && ho_late > 0 && job < 0 && slots > 0 && L.region == 0 && N < 64;
            const bool fetch_over = late_on && __hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= (int)B;
            for (;;) {
                const int age = qp_total + ((state == LS_QP) ? qp_it : 0);
                const bool live = ((state == LS_QP) || (state == LS_LIN)) && !((seen >> threadIdx.x) & 1ull);
                // problems in regions take all their iterations here; of the others, the oldest
                // boost_lanes (they move to a region if one is free)
                const unsigned long long sm = __ballot(live && solo);
                // (a QP whose steps have collapsed is about to be given up -- a handful of iterations more --
                // and is not worth a region: it stays with the one-lane-per-problem sweeps a while longer)
                const bool held = (state == LS_QP) && sp.have && (sp.alpha < hold_alpha) && (age < boost_min + hold_extra);
                const unsigned long long cm = (n_new < boost_lanes) ? __ballot(live && !solo && age >= boost_min && !held) : 0ull;
                if ((sm | cm) == 0) break;
                int p = -1, best = -1;
                if (sm != 0) {
                    p = __ffsll((long long)sm) - 1;
                    best = __builtin_amdgcn_readlane(age, p);
                    if (late_on && (best >= ho_late || (fetch_over && __popcll(sm) >= 2)) && queue_has_room()) {
                        const unsigned r = (unsigned)__builtin_amdgcn_readlane((int)L.roff, p);
                        give_away(p, (int)((r / rec_bytes - (unsigned)(N + 1)) / nblk));
                        seen |= 1ull << p;
                        continue;
                    }
                } else {
                    if (budget <= 0) break;
                    for (unsigned long long rest = cm; rest != 0; rest &= rest - 1) {
                        const int i = __ffsll((long long)rest) - 1;
                        const int a = __builtin_amdgcn_readlane(age, i);
                        if (a > best) { best = a; p = i; }
                    }
                    ++n_new;
                    // (handed over while the workers keep up: with ho_limit descriptors waiting the wavefront gives
                    // the problem its bursts itself, as it would without workers)
                    ++n_old;
                    bool hand = ho != nullptr && job < 0 && slots > 0 && L.region == 0 && N < 64 && 4u * n_old <= n_got + 64u;
                    if (hand) hand = queue_has_room();
                    if (hand) {
                        // regions whose problems a worker has finished are free again
                        if (given != 0) {
                            const unsigned mine = (threadIdx.x < 32u && ((given >> threadIdx.x) & 1u))
                                                      ? __hip_atomic_load(ho_busy + blockIdx.x * (unsigned)slots + threadIdx.x,
                                                                          __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
                                                      : 1u;
                            const unsigned back = (unsigned)__ballot(mine == 0u) & given;
                            given &= ~back;
                            slot_used &= ~back;
                        }
                        const unsigned before = slot_used;
                        if (move_to_region(p) && slot_used != before) {
                            give_away(p, __ffs((int)(slot_used & ~before)) - 1);
                            seen |= 1ull << p;
                            continue;
                        }
                    } else if (slots > 0 && L.region == 0) (void)move_to_region(p);
                }
                seen |= 1ull << p;
                int k = (int)(((long long)best * boost_mul) / boost_div);
                if (k > budget) k = budget;
                if (k < 1) k = 1;
                budget -= k;
                if (threadIdx.x == 0 && best >= 1000) MARK(23);
                if (threadIdx.x == 0) {
                    MARK_AGE(12, 13, 1000, best); MARK_AGE(14, 15, 2000, best); MARK_AGE(55, 57, 4000, best); MARK_AGE(59, 61, 8000, best);
                }
                for (int t = 0; t < k;) {
                    const int st = __builtin_amdgcn_readlane(state, p);
                    if (st == LS_QP) {
                        if (N < 64 && burst_few > 0) {
                            const int d = single_burst(p, k - t);
                            t += (d > 0) ? d : 1;
                        } else {
                            single_qp(p);
                            ++t;
                        }
                    } else if (st == LS_LIN) {
                        single_L(p);
                        if ((int)threadIdx.x == p) after_L();
                        ++t;
                    } else break;
                }
            }
            if (seen != 0) coop_sync();
            STAMP(11, seen != 0);
        }
        STAMP(9, state != LS_DONE);      // lanes-sum / 64 = ticks; lanes that still have work
        STAMP(10, true);
    }
}

// workspace header of a launch: problem counter (and the rest of its 256 bytes) zero, options behind it
// (and the counters, the ring and the busy words of the hand-over area)
__global__ void mpcx_header_kernel(int *counter, mpcx_opts *d_opts, mpcx_opts o, unsigned n_own)
{
    if (threadIdx.x < 64) counter[threadIdx.x] = 0;
    if (threadIdx.x == 0) *d_opts = o;
    unsigned *ho = reinterpret_cast<unsigned *>(reinterpret_cast<char *>(counter) + 512);
    for (unsigned i = threadIdx.x; i < MPCX_HO_DESC / 4; i += blockDim.x) ho[i] = (i == 64u) ? n_own : 0u;      // (byte 256: wavefronts with problems of their own)
}

// initial guess (reference core_solver.py:477-499, and :360-473 for problems with obstacles, up to
// MPCX_MAX_OBS each: guess_adjust above): one thread per problem, node by node; every store of a
// wavefront is one contiguous segment of xguess[i][c][:]
__global__ void mpcx_guess_kernel(const double *pdata, const int32_t *n_obs, const double *width, int B, int N,
                                  double *xguess)
{
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    const size_t Bs = (size_t)B;
    GuessData G;
    int nobs = n_obs ? n_obs[b] : 0;
    nobs = (nobs < 0) ? 0 : (nobs > MPCX_MAX_OBS) ? MPCX_MAX_OBS : nobs;
    guess_data(pdata, width, Bs, (size_t)b, nobs, G);
    for (int i = 0; i <= N; ++i) {
        double o[NX];
        guess_node(G, i, N, o);
#pragma unroll
        for (int c = 0; c < NX; ++c) xguess[((size_t)i * NX + c) * Bs + b] = o[c];
    }
}

// Outcome counters of a solved batch: stats[0..4] = problems per status 0..4, [5] = QP solves,
// [6] = interior-point iterations, [7] = linearisations (one after every QP solve that did not
// fail, plus the first).  One grid-stride pass, a shuffle reduction per wavefront, then one
// atomic per counter and wavefront.  `stats` must be zero on entry.
__global__ void __launch_bounds__(256) mpcx_stats_kernel(const int32_t *status, const int32_t *sqp_iter,
                                                         const int32_t *qp_iter, int B, unsigned long long *stats)
{
    unsigned long long acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int b = blockIdx.x * blockDim.x + threadIdx.x; b < B; b += gridDim.x * blockDim.x) {
        const int st = status[b], sq = sqp_iter[b], qp = qp_iter[b];
#pragma unroll
        for (int k = 0; k < 5; ++k) acc[k] += (st == k) ? 1u : 0u;
        acc[5] += (unsigned long long)sq;
        acc[6] += (unsigned long long)qp;
        acc[7] += (unsigned long long)sq + ((st == MPCX_QP_FAILURE) ? 0u : 1u);
    }
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        unsigned long long v = acc[k];
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) v += __shfl_down(v, off, 64);
        if ((threadIdx.x & 63) == 0 && v != 0) atomicAdd(&stats[k], v);
    }
}

// sin/cos as the solver computes them, for the arithmetic parity tests
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
                           double *workspace, int n_waves, int lanes, int coop_max, int lin_max, int coop_layout, int slots, hipStream_t stream)
{
    int boost_min = 16, boost_div = 1, boost_cap = 8192, boost_lanes = 4, boost_mul = 1;
    if (const char *e = getenv("MPCX_BOOST")) {      // "min,div,cap,lanes[,mul]"; div 0 switches the extra iterations off
        int a = 0, b = 0, c = 0, d = 1, m = 1;
        if (sscanf(e, "%d,%d,%d,%d,%d", &a, &b, &c, &d, &m) >= 3 && a >= 0 && b >= 0 && c >= 0 && d >= 0 && m >= 1) {
            boost_min = a; boost_div = b; boost_cap = c; boost_lanes = d; boost_mul = m;
        }
    }
    if (const char *e = getenv("MPCX_SLOTS")) { int v = atoi(e); if (v >= 0 && v <= slots) slots = v; }   // fewer only
    // interior-point iterations a problem gets in one go while its wavefront takes its few problems in turn,
    // the stages resident in registers (N < 64); 0 = the separate sweeps, one iteration at a time (diagnostic)
    int burst_few = 8;
    if (const char *e = getenv("MPCX_BURST")) { int v = atoi(e); if (v >= 0 && v <= 1024) burst_few = v; }

    // "alpha,extra": a problem inside a QP whose last step was shorter than alpha is left to the
    // one-lane-per-problem sweeps until it is `extra` iterations past the ageing threshold (alpha 0: off)
    double hold_alpha = 0.001;
    int hold_extra = 48;
    if (const char *e = getenv("MPCX_HOLD")) {
        double a = 0.0; int x = 0;
        if (sscanf(e, "%lf,%d", &a, &x) == 2 && a >= 0.0 && x >= 0) { hold_alpha = a; hold_extra = x; }
    }
    // 1: the Riccati recursion of a burst with one stage's algebra spread over the lanes; 0: as lane pipelines
    int spread_on = 1;
    if (const char *e = getenv("MPCX_SPREAD")) spread_on = atoi(e) != 0;
    // Hand-over of old problems to worker wavefronts (see the kernel): on when the grid is full wavefronts of 64
    // problems (large batches) with bursts available; 5/16 of the wavefronts are workers from the start, and a
    // wavefront keeps an old problem for itself while 64 descriptors wait (measured on 1 M problems: 128..512
    // workers, limits 16..2048; the reachable and obstacle workloads and 131 072 / 262 144 problems beside the
    // benchmark batch).  A problem a wavefront has kept is offered again from ho_late iterations on, and at the
    // end of the launch (see the kernel).  MPCX_HANDOVER="workers[,limit[,late]]" sets them, workers 0 switches
    // the hand-over off, late 0 the second offer.
    int n_workers = (n_waves >= 256) ? (n_waves * 5) / 16 : 0, ho_limit = 64, ho_late = 128;
    if (const char *e = getenv("MPCX_HANDOVER")) {      // (also on smaller grids: the tests)
        int v = 0, l = 64, a = 128;
        if (sscanf(e, "%d,%d,%d", &v, &l, &a) >= 1 && v >= 0 && v < n_waves && l >= 1 && a >= 0) { n_workers = v; ho_limit = l; ho_late = a; }
    }
    const bool handover = n_workers > 0 && lanes == 64 && !o.rti && in.N < 64 && burst_few > 0 &&
                          n_waves * slots <= MPCX_HANDOVER_SLOTS && coop_layout == 0 && coop_max > 0 && slots > 0;
    unsigned char *ho = handover ? reinterpret_cast<unsigned char *>(workspace) + 512 : nullptr;
    if (!handover) n_workers = 0;
    int svc_min = MPCX_SERVICE_MIN;
    if (const char *e = getenv("MPCX_SVC_MIN")) { int v = atoi(e); if (v >= 1 && v <= 64) svc_min = v; }
    if (coop_max < 0 || coop_max > 64) return hipErrorInvalidValue;
    if (lanes < 1 || lanes > 64 || n_waves < 1) return hipErrorInvalidValue;
    // workspace header: problem counter at byte 0, the options at byte 256, the hand-over area from byte 512
    static_assert(sizeof(mpcx_opts) <= 256, "options must fit the workspace header");
    int *counter = reinterpret_cast<int *>(workspace);
    mpcx_opts *d_opts = reinterpret_cast<mpcx_opts *>(reinterpret_cast<char *>(workspace) + 256);
    double *ws = workspace + MPCX_WS_HEADER_BYTES / 8;
    // (the options travel as a kernel argument: no copy from host memory, nothing to stage)
    hipLaunchKernelGGL(mpcx_header_kernel, dim3(1), dim3(1024), 0, stream, counter, d_opts, o, (unsigned)(n_waves - n_workers));
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    // fetch order inside the last (partial) block of 65536 problems: a multiplier coprime to its size
    // near the golden-section point; 0 = the caller's order throughout (MPCX_DEAL=0)
    unsigned deal = 1;
    const char *de = getenv("MPCX_DEAL");
    if (de && atoi(de) == 0) deal = 0;
    else {
        const unsigned long long rest = (unsigned long long)in.B & 65535u;
        if (rest > 2) {
            unsigned long long g = (unsigned long long)((double)rest * 0.6180339887498949);
            if (g < 1) g = 1;
            auto gcd = [](unsigned long long a, unsigned long long b) { while (b) { unsigned long long t = a % b; a = b; b = t; } return a; };
            while (gcd(g, rest) != 1) ++g;      // rest - 1 is coprime to rest, so this ends below rest
            deal = (unsigned)g;
        }
    }
    dim3 grid(n_waves), block(64);
    switch (nh) {
    case 0: hipLaunchKernelGGL(mpcx_sqp_kernel<0>, grid, block, 0, stream, in, d_opts, out, ws, counter, lanes, coop_max, lin_max, svc_min, boost_min, boost_div, boost_cap, boost_lanes, coop_layout, slots, deal, burst_few, hold_alpha, hold_extra, boost_mul, spread_on, ho, n_workers, ho_limit, ho_late); break;
#ifndef MPCX_DEV_ONLY_NH0   /* development builds compile one instantiation only */
    case 1: hipLaunchKernelGGL(mpcx_sqp_kernel<1>, grid, block, 0, stream, in, d_opts, out, ws, counter, lanes, coop_max, lin_max, svc_min, boost_min, boost_div, boost_cap, boost_lanes, coop_layout, slots, deal, burst_few, hold_alpha, hold_extra, boost_mul, spread_on, ho, n_workers, ho_limit, ho_late); break;
    case 2: hipLaunchKernelGGL(mpcx_sqp_kernel<2>, grid, block, 0, stream, in, d_opts, out, ws, counter, lanes, coop_max, lin_max, svc_min, boost_min, boost_div, boost_cap, boost_lanes, coop_layout, slots, deal, burst_few, hold_alpha, hold_extra, boost_mul, spread_on, ho, n_workers, ho_limit, ho_late); break;
    case 3: hipLaunchKernelGGL(mpcx_sqp_kernel<3>, grid, block, 0, stream, in, d_opts, out, ws, counter, lanes, coop_max, lin_max, svc_min, boost_min, boost_div, boost_cap, boost_lanes, coop_layout, slots, deal, burst_few, hold_alpha, hold_extra, boost_mul, spread_on, ho, n_workers, ho_limit, ho_late); break;
    default: hipLaunchKernelGGL(mpcx_sqp_kernel<4>, grid, block, 0, stream, in, d_opts, out, ws, counter, lanes, coop_max, lin_max, svc_min, boost_min, boost_div, boost_cap, boost_lanes, coop_layout, slots, deal, burst_few, hold_alpha, hold_extra, boost_mul, spread_on, ho, n_workers, ho_limit, ho_late); break;
#else
    default: return hipErrorInvalidValue;
#endif
    }
    return hipGetLastError();
}

hipError_t mpcx_launch_stats(const int32_t *status, const int32_t *sqp_iter, const int32_t *qp_iter, int B,
                             long long *stats, hipStream_t stream)
{
    hipError_t e = hipMemsetAsync(stats, 0, 8 * sizeof(long long), stream);
    if (e != hipSuccess) return e;
    int blocks = (B + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(mpcx_stats_kernel, dim3(blocks), dim3(256), 0, stream, status, sqp_iter, qp_iter, B,
                       reinterpret_cast<unsigned long long *>(stats));
    return hipGetLastError();
}

#ifdef MPCX_STAMPS
extern "C" int mpcx_debug_read_stamps(unsigned long long *out, int reset)
{
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(g_stamps), sizeof(g_stamps)) != hipSuccess) return -1;
    if (reset) {
        static unsigned long long zeros[2048][N_STAMP];
        if (hipMemcpyToSymbol(HIP_SYMBOL(g_stamps), zeros, sizeof(zeros)) != hipSuccess) return -1;
    }
    return 0;
}
#endif

hipError_t mpcx_launch_guess(const double *pdata, const int32_t *n_obs, const double *width, int B, int N, double *xguess,
                             hipStream_t stream)
{
    hipLaunchKernelGGL(mpcx_guess_kernel, dim3((B + 255) / 256), dim3(256), 0, stream, pdata, n_obs, width, B, N, xguess);
    return hipGetLastError();
}

hipError_t mpcx_launch_probe(const double *xin, int n, double *s, double *c, hipStream_t stream)
{
    hipLaunchKernelGGL(mpcx_probe_kernel, dim3((n + 255) / 256), dim3(256), 0, stream, xin, n, s, c);
    return hipGetLastError();
}
