/*
 * mpcx.h — C ABI of the batched kinematic-car OCP solver for MI355X (gfx950).
 *
 * The reference has no FFI of its own for this path: the seam is the Python
 * call  generate_controls(ego, sim_cfg)  (reference src/core_solver.py:112),
 * which builds ONE optimal-control problem and hands it to acados through
 * four methods (ctor :354, set :496/:499, solve :502, get :511-513).  The
 * entry points below replace exactly that inner seam, for a whole batch of
 * independent problems per call.  The host side that packs EgoConfig /
 * SimConfig objects into the tables declared here lives in
 * mpc_amd/core_solver.py and mirrors the reference's names and semantics.
 *
 * Conventions
 *  - plain C, no C++ types, no exceptions cross the boundary;
 *  - every pointer is caller-owned; the library keeps none after return;
 *  - return value 0 = the CALL succeeded (per-problem outcomes are in
 *    mpcx_result.status), negative = MPCX_E_* and mpcx_last_error() has text;
 *  - all floating point is IEEE binary64, all integers int32_t;
 *  - the library never falls back to a CPU solver: with no usable HIP device
 *    every compute entry point returns MPCX_E_NODEVICE.
 */
#ifndef MPCX_H
#define MPCX_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MPCX_VERSION 108

#define MPCX_NX 5       /* state  [x, y, theta, v, delta]        (core_solver.py:131) */
#define MPCX_NU 2       /* input  [acceleration, steering rate]  (core_solver.py:132) */
#define MPCX_MAX_OBS 4  /* circular obstacles per problem        (core_solver.py:332-351) */

/* Problem table: pdata[field * B + b], one column per problem (structure of
 * arrays, so that lane b of a wavefront reads consecutive addresses). */
enum {
    MPCX_F_X0   = 0,   /* 5  start state, heading already wrapped  (core_solver.py:312) */
    MPCX_F_GOAL = 5,   /* 5  goal state = yref[2:7] = yref_e       (core_solver.py:226-236) */
    MPCX_F_W    = 10,  /* 7  diag of W: a, steering rate, x, y, theta, v, delta (:183-193) */
    MPCX_F_WE   = 17,  /* 5  diag of W_e                           (core_solver.py:197-204) */
    MPCX_F_LBU  = 22,  /* 2  (core_solver.py:262) */
    MPCX_F_UBU  = 24,  /* 2  (core_solver.py:263) */
    MPCX_F_LBX  = 26,  /* 5  stages 1..N-1 (core_solver.py:307) */
    MPCX_F_UBX  = 31,  /* 5  (core_solver.py:308) */
    MPCX_F_LBXE = 36,  /* 5  terminal box  (core_solver.py:315-321) */
    MPCX_F_UBXE = 41,  /* 5  (core_solver.py:322-328) */
    MPCX_F_CG   = 46,  /* 4  corridor rows: C[0][0], C[0][1], C[1][0], C[1][1] (:280-286) */
    MPCX_F_LG   = 50,  /* 2  (core_solver.py:296) */
    MPCX_F_UG   = 52,  /* 2  (core_solver.py:297) */
    MPCX_F_L    = 54,  /* 1  wheelbase (core_solver.py:14,150) */
    MPCX_F_H    = 55,  /* 1  shooting interval = duration / N (core_solver.py:239) */
    MPCX_F_LH   = 56,  /* 1  lower bound of every obstacle row (core_solver.py:349) */
    MPCX_F_UH   = 57,  /* 1  upper bound of every obstacle row (core_solver.py:351) */
    MPCX_F_OBS  = 58,  /* 3*MPCX_MAX_OBS: ox, oy, (r+margin)^2 per obstacle (:338-340) */
    MPCX_F_CW   = 70,  /* 1  corridor_width: the clamp of the initial guess (core_solver.py:27,486-487);
                             read only when the library computes the guess (xguess == NULL) */
    MPCX_NFIELDS = 71
};

typedef struct mpcx_batch {
    int32_t B;              /* problems in this call                                  */
    int32_t N;              /* shooting intervals, same for the whole call (:133)     */
    const double  *pdata;   /* [MPCX_NFIELDS][B]                                      */
    const int32_t *n_obs;   /* [B], each 0..MPCX_MAX_OBS                              */
    const double  *xguess;  /* [N+1][5][B] initial state guess (core_solver.py:477-499), or NULL: the
                               library computes it on the device from the table (see
                               mpcx_initial_guess: identical to the host loop without obstacles;
                               with obstacles the reference's formulas :360-473 with products for
                               its pow(x, 2) and a unit vector for its cos / sin of arctan2)      */
    const double  *uguess;  /* [N][2][B] initial input guess, or NULL = zeros, which is what the
                               reference starts from (it never sets "u"); used by receding-
                               horizon callers that shift the previous solution              */
    const int32_t *order;   /* [B] a permutation of 0..B-1, or NULL: the caller's PRIORITY LIST.  The
                               GPU's wavefronts take the problems near its front first (the list is dealt
                               over the wavefronts, which take up to 64 problems at a time, so that
                               consecutive entries go to different wavefronts and do not wait for each
                               other).  Results are written at the problem's own index and do not depend
                               on it (bit for bit); the time of a large batch does, because a problem of
                               ten thousand iterations that is taken last ends long after everything else.
                               A caller who knows which problems are long (the qp_iter of an earlier
                               solve of the same or a similar batch: the last tick of a receding-horizon
                               loop, the last point of a sweep) puts them first.  NULL = the library's
                               own order (blocks of 65 536 in the caller's order, scattered inside a
                               block).  An entry outside 0..B-1 names no problem and is skipped; a
                               problem named twice is solved twice (same bits), one named nowhere keeps
                               whatever its result rows held.  mpcx_solve_batch and mpcx_ctx_solve check
                               the permutation on the host (MPCX_E_BADARG); the device entry points
                               cannot.  No counterpart in the reference, which solves one problem a
                               call.                                                            */
} mpcx_batch;

/* Per-problem status, acados numbering as seen by core_solver.py:502-507. */
enum {
    MPCX_SUCCESS    = 0,
    MPCX_NAN        = 1,  /* non-finite residual                         */
    MPCX_MAXITER    = 2,  /* max_sqp_iter QP solves without convergence  */
    MPCX_MINSTEP    = 3,  /* reserved (fixed full step never shrinks)    */
    MPCX_QP_FAILURE = 4   /* interior-point QP: minimum step or NaN      */
};

typedef struct mpcx_opts {
    int32_t max_sqp_iter;   /* nlp_solver_max_iter = 200 (core_solver.py:246)        */
    int32_t max_qp_iter;    /* qp_solver_iter_max  = 50  (core_solver.py:251)        */
    double  tol_stat, tol_eq, tol_ineq, tol_comp;   /* tol = 1e-3 (core_solver.py:252) */
    double  qp_tol_stat, qp_tol_eq, qp_tol_ineq, qp_tol_comp;   /* defaults 1e-6, 1e-8, 1e-8, 1e-6 */
    double  lm;             /* levenberg_marquardt = 1e-2 (core_solver.py:247)       */
    int32_t lm_scale_by_h;  /* 1 (default): add h*lm on stages < N, lm on stage N; 0: lm as is */
    int32_t cond_pred_corr; /* conditional Mehrotra corrector on/off                 */
    double  mu0;            /* initial barrier parameter                             */
    double  thr0;           /* slack floor of the cold start                         */
    double  alpha_min;      /* step below this = QP minimum-step failure             */
    double  lam_min, t_min; /* floors applied after every interior-point update      */
    int32_t rti;            /* 1: exactly one linearisation + one QP, no test        */
    int32_t max_obs;        /* largest n_obs[b] of the batch if known, -1 = unknown:
                               the host entry point then scans n_obs, the device
                               entry point assumes MPCX_MAX_OBS                      */
    /* Globalisation.  0 (default): every SQP step is taken in full, which is what the reference
     * runs (it sets no globalization, core_solver.py:242-253; acados' default is FIXED_STEP).
     * 1: backtracking line search on the exact penalty function
     *      phi(w) = f(w) + rho * max(|dynamics gaps|_inf, |violation of the inequality rows|_inf),
     * rho = max(rho so far, ls_rho_factor * sum |multipliers of the QP|): the step length starts
     * at 1 and is multiplied by ls_shrink until phi(w + a d) <= phi(w) - ls_c1 * a * rho *
     * (infeasibility at w) or a would fall below ls_alpha_min (that length is then taken); the
     * multipliers always take the full step.  No counterpart in the reference: parity unpinned.  */
    int32_t globalization;
    int32_t reserved_;      /* keeps the doubles below aligned; must be 0 */
    double  ls_shrink;      /* 0.5   */
    double  ls_alpha_min;   /* 1/64  */
    double  ls_c1;          /* 1e-4  */
    double  ls_rho_factor;  /* 2     */
} mpcx_opts;

typedef struct mpcx_result {
    double  *X;         /* [B][N+1][5] row b is the 'x' array of core_solver.py:521 */
    double  *U;         /* [B][N][2]   row b is the 'u' array of core_solver.py:522 */
    int32_t *status;    /* [B]   MPCX_SUCCESS ...                                   */
    int32_t *sqp_iter;  /* [B]   QP solves performed                                */
    int32_t *qp_iter;   /* [B]   interior-point iterations, summed over QP solves   */
    double  *cost;      /* [B]   objective at the returned iterate                  */
    double  *res;       /* [4][B] stat, eq, ineq, comp residual at the last test    */
} mpcx_result;

int         mpcx_version(void);
const char *mpcx_last_error(void);          /* thread-local, never NULL */
void        mpcx_opts_default(mpcx_opts *o);

/* Number of HIP devices this process can see (0 if none / no driver). */
int mpcx_device_count(void);

/*
 * Replaces AcadosOcpSolver(...) + set + solve + get (core_solver.py:354-513)
 * for B problems.  Pointers in `in` and `out` are HOST memory; the call
 * uploads, solves on HIP device `device`, downloads, and returns when the
 * results are in `out`.
 */
int mpcx_solve_batch(const mpcx_batch *in, const mpcx_opts *opts,
                     mpcx_result *out, int device);

/*
 * Same solve with every pointer in `in`, `out` and `workspace` already in the
 * memory of the CURRENT HIP device.  Asynchronous: kernels are queued on
 * `stream` (a hipStream_t, NULL = default stream) and the call returns.
 * `workspace` must hold mpcx_workspace_bytes(B, N) bytes, 256-byte aligned.
 * N is limited by the 32-bit byte offsets the kernels address a wavefront's workspace with:
 * every solve entry point returns MPCX_E_BADARG for N > mpcx_max_horizon() (about 54 000).
 */
int    mpcx_max_horizon(void);
size_t mpcx_workspace_bytes(int32_t B, int32_t N);
int    mpcx_solve_batch_device(const mpcx_batch *in, const mpcx_opts *opts,
                               mpcx_result *out, void *workspace,
                               size_t workspace_bytes, void *stream);

/*
 * The solve of mpcx_solve_batch_device for callers that cannot allocate device memory (code
 * under HIP graph capture, tight serving loops, harnesses that forbid hipMalloc).  It has no
 * counterpart in the reference.  All pointers in `in` and `out` are memory of the CURRENT HIP
 * device; the workspace is an array built into the library, MPCX_STATIC_SCRATCH_BYTES of device
 * memory on every GPU the library is loaded on; the call makes no hipMalloc / hipFree, no copy from
 * host memory and no other allocating call, and queues its kernels on `stream`.  The first call
 * on a device looks the array's address up (make it outside a graph capture).
 * Limits: the scratch must hold one wavefront's share, which bounds N: N > mpcx_static_max_horizon()
 * returns MPCX_E_BADARG; B is not limited, but only mpcx_static_max_wavefronts(N) wavefronts run (as
 * many as the scratch holds), so large batches are slow here.  All calls on one GPU share the array:
 * calls that overlap in time on the same GPU race and must not be made.  Calls queued on ONE stream run
 * one after the other and are fine; a call on another stream while the previous call's kernels have
 * not finished is detected (an event behind each call, queried on entry) and returns MPCX_E_BADARG
 * without launching.  Calls made under stream capture are not checked: whoever launches the captured
 * graphs keeps them apart.
 */
#define MPCX_STATIC_SCRATCH_BYTES (64u << 20)
int mpcx_solve_batch_static(const mpcx_batch *in, const mpcx_opts *opts, mpcx_result *out, void *stream);
int mpcx_static_max_horizon(void);              /* largest N the built-in scratch holds           */
int mpcx_static_max_wavefronts(int32_t N);      /* wavefronts it holds at horizon N (0: too long) */

/*
 * Initial state guess of core_solver.py:477-499, computed on the device from the problem table:
 * node i gets the start-goal interpolation at t = i/N, y clamped to the reference's centre-line
 * formula +- width/2, heading interpolated, v = 1 - t, delta = 0; node N gets the goal state.
 * Without obstacles (n_obs == NULL, or n_obs[b] == 0) these are the same operations in the same
 * order as the host loop and the result is bit-identical to it.  With obstacles every point is then
 * moved off them as adjust_point_for_obstacles does (:360-473: the start-goal line shifted sideways
 * past an obstacle that straddles it and the point projected on it, else a point inside an obstacle
 * pushed out radially to radius + margin + 0.05), with two differences in the last bits: squares
 * are products where the reference calls pow(x, 2), and the radial direction is (dx, dy) / distance
 * where the reference takes cos and sin of arctan2(dy, dx) from the host libm.
 * width[b] = EgoConfig.corridor_width (core_solver.py:27), or width == NULL: the table's own
 * MPCX_F_CW column.
 * _device: pdata [MPCX_NFIELDS][B], n_obs [B], width [B], xguess [N+1][5][B] are device pointers and
 * the kernel is queued on `stream`; the other form takes host pointers and device `device`.
 */
int mpcx_initial_guess_device(const double *pdata, const int32_t *n_obs, const double *width, int32_t B, int32_t N,
                              double *xguess, void *stream);
int mpcx_initial_guess(const double *pdata, const int32_t *n_obs, const double *width, int32_t B, int32_t N,
                       double *xguess, int device);

enum { MPCX_STAT_SQP = 5, MPCX_STAT_QP = 6, MPCX_STAT_LIN = 7, MPCX_NSTATS = 8 };

/*
 * Solver handle.  Plays the part of the AcadosOcpSolver object of core_solver.py:354 (created,
 * solved with, dropped), except that it is not tied to one problem: it owns what a solve needs on
 * HIP device `device` -- a stream, the device copies of the tables, the workspace, and page-locked
 * host arrays for the results -- and keeps them from call to call, growing them when a larger
 * batch comes, so that a caller who solves batch after batch pays for allocation once.
 *
 * mpcx_ctx_solve: `in` holds HOST pointers (read during the call only).  On return `out` POINTS
 * INTO the handle's page-locked result arrays (same layout as mpcx_result everywhere else); they
 * stay valid until the next mpcx_ctx_solve or mpcx_ctx_destroy on this handle, and the caller
 * never frees them.  stats (may be NULL) receives the MPCX_NSTATS outcome counters of
 * mpcx_batch_stats_device.  One call at a time per handle; different handles are independent.
 */
typedef struct mpcx_ctx mpcx_ctx;
int  mpcx_ctx_create(mpcx_ctx **ctx, int device);
int  mpcx_ctx_solve(mpcx_ctx *ctx, const mpcx_batch *in, const mpcx_opts *opts, mpcx_result *out,
                    int64_t *stats);
void mpcx_ctx_destroy(mpcx_ctx *ctx);

/*
 * Outcome counters of a solved batch, reduced on the device so that a caller who only needs
 * the statistics does not download the per-problem arrays.  Replaces the host-side counting of
 * the reference's loops (grid_path_plan.py:80-92 appends and counts the successes, scenarios.py:
 * 102-105 stops at the first failure): stats[s] for s = 0..4 = problems that ended with status s,
 * stats[MPCX_STAT_SQP] = QP solves, stats[MPCX_STAT_QP] = interior-point iterations,
 * stats[MPCX_STAT_LIN] = linearisations (one per QP solve that did not fail, plus the first).
 * All pointers are memory of the current HIP device, `stats` holds MPCX_NSTATS int64; the work is
 * queued on `stream` behind the solve that produced the arrays.
 */
int mpcx_batch_stats_device(const int32_t *status, const int32_t *sqp_iter, const int32_t *qp_iter,
                            int32_t B, int64_t *stats, void *stream);

/*
 * Diagnostic, no counterpart in the reference: sin and cos of n host doubles,
 * evaluated on device `device` by the same routine the solver's RK4 uses, so
 * that tests can compare the device arithmetic bit for bit with a CPU program.
 */
int mpcx_probe_sincos(const double *x, int32_t n, double *s, double *c, int device);

enum {
    MPCX_E_BADARG   = -1,
    MPCX_E_NODEVICE = -2,
    MPCX_E_HIP      = -3,
    MPCX_E_NOMEM    = -4
};

#ifdef __cplusplus
}
#endif
#endif /* MPCX_H */
