// mpcx_api.hip — the C ABI of include/mpcx.h on top of the kernels.
//
// Host code only: argument checks, device buffers, the launch.  There is no
// CPU solver behind these entry points; without a usable HIP device they
// return MPCX_E_NODEVICE.
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <atomic>
#include <cstring>
#include <mutex>
#include <memory>
#include <new>

#include "../../include/mpcx.h"
#include "mpcx_kernels.h"

// Workspace of mpcx_solve_batch_static: one copy per GPU the library is loaded on, held until the
// process ends.  Host code never passes this name to a kernel; it passes the address that
// hipGetSymbolAddress returns for the current device.
static __device__ __align__(256) unsigned char scratch[MPCX_STATIC_SCRATCH_BYTES];

namespace {

constexpr int kMaxDevices = 64;
std::atomic<void *> g_scratch_ptr[kMaxDevices];
// Overlap guard of mpcx_solve_batch_static, per device: an event recorded behind the last call's
// kernels and the stream they were queued on.  A call on ANOTHER stream while that event has not
// completed would share the scratch with a solve in flight and is refused; calls queued on the same
// stream run one after the other and are fine.  (Calls made under stream capture are neither checked
// nor recorded: a captured solve runs when its graph is launched, not when it is captured.)
struct StaticGuard { std::mutex mu; hipEvent_t done = nullptr; hipStream_t last = nullptr; bool have = false; };
StaticGuard g_guard[kMaxDevices];

thread_local char g_err[512] = "";

int fail(int code, const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

#define HIP_TRY(expr)                                                                 \
    do {                                                                              \
        hipError_t e_ = (expr);                                                       \
        if (e_ != hipSuccess)                                                         \
            return fail(MPCX_E_HIP, "%s: %s", #expr, hipGetErrorString(e_));          \
    } while (0)

constexpr int kWavesPerCU = 4;    // resident wavefronts per CU the kernel's register use admits

// How a batch is spread over persistent wavefronts.  Large batches fill every wavefront
// (64 problems each, refilled from the queue).  Small ones are spread thin, down to one
// problem per wavefront, because a wavefront with few problems finishes each of them
// sooner; the spread is limited by the workspace it needs (`target` bytes; the default is sized
// for a GPU with 288 GB, and the host entry point falls back to less if the allocation fails).
struct Plan { int waves, lanes, coop_max, lin_max, coop_layout; };
constexpr int kLinMax = 0;
constexpr size_t kHeader = MPCX_WS_HEADER_BYTES;      // problem counter, options, hand-over area
constexpr int kCoopMax = 8;     // problems per wavefront up to which the single-problem sweeps pay
constexpr int kCoopMaxRti = 16; // ... in a real-time iteration (a handful of iterations per problem, all in bursts; measured 8..32)
constexpr int kSlots = 8;        // regions per wavefront for problems worked on in single-problem mode only
constexpr int kPlanWaves = 1024;   // one wavefront per SIMD of 256 CUs; more would only wait for a slot
// Up to this many problems a full solve is fastest with ONE problem at a time per wavefront, each wavefront
// taking the next from the queue when its problem ends (single-problem mode throughout, no problem waits for
// a neighbour's bursts; measured against 2..64 per wavefront: 16 384 problems 81 -> 45 ms, 65 536 121 -> 118 ms,
// 131 072 256 -> 304 ms)
constexpr int kOneAtATimeMax = 65536;

size_t default_workspace_target()
{
    size_t gib = 64;
    if (const char *e = std::getenv("MPCX_WORKSPACE_GIB")) {
        const long v = std::atol(e);
        if (v >= 1 && v <= 256) gib = (size_t)v;
    }
    return gib << 30;
}

// a wavefront's workspace: N+1 stage records for 64 problems, then kSlots regions of
// ceil((N+1)/64) records each, one per problem, for the last problems of the wavefront
size_t wave_bytes(int N, int nh)
{
    return (size_t)((N + 1) + kSlots * ((N + 64) / 64)) * mpcx_stage_doubles(nh) * 64 * sizeof(double);
}

// largest N whose wavefront workspace stays below the 4 GiB that 32-bit byte offsets reach
int mpcx_max_horizon_impl()
{
    int lo = 1, hi = 1 << 20;
    while (lo < hi) {
        const int mid = lo + (hi - lo + 1) / 2;
        if (wave_bytes(mid, MPCX_MAX_OBS) < ((size_t)1 << 32)) lo = mid; else hi = mid - 1;
    }
    return lo;
}

Plan plan_for(int B, int N, size_t target, bool rti = false)
{
    const size_t per_wave = wave_bytes(N, MPCX_MAX_OBS);
    size_t cap = target / per_wave;
    if (cap < 1) cap = 1;
    if (cap > (size_t)kPlanWaves) cap = kPlanWaves;
    Plan p;
    p.lanes = (int)(((size_t)B + cap - 1) / cap);
    // One linearisation and one QP per problem (real-time iteration): all problems take about the
    // same few iterations, so once there are too many per wavefront for single-problem mode the
    // one-lane-per-problem sweeps are best run full (a sweep costs the same with 16 lanes as with 64).
    const int coop_default = rti ? kCoopMaxRti : kCoopMax;
    if (rti && p.lanes > coop_default) p.lanes = 64;
    // (when the workspace holds a wavefront per SIMD, or one per problem; a small workspace -- the static one, a
    // caller's own -- is better used by full wavefronts)
    if (!rti && B <= kOneAtATimeMax && (cap >= (size_t)kPlanWaves || cap >= (size_t)B)) p.lanes = 1;
    // diagnostic knob: MPCX_LANES=n fixes the problems per wavefront (tests use it to run
    // small batches through the one-lane-per-problem sweeps)
    if (const char *e = std::getenv("MPCX_LANES")) {
        const int v = std::atoi(e);
        if (v >= 1 && v <= 64) p.lanes = v;
    }
    if (p.lanes < 1) p.lanes = 1;
    if (p.lanes > 64) p.lanes = 64;
    p.coop_max = coop_default;
    if (const char *e = std::getenv("MPCX_COOP_MAX")) {      // diagnostic knob, 0 switches the mode off
        const int v = std::atoi(e);
        if (v >= 0 && v <= 64) p.coop_max = v;
    }
    p.lin_max = kLinMax;
    if (const char *e = std::getenv("MPCX_LIN_MAX")) {
        const int v = std::atoi(e);
        if (v >= 0 && v <= 64) p.lin_max = v;
    }
    // few enough problems per wavefront to stay in single-problem mode throughout: per-problem layout
    p.coop_layout = (p.coop_max > 0 && p.lanes <= p.coop_max && p.lanes * ((N + 64) / 64) <= N + 1) ? 1 : 0;
    if (const char *e = std::getenv("MPCX_COOP_LAYOUT")) {
        if (std::atoi(e) == 0) p.coop_layout = 0;
    }
    p.waves = (B + p.lanes - 1) / p.lanes;
    if (p.waves > (int)cap) p.waves = (int)cap;
    if (p.waves < 1) p.waves = 1;
    return p;
}

// persistent wavefronts actually launched on the current device
int waves_to_launch(const Plan &p)
{
    int cus = 256, dev = 0;
    if (hipGetDevice(&dev) == hipSuccess) {
        int v = 0;
        if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0) cus = v;
    }
    int w = cus * kWavesPerCU;
    return p.waves < w ? p.waves : w;
}

int check_call(const mpcx_batch *in, const mpcx_opts *o, const mpcx_result *out)
{
    if (!in || !o || !out) return fail(MPCX_E_BADARG, "null argument");
    if (in->B < 0 || in->N < 1) return fail(MPCX_E_BADARG, "need B >= 0 and N >= 1 (B=%d, N=%d)", in->B, in->N);
    if (in->B > 0 && !in->pdata) return fail(MPCX_E_BADARG, "null pdata");
    // the kernels address a wavefront's workspace with 32-bit byte offsets
    if (wave_bytes(in->N, MPCX_MAX_OBS) >= ((size_t)1 << 32))
        return fail(MPCX_E_BADARG, "N = %d too large: a wavefront's workspace would reach 4 GiB (N <= %d)", in->N, mpcx_max_horizon());
    if (in->B > 0 && (!out->X || !out->U || !out->status || !out->sqp_iter || !out->qp_iter ||
                      !out->cost || !out->res))
        return fail(MPCX_E_BADARG, "null output array");
    if (o->max_sqp_iter < 0 || o->max_qp_iter < 0) return fail(MPCX_E_BADARG, "negative iteration limit");
    if (o->max_obs > MPCX_MAX_OBS) return fail(MPCX_E_BADARG, "max_obs %d > %d", o->max_obs, MPCX_MAX_OBS);
    return 0;
}

// a caller's fetch order in HOST memory: a permutation of 0..B-1
int check_order(const int32_t *order, size_t B)
{
    // (no exception may cross the C boundary)
    std::unique_ptr<unsigned char[]> seen(new (std::nothrow) unsigned char[B ? B : 1]());
    if (!seen) return fail(MPCX_E_NOMEM, "no memory to check the order (%zu bytes)", B);
    for (size_t c = 0; c < B; ++c) {
        const int32_t b = order[c];
        if (b < 0 || (size_t)b >= B) return fail(MPCX_E_BADARG, "order[%zu] = %d outside [0, %zu)", c, b, B);
        if (seen[(size_t)b]) return fail(MPCX_E_BADARG, "order names problem %d twice (second time at %zu)", b, c);
        seen[(size_t)b] = 1;
    }
    return 0;
}

struct DevBuf {
    void *p = nullptr;
    ~DevBuf() { if (p) (void)hipFree(p); }
    hipError_t alloc(size_t bytes) { return hipMalloc(&p, bytes ? bytes : 1); }
    template <class T> T *as() const { return static_cast<T *>(p); }
};


// a buffer that grows on demand and is kept between calls: device memory, or page-locked host memory
struct Grow {
    void *p = nullptr;
    size_t cap = 0;
    bool host = false;
    void release()
    {
        if (p) (void)(host ? hipHostFree(p) : hipFree(p));
        p = nullptr;
        cap = 0;
    }
    hipError_t ensure(size_t bytes)
    {
        if (bytes <= cap) return hipSuccess;
        release();
        const hipError_t e = host ? hipHostMalloc(&p, bytes, hipHostMallocDefault) : hipMalloc(&p, bytes);
        if (e != hipSuccess) { p = nullptr; return e; }
        cap = bytes;
        return hipSuccess;
    }
    template <class T> T *as() const { return static_cast<T *>(p); }
};

}  // namespace

struct mpcx_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    Grow pd, no, xg, ug, ord, X, U, st, si, qi, co, re, ws, stats;      // device
    Grow hX, hU, hst, hsi, hqi, hco, hre, hstats;                  // page-locked host
    mpcx_ctx() { hX.host = hU.host = hst.host = hsi.host = hqi.host = hco.host = hre.host = hstats.host = true; }
};

namespace {
}  // namespace

extern "C" {

int mpcx_version(void) { return MPCX_VERSION; }

const char *mpcx_last_error(void) { return g_err; }

void mpcx_opts_default(mpcx_opts *o)
{
    if (!o) return;
    std::memset(o, 0, sizeof(*o));
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
    o->globalization = 0;      /* full steps, as the reference */
    o->ls_shrink = 0.5;
    o->ls_alpha_min = 1.0 / 64.0;
    o->ls_c1 = 1e-4;
    o->ls_rho_factor = 2.0;
}

int mpcx_max_horizon(void)
{
    static const int n = mpcx_max_horizon_impl();
    return n;
}

int mpcx_static_max_horizon(void)
{
    static const int n = [] {
        int lo = 0, hi = 1 << 20;
        while (lo < hi) {
            const int mid = lo + (hi - lo + 1) / 2;
            if (kHeader + wave_bytes(mid, MPCX_MAX_OBS) <= (size_t)MPCX_STATIC_SCRATCH_BYTES) lo = mid; else hi = mid - 1;
        }
        return lo;
    }();
    return n;
}

int mpcx_static_max_wavefronts(int32_t N)
{
    if (N < 1 || N > mpcx_static_max_horizon()) return 0;
    const size_t w = ((size_t)MPCX_STATIC_SCRATCH_BYTES - kHeader) / wave_bytes(N, MPCX_MAX_OBS);
    return (int)(w < (size_t)kPlanWaves ? w : (size_t)kPlanWaves);
}

int mpcx_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

size_t mpcx_workspace_bytes(int32_t B, int32_t N)
{
    if (B <= 0 || N < 1) return 0;
    return kHeader + (size_t)plan_for(B, N, default_workspace_target()).waves * wave_bytes(N, MPCX_MAX_OBS);
}

int mpcx_solve_batch_device(const mpcx_batch *in, const mpcx_opts *opts, mpcx_result *out,
                            void *workspace, size_t workspace_bytes, void *stream)
{
    if (int rc = check_call(in, opts, out)) return rc;
    if (in->B == 0) return 0;
    if (mpcx_device_count() <= 0) return fail(MPCX_E_NODEVICE, "no HIP device");
    const int nh = (opts->max_obs < 0) ? MPCX_MAX_OBS : opts->max_obs;
    // the batch is spread as far as the given workspace allows (mpcx_workspace_bytes() is the
    // size that allows the widest useful spread); one wavefront's worth is the minimum
    const size_t least = kHeader + wave_bytes(in->N, MPCX_MAX_OBS);
    if (!workspace || workspace_bytes < least)
        return fail(MPCX_E_BADARG, "workspace too small: %zu bytes given, at least %zu needed", workspace_bytes, least);
    const Plan plan = plan_for(in->B, in->N, workspace_bytes - kHeader, opts->rti != 0);
    if ((uintptr_t)workspace % 256 != 0) return fail(MPCX_E_BADARG, "workspace not 256-byte aligned");
    HIP_TRY(mpcx_launch_sqp(nh, *in, *opts, *out, static_cast<double *>(workspace), waves_to_launch(plan), plan.lanes, plan.coop_max, plan.lin_max, plan.coop_layout, kSlots,
                            static_cast<hipStream_t>(stream)));
    return 0;
}

int mpcx_solve_batch(const mpcx_batch *in, const mpcx_opts *opts, mpcx_result *out, int device)
{
    if (int rc = check_call(in, opts, out)) return rc;
    if (in->B == 0) return 0;
    const int ndev = mpcx_device_count();
    if (ndev <= 0) return fail(MPCX_E_NODEVICE, "no HIP device");
    if (device < 0 || device >= ndev) return fail(MPCX_E_BADARG, "device %d out of range [0, %d)", device, ndev);
    HIP_TRY(hipSetDevice(device));

    const size_t B = (size_t)in->B, N = (size_t)in->N;
    mpcx_opts o = *opts;
    int nh = 0;
    if (in->n_obs) {
        for (size_t b = 0; b < B; ++b) {
            if (in->n_obs[b] < 0 || in->n_obs[b] > MPCX_MAX_OBS)
                return fail(MPCX_E_BADARG, "n_obs[%zu] = %d outside [0, %d]", b, in->n_obs[b], MPCX_MAX_OBS);
            if (in->n_obs[b] > nh) nh = in->n_obs[b];
        }
    }
    if (o.max_obs >= 0 && o.max_obs < nh) return fail(MPCX_E_BADARG, "max_obs %d < largest n_obs %d", o.max_obs, nh);
    o.max_obs = nh;
    if (in->order) if (int rc = check_order(in->order, B)) return rc;

    const size_t n_pd = (size_t)MPCX_NFIELDS * B, n_xg = in->xguess ? (N + 1) * MPCX_NX * B : 0;
    const size_t n_X = B * (N + 1) * MPCX_NX, n_U = B * N * MPCX_NU;
    // the widest spread whose workspace can be allocated
    DevBuf d_ws;
    Plan plan;
    size_t ws_bytes = 0;
    for (size_t target = default_workspace_target();; target /= 2) {
        plan = plan_for(in->B, in->N, target, o.rti != 0);
        ws_bytes = kHeader + (size_t)plan.waves * wave_bytes(in->N, nh);
        if (d_ws.alloc(ws_bytes) == hipSuccess) break;
        (void)hipGetLastError();
        d_ws.p = nullptr;
        if (plan.waves == 1 || target < ((size_t)1 << 28))
            return fail(MPCX_E_NOMEM, "device allocation failed (workspace %zu bytes)", ws_bytes);
    }

    DevBuf d_pd, d_no, d_xg, d_ug, d_X, d_U, d_st, d_si, d_qi, d_co, d_re;
    if (d_pd.alloc(n_pd * 8) != hipSuccess || d_no.alloc(B * 4) != hipSuccess ||
        d_xg.alloc(n_xg * 8) != hipSuccess || d_X.alloc(n_X * 8) != hipSuccess ||
        d_U.alloc(n_U * 8) != hipSuccess || d_st.alloc(B * 4) != hipSuccess ||
        d_si.alloc(B * 4) != hipSuccess || d_qi.alloc(B * 4) != hipSuccess ||
        d_co.alloc(B * 8) != hipSuccess || d_re.alloc(4 * B * 8) != hipSuccess) {
        (void)hipGetLastError();
        return fail(MPCX_E_NOMEM, "device allocation failed (workspace %zu bytes)", ws_bytes);
    }
    HIP_TRY(hipMemcpy(d_pd.p, in->pdata, n_pd * 8, hipMemcpyHostToDevice));
    if (in->n_obs) HIP_TRY(hipMemcpy(d_no.p, in->n_obs, B * 4, hipMemcpyHostToDevice));
    else HIP_TRY(hipMemset(d_no.p, 0, B * 4));
    if (in->xguess) HIP_TRY(hipMemcpy(d_xg.p, in->xguess, n_xg * 8, hipMemcpyHostToDevice));
    if (in->uguess) {
        if (d_ug.alloc(n_U * 8) != hipSuccess) { (void)hipGetLastError(); return fail(MPCX_E_NOMEM, "device allocation failed"); }
        HIP_TRY(hipMemcpy(d_ug.p, in->uguess, n_U * 8, hipMemcpyHostToDevice));
    }
    DevBuf d_or;
    if (in->order) {
        if (d_or.alloc(B * 4) != hipSuccess) { (void)hipGetLastError(); return fail(MPCX_E_NOMEM, "device allocation failed"); }
        HIP_TRY(hipMemcpy(d_or.p, in->order, B * 4, hipMemcpyHostToDevice));
    }
    HIP_TRY(hipMemset(d_st.p, 0xff, B * 4));

    mpcx_batch din = *in;
    din.pdata = d_pd.as<double>();
    din.n_obs = d_no.as<int32_t>();
    din.xguess = in->xguess ? d_xg.as<double>() : nullptr;
    din.uguess = in->uguess ? d_ug.as<double>() : nullptr;
    din.order = in->order ? d_or.as<int32_t>() : nullptr;
    mpcx_result dout;
    dout.X = d_X.as<double>();
    dout.U = d_U.as<double>();
    dout.status = d_st.as<int32_t>();
    dout.sqp_iter = d_si.as<int32_t>();
    dout.qp_iter = d_qi.as<int32_t>();
    dout.cost = d_co.as<double>();
    dout.res = d_re.as<double>();

    HIP_TRY(mpcx_launch_sqp(nh, din, o, dout, d_ws.as<double>(), waves_to_launch(plan), plan.lanes, plan.coop_max, plan.lin_max, plan.coop_layout, kSlots, nullptr));
    HIP_TRY(hipDeviceSynchronize());

    HIP_TRY(hipMemcpy(out->X, d_X.p, n_X * 8, hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(out->U, d_U.p, n_U * 8, hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(out->status, d_st.p, B * 4, hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(out->sqp_iter, d_si.p, B * 4, hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(out->qp_iter, d_qi.p, B * 4, hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(out->cost, d_co.p, B * 8, hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(out->res, d_re.p, 4 * B * 8, hipMemcpyDeviceToHost));
    return 0;
}

int mpcx_initial_guess_device(const double *pdata, const int32_t *n_obs, const double *width, int32_t B, int32_t N,
                              double *xguess, void *stream)
{
    if (B < 0 || N < 1) return fail(MPCX_E_BADARG, "need B >= 0 and N >= 1 (B=%d, N=%d)", B, N);
    if (B == 0) return 0;
    if (!pdata || !xguess) return fail(MPCX_E_BADARG, "null argument");
    if (mpcx_device_count() <= 0) return fail(MPCX_E_NODEVICE, "no HIP device");
    HIP_TRY(mpcx_launch_guess(pdata, n_obs, width, B, N, xguess, static_cast<hipStream_t>(stream)));
    return 0;
}

int mpcx_initial_guess(const double *pdata, const int32_t *n_obs, const double *width, int32_t B, int32_t N,
                       double *xguess, int device)
{
    if (B < 0 || N < 1) return fail(MPCX_E_BADARG, "need B >= 0 and N >= 1 (B=%d, N=%d)", B, N);
    if (B == 0) return 0;
    if (!pdata || !xguess) return fail(MPCX_E_BADARG, "null argument");
    const int ndev = mpcx_device_count();
    if (ndev <= 0) return fail(MPCX_E_NODEVICE, "no HIP device");
    if (device < 0 || device >= ndev) return fail(MPCX_E_BADARG, "device %d out of range [0, %d)", device, ndev);
    HIP_TRY(hipSetDevice(device));
    const size_t n_pd = (size_t)MPCX_NFIELDS * B, n_xg = (size_t)(N + 1) * MPCX_NX * B;
    if (n_obs) {
        for (int32_t b = 0; b < B; ++b)
            if (n_obs[b] < 0 || n_obs[b] > MPCX_MAX_OBS)
                return fail(MPCX_E_BADARG, "n_obs[%d] = %d outside [0, %d]", b, n_obs[b], MPCX_MAX_OBS);
    }
    DevBuf d_pd, d_w, d_no, d_xg;
    if (d_pd.alloc(n_pd * 8) != hipSuccess || d_w.alloc((size_t)B * 8) != hipSuccess ||
        d_no.alloc((size_t)B * 4) != hipSuccess || d_xg.alloc(n_xg * 8) != hipSuccess) {
        (void)hipGetLastError();
        return fail(MPCX_E_NOMEM, "device allocation failed");
    }
    HIP_TRY(hipMemcpy(d_pd.p, pdata, n_pd * 8, hipMemcpyHostToDevice));
    if (width) HIP_TRY(hipMemcpy(d_w.p, width, (size_t)B * 8, hipMemcpyHostToDevice));
    if (n_obs) HIP_TRY(hipMemcpy(d_no.p, n_obs, (size_t)B * 4, hipMemcpyHostToDevice));
    HIP_TRY(mpcx_launch_guess(d_pd.as<double>(), n_obs ? d_no.as<int32_t>() : nullptr, width ? d_w.as<double>() : nullptr,
                              B, N, d_xg.as<double>(), nullptr));
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpy(xguess, d_xg.p, n_xg * 8, hipMemcpyDeviceToHost));
    return 0;
}

int mpcx_batch_stats_device(const int32_t *status, const int32_t *sqp_iter, const int32_t *qp_iter, int32_t B,
                            int64_t *stats, void *stream)
{
    if (B < 0) return fail(MPCX_E_BADARG, "need B >= 0 (B=%d)", B);
    if (!stats || (B > 0 && (!status || !sqp_iter || !qp_iter))) return fail(MPCX_E_BADARG, "null argument");
    if (mpcx_device_count() <= 0) return fail(MPCX_E_NODEVICE, "no HIP device");
    static_assert(sizeof(long long) == sizeof(int64_t), "int64_t is long long on this ABI");
    if (B == 0) {
        HIP_TRY(hipMemsetAsync(stats, 0, MPCX_NSTATS * sizeof(int64_t), static_cast<hipStream_t>(stream)));
        return 0;
    }
    HIP_TRY(mpcx_launch_stats(status, sqp_iter, qp_iter, B, reinterpret_cast<long long *>(stats),
                              static_cast<hipStream_t>(stream)));
    return 0;
}

int mpcx_ctx_create(mpcx_ctx **ctx, int device)
{
    if (!ctx) return fail(MPCX_E_BADARG, "null argument");
    *ctx = nullptr;
    const int ndev = mpcx_device_count();
    if (ndev <= 0) return fail(MPCX_E_NODEVICE, "no HIP device");
    if (device < 0 || device >= ndev) return fail(MPCX_E_BADARG, "device %d out of range [0, %d)", device, ndev);
    HIP_TRY(hipSetDevice(device));
    mpcx_ctx *c = new (std::nothrow) mpcx_ctx;
    if (!c) return fail(MPCX_E_NOMEM, "out of host memory");
    c->device = device;
    const hipError_t e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking);
    if (e != hipSuccess) {
        delete c;
        return fail(MPCX_E_HIP, "hipStreamCreate: %s", hipGetErrorString(e));
    }
    *ctx = c;
    return 0;
}

void mpcx_ctx_destroy(mpcx_ctx *c)
{
    if (!c) return;
    (void)hipSetDevice(c->device);
    if (c->stream) { (void)hipStreamSynchronize(c->stream); (void)hipStreamDestroy(c->stream); }
    Grow *all[] = {&c->pd, &c->no, &c->xg, &c->ug, &c->ord, &c->X, &c->U, &c->st, &c->si, &c->qi, &c->co, &c->re, &c->ws,
                   &c->stats, &c->hX, &c->hU, &c->hst, &c->hsi, &c->hqi, &c->hco, &c->hre, &c->hstats};
    for (Grow *g : all) g->release();
    delete c;
}

int mpcx_ctx_solve(mpcx_ctx *c, const mpcx_batch *in, const mpcx_opts *opts, mpcx_result *out, int64_t *stats)
{
    if (!c || !in || !opts || !out) return fail(MPCX_E_BADARG, "null argument");
    if (in->B < 0 || in->N < 1) return fail(MPCX_E_BADARG, "need B >= 0 and N >= 1 (B=%d, N=%d)", in->B, in->N);
    if (in->B > 0 && !in->pdata) return fail(MPCX_E_BADARG, "null pdata");
    if (wave_bytes(in->N, MPCX_MAX_OBS) >= ((size_t)1 << 32))
        return fail(MPCX_E_BADARG, "N = %d too large: a wavefront's workspace would reach 4 GiB (N <= %d)", in->N, mpcx_max_horizon());
    if (opts->max_sqp_iter < 0 || opts->max_qp_iter < 0) return fail(MPCX_E_BADARG, "negative iteration limit");
    if (opts->max_obs > MPCX_MAX_OBS) return fail(MPCX_E_BADARG, "max_obs %d > %d", opts->max_obs, MPCX_MAX_OBS);
    HIP_TRY(hipSetDevice(c->device));
    const size_t B = (size_t)in->B, N = (size_t)in->N;
    mpcx_opts o = *opts;
    int nh = 0;
    if (in->n_obs) {
        for (size_t b = 0; b < B; ++b) {
            if (in->n_obs[b] < 0 || in->n_obs[b] > MPCX_MAX_OBS)
                return fail(MPCX_E_BADARG, "n_obs[%zu] = %d outside [0, %d]", b, in->n_obs[b], MPCX_MAX_OBS);
            if (in->n_obs[b] > nh) nh = in->n_obs[b];
        }
    }
    if (o.max_obs >= 0 && o.max_obs < nh) return fail(MPCX_E_BADARG, "max_obs %d < largest n_obs %d", o.max_obs, nh);
    o.max_obs = nh;
    if (in->order) if (int rc = check_order(in->order, B)) return rc;

    const char *stage_env = std::getenv("MPCX_CTX_STAGE");
    const bool staged = stage_env && std::atoi(stage_env) != 0;
    const size_t n_pd = (size_t)MPCX_NFIELDS * B, n_xg = in->xguess ? (N + 1) * MPCX_NX * B : 0;
    const size_t n_X = B * (N + 1) * MPCX_NX, n_U = B * N * MPCX_NU;
    // buffers: kept from the last call where large enough (at least one byte each, so B = 0 yields valid pointers)
    struct { Grow *g; size_t bytes; } need[] = {
        {&c->pd, n_pd * 8}, {&c->no, B * 4}, {&c->xg, n_xg * 8}, {&c->ug, in->uguess ? n_U * 8 : 0}, {&c->ord, in->order ? B * 4 : 0},
        {&c->X, staged ? n_X * 8 : 0}, {&c->U, staged ? n_U * 8 : 0}, {&c->st, B * 4}, {&c->si, B * 4}, {&c->qi, B * 4}, {&c->co, B * 8},
        {&c->re, 4 * B * 8}, {&c->stats, MPCX_NSTATS * 8},
        {&c->hX, n_X * 8}, {&c->hU, n_U * 8}, {&c->hst, B * 4}, {&c->hsi, B * 4}, {&c->hqi, B * 4}, {&c->hco, B * 8},
        {&c->hre, 4 * B * 8}, {&c->hstats, MPCX_NSTATS * 8}};
    for (auto &n : need) {
        if (n.g->ensure(n.bytes ? n.bytes : 1) != hipSuccess) {
            (void)hipGetLastError();
            return fail(MPCX_E_NOMEM, "allocation of %zu bytes failed", n.bytes);
        }
    }
    out->X = c->hX.as<double>();
    out->U = c->hU.as<double>();
    out->status = c->hst.as<int32_t>();
    out->sqp_iter = c->hsi.as<int32_t>();
    out->qp_iter = c->hqi.as<int32_t>();
    out->cost = c->hco.as<double>();
    out->res = c->hre.as<double>();
    if (B == 0) {
        if (stats) std::memset(stats, 0, MPCX_NSTATS * sizeof(int64_t));
        return 0;
    }
    // the widest spread whose workspace is there already or can be allocated
    Plan plan;
    size_t ws_bytes = 0;
    for (size_t target = default_workspace_target();; target /= 2) {
        plan = plan_for(in->B, in->N, target, o.rti != 0);
        ws_bytes = kHeader + (size_t)plan.waves * wave_bytes(in->N, nh);
        if (c->ws.ensure(ws_bytes) == hipSuccess) break;
        (void)hipGetLastError();
        if (plan.waves == 1 || target < ((size_t)1 << 28))
            return fail(MPCX_E_NOMEM, "device allocation failed (workspace %zu bytes)", ws_bytes);
    }
    hipStream_t q = c->stream;
    HIP_TRY(hipMemcpyAsync(c->pd.p, in->pdata, n_pd * 8, hipMemcpyHostToDevice, q));
    if (in->n_obs) HIP_TRY(hipMemcpyAsync(c->no.p, in->n_obs, B * 4, hipMemcpyHostToDevice, q));
    else HIP_TRY(hipMemsetAsync(c->no.p, 0, B * 4, q));
    if (in->xguess) HIP_TRY(hipMemcpyAsync(c->xg.p, in->xguess, n_xg * 8, hipMemcpyHostToDevice, q));
    if (in->uguess) HIP_TRY(hipMemcpyAsync(c->ug.p, in->uguess, n_U * 8, hipMemcpyHostToDevice, q));
    if (in->order) HIP_TRY(hipMemcpyAsync(c->ord.p, in->order, B * 4, hipMemcpyHostToDevice, q));
    HIP_TRY(hipMemsetAsync(c->st.p, 0xff, B * 4, q));
    mpcx_batch din = *in;
    din.pdata = c->pd.as<double>();
    din.n_obs = c->no.as<int32_t>();
    din.xguess = in->xguess ? c->xg.as<double>() : nullptr;
    din.uguess = in->uguess ? c->ug.as<double>() : nullptr;
    din.order = in->order ? c->ord.as<int32_t>() : nullptr;
    // X and U, the bulk of the results, are written by the kernel straight into the handle's page-locked
    // host arrays as each problem ends (page-locked memory is addressable from the device), so their
    // transfer runs under the solve instead of behind it; MPCX_CTX_STAGE=1 keeps them in device memory
    // and copies afterwards.  The small per-problem arrays stay on the device (the counters are reduced there).
    mpcx_result dout;
    dout.X = staged ? c->X.as<double>() : out->X;
    dout.U = staged ? c->U.as<double>() : out->U;
    dout.status = c->st.as<int32_t>();
    dout.sqp_iter = c->si.as<int32_t>();
    dout.qp_iter = c->qi.as<int32_t>();
    dout.cost = c->co.as<double>();
    dout.res = c->re.as<double>();
    HIP_TRY(mpcx_launch_sqp(nh, din, o, dout, c->ws.as<double>(), waves_to_launch(plan), plan.lanes, plan.coop_max,
                            plan.lin_max, plan.coop_layout, kSlots, q));
    HIP_TRY(mpcx_launch_stats(dout.status, dout.sqp_iter, dout.qp_iter, in->B, c->stats.as<long long>(), q));
    if (staged) {
        HIP_TRY(hipMemcpyAsync(out->X, dout.X, n_X * 8, hipMemcpyDeviceToHost, q));
        HIP_TRY(hipMemcpyAsync(out->U, dout.U, n_U * 8, hipMemcpyDeviceToHost, q));
    }
    HIP_TRY(hipMemcpyAsync(out->status, dout.status, B * 4, hipMemcpyDeviceToHost, q));
    HIP_TRY(hipMemcpyAsync(out->sqp_iter, dout.sqp_iter, B * 4, hipMemcpyDeviceToHost, q));
    HIP_TRY(hipMemcpyAsync(out->qp_iter, dout.qp_iter, B * 4, hipMemcpyDeviceToHost, q));
    HIP_TRY(hipMemcpyAsync(out->cost, dout.cost, B * 8, hipMemcpyDeviceToHost, q));
    HIP_TRY(hipMemcpyAsync(out->res, dout.res, 4 * B * 8, hipMemcpyDeviceToHost, q));
    HIP_TRY(hipMemcpyAsync(c->hstats.p, c->stats.p, MPCX_NSTATS * 8, hipMemcpyDeviceToHost, q));
    HIP_TRY(hipStreamSynchronize(q));
    if (stats) std::memcpy(stats, c->hstats.p, MPCX_NSTATS * sizeof(int64_t));
    return 0;
}

// The solve for callers that cannot allocate.  Its workspace is the file-scope device array `scratch`
// declared at the top of this file;
// `scratch` names it on the host side only (a stand-in, not a device address): the address on each
// device is looked up once, with that device current, and kept in g_scratch_ptr.
int mpcx_solve_batch_static(const mpcx_batch *in, const mpcx_opts *opts, mpcx_result *out, void *stream)
{
    if (int rc = check_call(in, opts, out)) return rc;
    if (in->B == 0) return 0;
    if (mpcx_device_count() <= 0) return fail(MPCX_E_NODEVICE, "no HIP device");
    const size_t least = kHeader + wave_bytes(in->N, MPCX_MAX_OBS);
    if (least > (size_t)MPCX_STATIC_SCRATCH_BYTES)
        return fail(MPCX_E_BADARG, "N = %d needs %zu bytes of workspace, the built-in scratch holds %zu: use "
                    "mpcx_solve_batch_device with a workspace of the caller's", in->N, least, (size_t)MPCX_STATIC_SCRATCH_BYTES);
    int dev = -1;
    HIP_TRY(hipGetDevice(&dev));
    if (dev < 0 || dev >= kMaxDevices) return fail(MPCX_E_BADARG, "device %d outside [0, %d)", dev, kMaxDevices);
    void *ptr = g_scratch_ptr[dev].load(std::memory_order_acquire);
    if (!ptr) {      // set-up, once per device: this device's copy of the array
        HIP_TRY(hipGetSymbolAddress(&ptr, HIP_SYMBOL(scratch)));
        if (!ptr || (uintptr_t)ptr % 256 != 0) return fail(MPCX_E_HIP, "scratch symbol lookup gave %p", ptr);
        g_scratch_ptr[dev].store(ptr, std::memory_order_release);
    }
    const int nh = (opts->max_obs < 0) ? MPCX_MAX_OBS : opts->max_obs;
    const Plan plan = plan_for(in->B, in->N, (size_t)MPCX_STATIC_SCRATCH_BYTES - kHeader, opts->rti != 0);
    hipStream_t q = static_cast<hipStream_t>(stream);
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(q, &cap) != hipSuccess) { (void)hipGetLastError(); cap = hipStreamCaptureStatusNone; }
    const bool capturing = cap != hipStreamCaptureStatusNone;
    StaticGuard &g = g_guard[dev];
    std::lock_guard<std::mutex> lock(g.mu);
    if (!capturing) {
        if (!g.done) HIP_TRY(hipEventCreateWithFlags(&g.done, hipEventDisableTiming));     // set-up, once per device
        if (g.have && g.last != q && hipEventQuery(g.done) == hipErrorNotReady)
            return fail(MPCX_E_BADARG, "overlapping mpcx_solve_batch_static calls on device %d: the previous call (another "
                        "stream) has not finished and both would use the one built-in scratch", dev);
        (void)hipGetLastError();
    }
    // (the launch zeroes the problem counter in the scratch header itself; everything else a solve
    // reads of the workspace it has written before, so the last call's contents do not matter)
    HIP_TRY(mpcx_launch_sqp(nh, *in, *opts, *out, static_cast<double *>(ptr), waves_to_launch(plan), plan.lanes,
                            plan.coop_max, plan.lin_max, plan.coop_layout, kSlots, q));
    if (!capturing) {
        HIP_TRY(hipEventRecord(g.done, q));
        g.last = q;
        g.have = true;
    }
    return 0;
}

// arithmetic probe used by the parity tests: sin/cos of n host doubles as the
// solver computes them on the device
int mpcx_probe_sincos(const double *x, int32_t n, double *s, double *c, int device)
{
    if (!x || !s || !c || n < 0) return fail(MPCX_E_BADARG, "bad probe arguments");
    if (n == 0) return 0;
    const int ndev = mpcx_device_count();
    if (ndev <= 0) return fail(MPCX_E_NODEVICE, "no HIP device");
    if (device < 0 || device >= ndev) return fail(MPCX_E_BADARG, "device %d out of range [0, %d)", device, ndev);
    HIP_TRY(hipSetDevice(device));
    DevBuf dx, ds, dc;
    if (dx.alloc((size_t)n * 8) != hipSuccess || ds.alloc((size_t)n * 8) != hipSuccess ||
        dc.alloc((size_t)n * 8) != hipSuccess)
        return fail(MPCX_E_NOMEM, "device allocation failed");
    HIP_TRY(hipMemcpy(dx.p, x, (size_t)n * 8, hipMemcpyHostToDevice));
    HIP_TRY(mpcx_launch_probe(dx.as<double>(), n, ds.as<double>(), dc.as<double>(), nullptr));
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpy(s, ds.p, (size_t)n * 8, hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(c, dc.p, (size_t)n * 8, hipMemcpyDeviceToHost));
    return 0;
}

}  // extern "C"
