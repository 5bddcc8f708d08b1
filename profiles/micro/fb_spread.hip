// The two-system recursions of a single-problem interior-point iteration (nx = 5, nu = 2, N stages) by one
// wavefront, as the lane pipelines of the solver (PIPE) and with the algebra of one stage spread over the lanes
// (SPREAD):
//
//   F   forward, both corrections:  du = kv + K dx,  dx(k+1) = A dx + B du        (two right-hand sides q = 0, 1)
//   Bs  backward, both corrections: m = gt + [B A]' p(k+1),  p(k) = m_x + K' m_u  (two right-hand sides)
//
// SPREAD: system q lives in DPP row q (lanes 16 q .. 16 q + 6).  Both recursions are "five values of the
// chain broadcast inside the row -> a five-term multiply-add chain with per-lane coefficients -> two values
// broadcast -> two more multiply-adds":
//   F : lanes (q, 0..1) form du_c from kv_c and row c of K, lanes (q, 2+i) row i of A dx; then the lanes (q, 2+i)
//       add B[i][0] du_0 + B[i][1] du_1 and hold dx(k+1)[i];
//   Bs: lane (q, r) forms m[r] from gt[r] and column r of [B A]; then the lanes (q, 2+i) add
//       K[0][i] m[0] + K[1][i] m[1] and hold p(k)[i]; the lanes (q, 0..1) keep m[0], m[1] (the stage's output).
// The coefficients come from an LDS staging record per stage, one step ahead; the chains are the dense ones
// (every operation of the pattern chains in its order; a structural 1 or 0 changes no finite number).
// Results must equal the pipelines' bit for bit; both are timed with every SIMD busy.
//
// Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fno-fast-math fb_spread.hip -o fb_spread
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#define NX 5
#define NU 2
#define NZ 7
#define DEV __device__ __forceinline__

struct StageIn {
    double ab[14];        // a02 a03 a04 a12 a13 a14 a23 a24 b00 b10 b20 b01 b11 b21
    double K[2][NX], kv[2][NU];      // feedback; feed-forward of the two systems
    double gt[2][NZ];                 // right-hand sides of the two systems (Bs)
};
struct FOut { double dx[2][NX]; };
struct BsOut { double kvo[2][NU]; };

DEV double lane_above0(double v)
{
    const int lo = __builtin_amdgcn_mov_dpp(__double2loint(v), 0x130, 0xf, 0xf, true);
    const int hi = __builtin_amdgcn_mov_dpp(__double2hiint(v), 0x130, 0xf, 0xf, true);
    return __hiloint2double(hi, lo);
}
DEV double lane_below0(double v)
{
    const int lo = __builtin_amdgcn_mov_dpp(__double2loint(v), 0x138, 0xf, 0xf, true);
    const int hi = __builtin_amdgcn_mov_dpp(__double2hiint(v), 0x138, 0xf, 0xf, true);
    return __hiloint2double(hi, lo);
}
template <int L>
DEV double row_bcast(double v)      // the value of lane L of the lane's own row of 16
{
    const int lo = __builtin_amdgcn_mov_dpp(__double2loint(v), 0x150 + L, 0xf, 0xf, true);
    const int hi = __builtin_amdgcn_mov_dpp(__double2hiint(v), 0x150 + L, 0xf, 0xf, true);
    return __hiloint2double(hi, lo);
}
DEV double fresh(double x) { asm volatile("" : "+v"(x)); return x; }

// ---- the solver's pattern code ----
DEV void mul_AB(const double *ab, double hB, const double *dx, const double *du, const double *acc0, double *out)
{
    const double a02 = ab[0], a03 = ab[1], a04 = ab[2], a12 = ab[3], a13 = ab[4], a14 = ab[5],
                 a23 = ab[6], a24 = ab[7], b00 = ab[8], b10 = ab[9], b20 = ab[10], b01 = ab[11], b11 = ab[12], b21 = ab[13];
    out[0] = fma(b01, du[1], fma(b00, du[0], fma(a04, dx[4], fma(a03, dx[3], fma(a02, dx[2], acc0[0] + dx[0])))));
    out[1] = fma(b11, du[1], fma(b10, du[0], fma(a14, dx[4], fma(a13, dx[3], fma(a12, dx[2], acc0[1] + dx[1])))));
    out[2] = fma(b21, du[1], fma(b20, du[0], fma(a24, dx[4], fma(a23, dx[3], acc0[2] + dx[2]))));
    out[3] = fma(hB, du[0], acc0[3] + dx[3]);
    out[4] = fma(hB, du[1], acc0[4] + dx[4]);
}
DEV void F_rec(const double *ab, double hB, const double (*kg)[NX], const double *kv, const double *rb, const double *dx,
               double *du, double *dxn)
{
#pragma unroll
    for (int c = 0; c < NU; ++c) {
        double acc = kv[c];
#pragma unroll
        for (int i = 0; i < NX; ++i) acc = fma(kg[c][i], dx[i], acc);
        du[c] = acc;
    }
    mul_AB(ab, hB, dx, du, rb, dxn);
}
DEV void add_BAt(const double *ab, double hB, const double *w, double *acc)
{
    const double a02 = ab[0], a03 = ab[1], a04 = ab[2], a12 = ab[3], a13 = ab[4], a14 = ab[5],
                 a23 = ab[6], a24 = ab[7], b00 = ab[8], b10 = ab[9], b20 = ab[10], b01 = ab[11], b11 = ab[12], b21 = ab[13];
    acc[0] = fma(hB, w[3], fma(b20, w[2], fma(b10, w[1], fma(b00, w[0], acc[0]))));
    acc[1] = fma(hB, w[4], fma(b21, w[2], fma(b11, w[1], fma(b01, w[0], acc[1]))));
    acc[2] = acc[2] + w[0];
    acc[3] = acc[3] + w[1];
    acc[4] = fma(a12, w[1], fma(a02, w[0], acc[4])) + w[2];
    acc[5] = fma(a23, w[2], fma(a13, w[1], fma(a03, w[0], acc[5]))) + w[3];
    acc[6] = fma(a24, w[2], fma(a14, w[1], fma(a04, w[0], acc[6]))) + w[4];
}

__global__ void __launch_bounds__(64) k_pipe(const StageIn *stages, double hB, int N, int reps, FOut *fout, BsOut *bout,
                                             long long *cycles)
{
    const int kj = (int)threadIdx.x;
    StageIn in = stages[(size_t)blockIdx.x * 64 + kj];
    double dxi[2][NX], dxo[2][NX];
    double o1[NX], o2[NX], kvo[2][NU] = {{0.0, 0.0}, {0.0, 0.0}};
    const double zero5[NX] = {0.0, 0.0, 0.0, 0.0, 0.0};
    long long tF = 0, tB = 0;
    for (int rep = 0; rep < reps; ++rep) {
        long long t0 = __builtin_amdgcn_s_memtime();
        // ---- F, both systems ----
#pragma unroll
        for (int q = 0; q < 2; ++q)
#pragma unroll
            for (int i = 0; i < NX; ++i) { dxi[q][i] = 0.0; dxo[q][i] = 0.0; }
#pragma unroll
        for (int q = 0; q < 14; ++q) in.ab[q] = fresh(in.ab[q]);
        for (int s = 0; s <= N; ++s) {
#pragma unroll
            for (int q = 0; q < 2; ++q) {
#pragma unroll
                for (int i = 0; i < NX; ++i) dxi[q][i] = lane_below0(dxo[q][i]);
                double du[NU];
                F_rec(in.ab, hB, in.K, in.kv[q], zero5, dxi[q], du, dxo[q]);
            }
        }
        long long t1 = __builtin_amdgcn_s_memtime();
        // ---- Bs, both systems ----
#pragma unroll
        for (int i = 0; i < NX; ++i) { o1[i] = (kj == N) ? in.gt[0][NU + i] : 0.0; o2[i] = (kj == N) ? in.gt[1][NU + i] : 0.0; }
#pragma unroll
        for (int q = 0; q < 14; ++q) in.ab[q] = fresh(in.ab[q]);
        for (int s = 0; s < N; ++s) {
            double i1[NX], i2[NX];
#pragma unroll
            for (int i = 0; i < NX; ++i) { i1[i] = lane_above0(o1[i]); i2[i] = lane_above0(o2[i]); }
            if (kj < N) {
#pragma unroll
                for (int sidx = 0; sidx < 2; ++sidx) {
                    double *pn = sidx ? o2 : o1;
                    const double *pin = sidx ? i2 : i1;
                    double w[NX], m[NZ];
#pragma unroll
                    for (int i = 0; i < NX; ++i) w[i] = pin[i];
#pragma unroll
                    for (int j = 0; j < NZ; ++j) m[j] = in.gt[sidx][j];
                    add_BAt(in.ab, hB, w, m);
                    kvo[sidx][0] = m[0]; kvo[sidx][1] = m[1];
#pragma unroll
                    for (int i = 0; i < NX; ++i) pn[i] = fma(in.K[1][i], m[1], fma(in.K[0][i], m[0], m[NU + i]));
                }
            }
        }
        long long t2 = __builtin_amdgcn_s_memtime();
        tF += t1 - t0; tB += t2 - t1;
        in.kv[0][0] = fresh(in.kv[0][0]);
    }
    if (kj <= N) {
        FOut f;
#pragma unroll
        for (int q = 0; q < 2; ++q)
#pragma unroll
            for (int i = 0; i < NX; ++i) f.dx[q][i] = dxi[q][i];
        fout[(size_t)blockIdx.x * 64 + kj] = f;
        BsOut b;
#pragma unroll
        for (int q = 0; q < 2; ++q) { b.kvo[q][0] = kvo[q][0]; b.kvo[q][1] = kvo[q][1]; }
        bout[(size_t)blockIdx.x * 64 + kj] = b;
    }
    if (kj == 0) { cycles[2 * blockIdx.x] = tF; cycles[2 * blockIdx.x + 1] = tB; }
}

// ---------------------------------------------------------------------------------------------------
// SPREAD.  Staging record of a stage (doubles): [0,14) ab; 14 hB; 15 zero; 16 one; [17,27) K row 0, row 1;
// [27,31) kv of the two systems; [31,45) gt of the two systems.  F writes dx(k) of both systems into [0,10) of
// the record of stage k when it reaches it, Bs writes m[0], m[1] of both systems into [0,4).
// ---------------------------------------------------------------------------------------------------
#define REC 45
struct Ops { double st, co[NX], c2[2]; };
DEV int a_word(int i, int j)      // word of A[i][j] in a record
{
    if (i == j) return 16;
    if (i == 0) return (j == 2) ? 0 : (j == 3) ? 1 : (j == 4) ? 2 : 15;
    if (i == 1) return (j == 2) ? 3 : (j == 3) ? 4 : (j == 4) ? 5 : 15;
    if (i == 2) return (j == 3) ? 6 : (j == 4) ? 7 : 15;
    return 15;
}
DEV int b_word(int i, int c)      // word of B[i][c]
{
    if (i < 3) return 8 + 3 * c + i;
    return (i - 3 == c) ? 14 : 15;
}
DEV int ba_word(int c, int j)     // word of entry (j, c) of [B A]
{
    return (c < 2) ? b_word(j, c) : a_word(j, c - 2);
}

template <int CH>
__global__ void __launch_bounds__(64) k_spread(const StageIn *stages, double hB, int N, int reps, FOut *fout, BsOut *bout,
                                               long long *cycles)
{
    __shared__ double rec[CH * REC];
    const int lane = (int)threadIdx.x;
    const int q = (lane >> 4) & 1, rho = lane & 15;
    const bool act = lane < 32 && rho < NZ;
    const int rr = act ? rho : 0;
    const StageIn in = stages[(size_t)blockIdx.x * 64 + lane];
    // ---- per-lane words: F ----
    unsigned fs, fc[NX], f2[2];
    fs = 8u * (unsigned)((rr < 2) ? 27 + 2 * q + rr : 15);                 // start: kv_q[c], or zero (no dynamics residual)
#pragma unroll
    for (int j = 0; j < NX; ++j) fc[j] = 8u * (unsigned)((rr < 2) ? 17 + 5 * rr + j : a_word(rr - 2, j));
#pragma unroll
    for (int c = 0; c < 2; ++c) f2[c] = 8u * (unsigned)((rr < 2) ? 15 : b_word(rr - 2, c));
    // ---- per-lane words: Bs ----
    unsigned bs, bc[NX], b2[2];
    bs = 8u * (unsigned)(31 + 7 * q + rr);
#pragma unroll
    for (int j = 0; j < NX; ++j) bc[j] = 8u * (unsigned)ba_word(rr, j);
#pragma unroll
    for (int c = 0; c < 2; ++c) b2[c] = 8u * (unsigned)((rr < 2) ? 15 : 17 + 5 * c + (rr - 2));
    const unsigned ofx = 8u * (unsigned)((act && rr >= 2) ? 5 * q + rr - 2 : 44);      // F: where the lane's dx entry goes (44: gt2[6], dead in F)
    const unsigned obk = 8u * (unsigned)((act && rr < 2) ? 2 * q + rr : 16 + 0 * 44);  // Bs: where m[0], m[1] go (else: rewrites "one")
    FOut fo = {};
    BsOut bo = {};
    long long tF = 0, tB = 0;
    for (int rep = 0; rep < reps; ++rep) {
        long long t0 = __builtin_amdgcn_s_memtime();
        // =================== F ===================
        double val = 0.0;      // lanes (q, 2+i): dx(k)[i] of system q
        for (int klo = 0; klo <= N; klo += CH) {
            const int khi = (klo + CH - 1 < N) ? klo + CH - 1 : N;
            __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
            __builtin_amdgcn_wave_barrier();
            if (lane >= klo && lane <= khi) {
                double *w = rec + (lane - klo) * REC;
#pragma unroll
                for (int i = 0; i < 14; ++i) w[i] = in.ab[i];
                w[14] = hB; w[15] = 0.0; w[16] = 1.0;
#pragma unroll
                for (int i = 0; i < NX; ++i) { w[17 + i] = in.K[0][i]; w[22 + i] = in.K[1][i]; }
                w[27] = in.kv[0][0]; w[28] = in.kv[0][1]; w[29] = in.kv[1][0]; w[30] = in.kv[1][1];
            }
            __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
            __builtin_amdgcn_wave_barrier();
            char *base = (char *)rec;
            Ops A, B;
            A.st = *(double *)(base + fs);
#pragma unroll
            for (int j = 0; j < NX; ++j) A.co[j] = *(double *)(base + fc[j]);
            A.c2[0] = *(double *)(base + f2[0]); A.c2[1] = *(double *)(base + f2[1]);
            auto stepF = [&](const Ops &cur, Ops &nxt, int k) {
                char *nb = (k < khi) ? base + REC * 8 : base;
                nxt.st = *(double *)(nb + fs);
#pragma unroll
                for (int j = 0; j < NX; ++j) nxt.co[j] = *(double *)(nb + fc[j]);
                nxt.c2[0] = *(double *)(nb + f2[0]); nxt.c2[1] = *(double *)(nb + f2[1]);
                // dx(k) of the lane's system, inside its row; the stage's lane gets it through its record
                const double x0 = row_bcast<2>(val), x1 = row_bcast<3>(val), x2 = row_bcast<4>(val), x3 = row_bcast<5>(val),
                             x4 = row_bcast<6>(val);
                *(double *)(base + ofx) = val;
                double acc = cur.st;
                acc = fma(cur.co[0], x0, acc); acc = fma(cur.co[1], x1, acc); acc = fma(cur.co[2], x2, acc);
                acc = fma(cur.co[3], x3, acc); acc = fma(cur.co[4], x4, acc);
                const double u0 = row_bcast<0>(acc), u1 = row_bcast<1>(acc);
                acc = fma(cur.c2[0], u0, acc);
                acc = fma(cur.c2[1], u1, acc);
                val = acc;
                base = nb;
            };
            for (int k = klo; k <= khi; k += 2) {
                stepF(A, B, k);
                if (k + 1 <= khi) stepF(B, A, k + 1);
            }
            __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
            __builtin_amdgcn_wave_barrier();
            if (lane >= klo && lane <= khi) {
                const double *w = rec + (lane - klo) * REC;
#pragma unroll
                for (int i = 0; i < NX; ++i) { fo.dx[0][i] = w[i]; fo.dx[1][i] = w[5 + i]; }
            }
        }
        long long t1 = __builtin_amdgcn_s_memtime();
        // =================== Bs ===================
        {
            // p(N) = gt_x of the last stage, from its lane
            double e = 0.0;
#pragma unroll
            for (int i = 0; i < NX; ++i) {
                const double g0 = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(in.gt[0][NU + i]), N),
                                                   __builtin_amdgcn_readlane(__double2loint(in.gt[0][NU + i]), N));
                const double g1 = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(in.gt[1][NU + i]), N),
                                                   __builtin_amdgcn_readlane(__double2loint(in.gt[1][NU + i]), N));
                e = (act && rr == 2 + i) ? (q ? g1 : g0) : e;
            }
            val = e;
        }
        for (int khi = N - 1; khi >= 0; khi -= CH) {
            const int klo = (khi - CH + 1 > 0) ? khi - CH + 1 : 0;
            __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
            __builtin_amdgcn_wave_barrier();
            if (lane >= klo && lane <= khi) {
                double *w = rec + (lane - klo) * REC;
#pragma unroll
                for (int i = 0; i < 14; ++i) w[i] = in.ab[i];
                w[14] = hB; w[15] = 0.0; w[16] = 1.0;
#pragma unroll
                for (int i = 0; i < NX; ++i) { w[17 + i] = in.K[0][i]; w[22 + i] = in.K[1][i]; }
#pragma unroll
                for (int i = 0; i < NZ; ++i) { w[31 + i] = in.gt[0][i]; w[38 + i] = in.gt[1][i]; }
            }
            __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
            __builtin_amdgcn_wave_barrier();
            char *base = (char *)rec + (size_t)(khi - klo) * REC * 8;
            Ops A, B;
            A.st = *(double *)(base + bs);
#pragma unroll
            for (int j = 0; j < NX; ++j) A.co[j] = *(double *)(base + bc[j]);
            A.c2[0] = *(double *)(base + b2[0]); A.c2[1] = *(double *)(base + b2[1]);
            auto stepB = [&](const Ops &cur, Ops &nxt, int k) {
                char *nb = (k > klo) ? base - REC * 8 : base;
                nxt.st = *(double *)(nb + bs);
#pragma unroll
                for (int j = 0; j < NX; ++j) nxt.co[j] = *(double *)(nb + bc[j]);
                nxt.c2[0] = *(double *)(nb + b2[0]); nxt.c2[1] = *(double *)(nb + b2[1]);
                const double w0 = row_bcast<2>(val), w1 = row_bcast<3>(val), w2 = row_bcast<4>(val), w3 = row_bcast<5>(val),
                             w4 = row_bcast<6>(val);
                double m = cur.st;
                m = fma(cur.co[0], w0, m); m = fma(cur.co[1], w1, m); m = fma(cur.co[2], w2, m); m = fma(cur.co[3], w3, m);
                m = fma(cur.co[4], w4, m);
                const double m0 = row_bcast<0>(m), m1 = row_bcast<1>(m);
                *(double *)(base + obk) = (act && rr < 2) ? m : 1.0;
                double acc = fma(cur.c2[0], m0, m);
                acc = fma(cur.c2[1], m1, acc);
                val = acc;
                base = nb;
            };
            for (int k = khi; k >= klo; k -= 2) {
                stepB(A, B, k);
                if (k - 1 >= klo) stepB(B, A, k - 1);
            }
            __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
            __builtin_amdgcn_wave_barrier();
            if (lane >= klo && lane <= khi) {
                const double *w = rec + (lane - klo) * REC;
                bo.kvo[0][0] = w[0]; bo.kvo[0][1] = w[1]; bo.kvo[1][0] = w[2]; bo.kvo[1][1] = w[3];
            }
        }
        long long t2 = __builtin_amdgcn_s_memtime();
        tF += t1 - t0; tB += t2 - t1;
        val = fresh(val);
    }
    if (lane <= N) {
        fout[(size_t)blockIdx.x * 64 + lane] = fo;
        bout[(size_t)blockIdx.x * 64 + lane] = bo;
    }
    if (lane == 0) { cycles[2 * blockIdx.x] = tF; cycles[2 * blockIdx.x + 1] = tB; }
}

__global__ void k_dpp_check(int *o)
{
    const double v = 100.0 + threadIdx.x;
    o[threadIdx.x] = (int)row_bcast<3>(v);
}

static double urand(unsigned long long &s)
{
    s = s * 6364136223846793005ull + 1442695040888963407ull;
    return (double)(s >> 11) / 9007199254740992.0;
}

int main(int argc, char **argv)
{
    const int N = 50, WG = (argc > 1) ? atoi(argv[1]) : 1024, reps = (argc > 2) ? atoi(argv[2]) : 200;
    const double h = 0.1, hB = (h / 6.0) * 6.0;
    {
        int *d; hipMalloc(&d, 256);
        hipLaunchKernelGGL(k_dpp_check, dim3(1), dim3(64), 0, 0, d);
        int hh[64]; hipMemcpy(hh, d, 256, hipMemcpyDeviceToHost);
        bool ok = true;
        for (int i = 0; i < 64; ++i) ok = ok && hh[i] == 100 + (i & ~15) + 3;
        printf("row_newbcast:3 gives every lane the value of lane 3 of its row of 16: %s\n", ok ? "yes" : "NO");
        if (!ok) return 3;
    }
    std::vector<StageIn> st((size_t)WG * 64);
    unsigned long long seed = 4321;
    for (auto &s : st) {
        const double v = 2.0 * urand(seed) - 0.5, th = 6.0 * urand(seed) - 3.0;
        s.ab[0] = -h * v * sin(th); s.ab[1] = h * cos(th); s.ab[2] = 1e-3 * (urand(seed) - 0.5);
        s.ab[3] = h * v * cos(th);  s.ab[4] = h * sin(th); s.ab[5] = 1e-3 * (urand(seed) - 0.5);
        s.ab[6] = h * 0.1 * (urand(seed) - 0.5); s.ab[7] = h * v / 2.7 * (1.0 + urand(seed));
        s.ab[8] = 0.005 * cos(th); s.ab[9] = 0.005 * sin(th); s.ab[10] = 1e-4 * (urand(seed) - 0.5);
        s.ab[11] = 1e-5 * (urand(seed) - 0.5); s.ab[12] = 1e-5 * (urand(seed) - 0.5); s.ab[13] = 0.005 * v / 2.7;
        for (int c = 0; c < 2; ++c)
            for (int i = 0; i < NX; ++i) s.K[c][i] = -(0.2 + urand(seed)) * ((i + c) % 3 ? 1.0 : -0.3);
        for (int q = 0; q < 2; ++q) {
            for (int c = 0; c < NU; ++c) s.kv[q][c] = 1e-2 * (urand(seed) - 0.5);
            for (int j = 0; j < NZ; ++j) s.gt[q][j] = pow(10.0, 4.0 * urand(seed) - 2.0) * (urand(seed) - 0.5);
        }
    }
    StageIn *d_st; FOut *d_f1, *d_f2; BsOut *d_b1, *d_b2; long long *d_cy;
    hipMalloc(&d_st, st.size() * sizeof(StageIn));
    hipMalloc(&d_f1, st.size() * sizeof(FOut)); hipMalloc(&d_f2, st.size() * sizeof(FOut));
    hipMalloc(&d_b1, st.size() * sizeof(BsOut)); hipMalloc(&d_b2, st.size() * sizeof(BsOut));
    hipMalloc(&d_cy, (size_t)WG * 16);
    hipMemcpy(d_st, st.data(), st.size() * sizeof(StageIn), hipMemcpyHostToDevice);
    std::vector<FOut> f1(st.size()), f2(st.size());
    std::vector<BsOut> b1(st.size()), b2(st.size());
    std::vector<long long> cy((size_t)WG * 2);
    auto report = [&](const char *name) {
        hipMemcpy(cy.data(), d_cy, (size_t)WG * 16, hipMemcpyDeviceToHost);
        double sf = 0.0, sb = 0.0;
        for (int w = 0; w < WG; ++w) { sf += (double)cy[2 * w]; sb += (double)cy[2 * w + 1]; }
        printf("%-26s F (both systems) %7.1f cycles per stage, Bs (both systems) %7.1f cycles per stage (mean over %d wavefronts)\n",
               name, sf / WG / ((double)reps * (N + 1)), sb / WG / ((double)reps * N), WG);
    };
    for (int pass = 0; pass < 2; ++pass) {
        hipMemset(d_f1, 0, st.size() * sizeof(FOut)); hipMemset(d_b1, 0, st.size() * sizeof(BsOut));
        hipLaunchKernelGGL(k_pipe, dim3(WG), dim3(64), 0, 0, d_st, hB, N, reps, d_f1, d_b1, d_cy);
        hipDeviceSynchronize();
        if (pass) report("PIPE (lane pipelines)");
    }
    hipMemcpy(f1.data(), d_f1, st.size() * sizeof(FOut), hipMemcpyDeviceToHost);
    hipMemcpy(b1.data(), d_b1, st.size() * sizeof(BsOut), hipMemcpyDeviceToHost);
    bool ok = true;
    auto check = [&](const char *name) {
        hipMemcpy(f2.data(), d_f2, st.size() * sizeof(FOut), hipMemcpyDeviceToHost);
        hipMemcpy(b2.data(), d_b2, st.size() * sizeof(BsOut), hipMemcpyDeviceToHost);
        size_t badf = 0, badb = 0;
        for (int w = 0; w < WG; ++w) {
            for (int k = 0; k <= N; ++k)
                if (memcmp(&f1[(size_t)w * 64 + k], &f2[(size_t)w * 64 + k], sizeof(FOut)) != 0) ++badf;
            for (int k = 0; k < N; ++k)
                if (memcmp(&b1[(size_t)w * 64 + k], &b2[(size_t)w * 64 + k], sizeof(BsOut)) != 0) ++badb;
        }
        printf("%-26s dx of every stage: %zu differ; right-hand sides of every stage: %zu differ (%d problems)\n", name, badf, badb, WG);
        ok = ok && badf == 0 && badb == 0;
    };
#define RUN(CH_)                                                                                         \
    for (int pass = 0; pass < 2; ++pass) {                                                               \
        hipMemset(d_f2, 0, st.size() * sizeof(FOut)); hipMemset(d_b2, 0, st.size() * sizeof(BsOut));     \
        hipLaunchKernelGGL(k_spread<CH_>, dim3(WG), dim3(64), 0, 0, d_st, hB, N, reps, d_f2, d_b2, d_cy); \
        hipDeviceSynchronize();                                                                          \
        if (pass) { report("SPREAD, " #CH_ " stages staged"); check("SPREAD, " #CH_ " stages staged"); } \
    }
    RUN(64)
    RUN(24)
    if (hipDeviceSynchronize() != hipSuccess) { printf("HIP error\n"); return 2; }
    return ok ? 0 : 1;
}
