// The backward Riccati recursion of ONE problem (nx = 5, nu = 2, N stages) by one wavefront -- factorisation AND
// the backward substitution of one right-hand side -- two ways:
//
//   PIPE     as the shipped single-problem mode does it: lane k owns stage k; a lane pipeline of N steps for
//            the factors (P(k) from P(k+1), handed down with DPP wave_shl:1), P(k+1) rb once per stage, then a
//            second lane pipeline of N steps for p(k);
//   SPREAD3  one serial loop over the stages with the algebra of the current stage spread over the lanes:
//            lane 8 r + c, r, c < 7, owns entry (r, c) of M = Ht + [B A]' P [B A]; the lanes 8 r + 7 own the
//            right-hand side as an eighth column (P(k+1) rb instead of a column of P [B A], p(k+1) added, the
//            stage's gradient instead of the Hessian entry; the same multiply-add chains then give
//            m = gt + [B A]' (p + P rb), and p(k) = m_x + K' m_u comes out of the instructions that give
//            P(k) = M_xx + M_xu K in the other lanes, with other exchange partners).  P(k+1), p(k+1) travel as
//            20 wave-uniform values; the operands of the stages come from an LDS staging area, CH stages at a time.
//
// Every number is produced by the operations of the CPU restatement's riccati_factor() / riccati_solve() in
// their order, so all results must agree bit for bit: the program checks that, then times both with every
// SIMD of the GPU busy (4 one-wavefront workgroups per CU, as the solver runs).
//
// Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fno-fast-math ric_spread3.hip -o ric_spread3
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
    double Ht[NZ], hxx, hxy, hyy;
    double gt[NZ], rb[NX];
};
struct StageOut { double K[2][NX], l10, d0, d1, kv[2]; };

DEV constexpr int tri(int i, int j) { return i * (i + 1) / 2 + j; }
DEV double psym(const double *P, int i, int j) { return (i >= j) ? P[tri(i, j)] : P[tri(j, i)]; }
DEV double lane_above0(double v)
{
    const int lo = __builtin_amdgcn_mov_dpp(__double2loint(v), 0x130, 0xf, 0xf, true);
    const int hi = __builtin_amdgcn_mov_dpp(__double2hiint(v), 0x130, 0xf, 0xf, true);
    return __hiloint2double(hi, lo);
}
DEV double rdlane(double v, int src)
{
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), src);
    const int hi = __builtin_amdgcn_readlane(__double2hiint(v), src);
    return __hiloint2double(hi, lo);
}
DEV double bperm(int addr, double v)
{
    const int lo = __builtin_amdgcn_ds_bpermute(addr, __double2loint(v));
    const int hi = __builtin_amdgcn_ds_bpermute(addr, __double2hiint(v));
    return __hiloint2double(hi, lo);
}
DEV double fresh(double x) { asm volatile("" : "+v"(x)); return x; }

// ---- the shipped kernel's stage code (sparsity pattern of A and B) ----
DEV void ric_stage(const StageIn &in, double hB, const double *Pn, double *Po, StageOut &ro)
{
    const double a02 = in.ab[0], a03 = in.ab[1], a04 = in.ab[2], a12 = in.ab[3], a13 = in.ab[4], a14 = in.ab[5],
                 a23 = in.ab[6], a24 = in.ab[7], b00 = in.ab[8], b10 = in.ab[9], b20 = in.ab[10], b01 = in.ab[11],
                 b11 = in.ab[12], b21 = in.ab[13];
    double PBA[NX][NZ];
#pragma unroll
    for (int i = 0; i < NX; ++i) {
        const double p0 = psym(Pn, i, 0), p1 = psym(Pn, i, 1), p2 = psym(Pn, i, 2), p3 = psym(Pn, i, 3), p4 = psym(Pn, i, 4);
        PBA[i][0] = fma(p3, hB, fma(p2, b20, fma(p1, b10, p0 * b00)));
        PBA[i][1] = fma(p4, hB, fma(p2, b21, fma(p1, b11, p0 * b01)));
        PBA[i][2] = p0;
        PBA[i][3] = p1;
        PBA[i][4] = fma(p1, a12, p0 * a02) + p2;
        PBA[i][5] = fma(p2, a23, fma(p1, a13, p0 * a03)) + p3;
        PBA[i][6] = fma(p2, a24, fma(p1, a14, p0 * a04)) + p4;
    }
    double Mm[NZ][NZ];
#pragma unroll
    for (int c = 0; c < NZ; ++c) {
#pragma unroll
        for (int r = c; r < NZ; ++r) {
            double acc = 0.0;
            if (r == c) acc = (r == 2) ? in.Ht[2] + in.hxx : (r == 3) ? in.Ht[3] + in.hyy : in.Ht[r];
            else if (r == 3 && c == 2) acc = in.hxy;
            if (r == 0) acc = fma(hB, PBA[3][c], fma(b20, PBA[2][c], fma(b10, PBA[1][c], fma(b00, PBA[0][c], acc))));
            else if (r == 1) acc = fma(hB, PBA[4][c], fma(b21, PBA[2][c], fma(b11, PBA[1][c], fma(b01, PBA[0][c], acc))));
            else if (r == 2) acc = acc + PBA[0][c];
            else if (r == 3) acc = acc + PBA[1][c];
            else if (r == 4) acc = fma(a12, PBA[1][c], fma(a02, PBA[0][c], acc)) + PBA[2][c];
            else if (r == 5) acc = fma(a23, PBA[2][c], fma(a13, PBA[1][c], fma(a03, PBA[0][c], acc))) + PBA[3][c];
            else acc = fma(a24, PBA[2][c], fma(a14, PBA[1][c], fma(a04, PBA[0][c], acc))) + PBA[4][c];
            Mm[r][c] = acc;
        }
    }
    const double d0 = 1.0 / Mm[0][0];
    const double l10 = Mm[1][0] * d0;
    const double d1 = 1.0 / fma(-l10, Mm[1][0], Mm[1][1]);
    ro.l10 = l10; ro.d0 = d0; ro.d1 = d1;
    double Kg[NU][NX];
#pragma unroll
    for (int i = 0; i < NX; ++i) {
        const double y0 = Mm[NU + i][0];
        const double y1 = fma(-l10, y0, Mm[NU + i][1]);
        const double z0 = y0 * d0, z1 = y1 * d1;
        const double s0 = fma(-l10, z1, z0);
        Kg[0][i] = -s0; Kg[1][i] = -z1;
        ro.K[0][i] = Kg[0][i]; ro.K[1][i] = Kg[1][i];
    }
#pragma unroll
    for (int i = 0; i < NX; ++i)
#pragma unroll
        for (int j = 0; j <= i; ++j) {
            const double e = fma(Mm[NU + i][0], Kg[0][j], Mm[NU + i][NU + j]);
            Po[tri(i, j)] = fma(Mm[NU + i][1], Kg[1][j], e);
        }
}
DEV void add_BAt(const double *ab, double hB, const double *w, double *acc)
{
    const double a02 = ab[0], a03 = ab[1], a04 = ab[2], a12 = ab[3], a13 = ab[4], a14 = ab[5],
                 a23 = ab[6], a24 = ab[7], b00 = ab[8], b10 = ab[9], b20 = ab[10], b01 = ab[11],
                 b11 = ab[12], b21 = ab[13];
    acc[0] = fma(hB, w[3], fma(b20, w[2], fma(b10, w[1], fma(b00, w[0], acc[0]))));
    acc[1] = fma(hB, w[4], fma(b21, w[2], fma(b11, w[1], fma(b01, w[0], acc[1]))));
    acc[2] = acc[2] + w[0];
    acc[3] = acc[3] + w[1];
    acc[4] = fma(a12, w[1], fma(a02, w[0], acc[4])) + w[2];
    acc[5] = fma(a23, w[2], fma(a13, w[1], fma(a03, w[0], acc[5]))) + w[3];
    acc[6] = fma(a24, w[2], fma(a14, w[1], fma(a04, w[0], acc[6]))) + w[4];
}
DEV void sub_stage(const StageIn &in, double hB, const double *prb, const double *pn, double *po, StageOut &ro)
{
    double w[NX], m[NZ];
#pragma unroll
    for (int i = 0; i < NX; ++i) w[i] = pn[i] + prb[i];
#pragma unroll
    for (int j = 0; j < NZ; ++j) m[j] = in.gt[j];
    add_BAt(in.ab, hB, w, m);
    ro.kv[0] = m[0]; ro.kv[1] = m[1];
#pragma unroll
    for (int i = 0; i < NX; ++i) po[i] = fma(ro.K[1][i], m[1], fma(ro.K[0][i], m[0], m[NU + i]));
}
DEV double ht_entry(const StageIn &in, int i, int j)
{
    if (i == j) return (i == 0) ? in.Ht[2] + in.hxx : (i == 1) ? in.Ht[3] + in.hyy : in.Ht[NU + i];
    return (i == 1 && j == 0) ? in.hxy : 0.0;
}

__global__ void __launch_bounds__(64) k_pipe(const StageIn *stages, double hB, int N, int reps, StageOut *out, double *P0,
                                             long long *cycles)
{
    const int kj = (int)threadIdx.x;
    StageIn in = stages[(size_t)blockIdx.x * 64 + kj];
    StageOut ro = {};
    double Po[15], po[NX];
    long long t0 = __builtin_amdgcn_s_memtime();
    for (int rep = 0; rep < reps; ++rep) {
#pragma unroll
        for (int i = 0; i < NX; ++i) {
#pragma unroll
            for (int j = 0; j <= i; ++j) Po[tri(i, j)] = (kj == N) ? ht_entry(in, i, j) : 0.0;
            po[i] = (kj == N) ? in.gt[NU + i] : 0.0;
        }
#pragma unroll
        for (int q = 0; q < 14; ++q) in.ab[q] = fresh(in.ab[q]);
        for (int s = 0; s < N; ++s) {
            double Pi[15];
#pragma unroll
            for (int i = 0; i < 15; ++i) Pi[i] = lane_above0(Po[i]);
            if (kj < N) ric_stage(in, hB, Pi, Po, ro);
        }
        double prb[NX];
        {
            double Pi[15];
#pragma unroll
            for (int i = 0; i < 15; ++i) Pi[i] = lane_above0(Po[i]);
#pragma unroll
            for (int i = 0; i < NX; ++i) {
                double acc = psym(Pi, i, 0) * in.rb[0];
#pragma unroll
                for (int j = 1; j < NX; ++j) acc = fma(psym(Pi, i, j), in.rb[j], acc);
                prb[i] = acc;
            }
        }
#pragma unroll
        for (int q = 0; q < 14; ++q) in.ab[q] = fresh(in.ab[q]);
        for (int s = 0; s < N; ++s) {
            double pi[NX];
#pragma unroll
            for (int i = 0; i < NX; ++i) pi[i] = lane_above0(po[i]);
            if (kj < N) sub_stage(in, hB, prb, pi, po, ro);
        }
        in.Ht[0] = fresh(in.Ht[0]);
    }
    long long t1 = __builtin_amdgcn_s_memtime();
    if (kj <= N) out[(size_t)blockIdx.x * 64 + kj] = ro;
    if (kj == 0) {
        for (int i = 0; i < 15; ++i) P0[(size_t)blockIdx.x * 20 + i] = Po[i];
        for (int i = 0; i < NX; ++i) P0[(size_t)blockIdx.x * 20 + 15 + i] = po[i];
        cycles[blockIdx.x] = t1 - t0;
    }
}

// ---------------------------------------------------------------------------------------------------
// SPREAD3.  Staging record of a stage (doubles): [0,35) [B A] dense by column, column c at 5 c; [35,42) the
// diagonal of the barrier Hessian (the (x, y) block's already added); 42 hxy; 43 zero; [44,49) rb; [49,56) gt.
// When the stage is done its record receives the results: [0,5) K row 0, [5,10) K row 1, 10 l10, 11 d0, 12 d1,
// 13, 14 the input part of the right-hand side (read back by the stage's lane).
// ---------------------------------------------------------------------------------------------------
#define REC 56

// VAR bit 0: the four exchanges and the pivot broadcasts are issued before the divisions (scheduling barrier);
//     bit 1: P(k+1), p(k+1) are broadcast through 20 words of LDS instead of 40 v_readlane;
//     bit 2: the loop over the stages is unrolled twice (no register copies for the operand prefetch)
template <int CH, int VAR>
__global__ void __launch_bounds__(64) k_spread3(const StageIn *stages, double hB, int N, int reps, StageOut *out, double *P0,
                                                long long *cycles)
{
    __shared__ double rec[CH * REC + 8];
    __shared__ __attribute__((aligned(16))) double pbc[24];
    const int lane = (int)threadIdx.x;
    const int r = lane >> 3, c = lane & 7;
    const bool ent = r < NZ;                                // the lane owns an entry of M (c < 7) or of the right-hand side (c = 7)
    const bool rhs = ent && c == 7;
    const int rr = ent ? r : 0, cc = ent ? c : 0;           // (lanes without an entry run along on entry (0,0))
    const StageIn in = stages[(size_t)blockIdx.x * 64 + lane];      // the lane's OWN stage (lane = stage)
    StageOut ro = {};
    const unsigned off_c = (cc == 7) ? 8u * 44u : 8u * 5u * (unsigned)cc, off_r = 8u * 5u * (unsigned)rr;
    const unsigned off_h = (cc == 7) ? 8u * (unsigned)(49 + rr)
                                     : 8u * (unsigned)((rr == cc) ? 35 + rr : (rr == 3 && cc == 2) ? 42 : 43);
    // results: diagonal lanes (NU+j, NU+j): K[0][j] -> j, K[1][j] -> 5 + j; lane (0,0): l10, d0 (and d1) -> 10, 11, 12;
    // lanes (0,7), (1,7): their entry of the right-hand side -> 13, 14; the others into a spare word
    const bool diag = ent && cc < 7 && rr == cc && rr >= NU;
    const bool kvl = rhs && rr < NU;
    const int spare = CH * REC;
    const int oA = diag ? (cc - NU) : (lane == 0) ? 10 : kvl ? 13 + rr : -1;
    const int oB = diag ? (5 + cc - NU) : (lane == 0) ? 11 : -1;
    // exchange partners.  Entry lanes: M[r][0], M[r][1] of the lane's row and M[c][0], M[c][1] of its column.
    // Right-hand-side lanes: the input part m[0], m[1] of the right-hand side, and M[r][0], M[r][1] of the row
    const int src_r0 = 4 * (rhs ? 8 * 0 + 7 : 8 * rr + 0), src_r1 = 4 * (rhs ? 8 * 1 + 7 : 8 * rr + 1);
    const int src_c0 = 4 * (rhs ? 8 * rr + 0 : 8 * cc + 0), src_c1 = 4 * (rhs ? 8 * rr + 1 : 8 * cc + 1);
    int pslot = -1;
#pragma unroll
    for (int i = 0; i < NX; ++i) {
#pragma unroll
        for (int j = 0; j <= i; ++j)
            if (lane == 8 * (NU + i) + (NU + j)) pslot = tri(i, j);
        if (lane == 8 * (NU + i) + 7) pslot = 15 + i;
    }
    double pnew = 0.0;
    long long t0 = __builtin_amdgcn_s_memtime();
    for (int rep = 0; rep < reps; ++rep) {
        {   // P(N) and p(N) of the last stage, entry by entry from lane N
            double e = 0.0;
#pragma unroll
            for (int i = 0; i < NX; ++i) {
#pragma unroll
                for (int j = 0; j <= i; ++j) {
                    const double v = rdlane(ht_entry(in, i, j), N);
                    e = (rr == NU + i && cc == NU + j) ? v : e;
                }
                const double g = rdlane(in.gt[NU + i], N);
                e = (rr == NU + i && cc == 7) ? g : e;
            }
            pnew = e;
        }
        for (int khi = N - 1; khi >= 0; khi -= CH) {
            const int klo = (khi - CH + 1 > 0) ? khi - CH + 1 : 0;
            __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
            __builtin_amdgcn_wave_barrier();
            if (lane >= klo && lane <= khi) {
                double *w = rec + (lane - klo) * REC;
                const double *ab = in.ab;
                w[0] = ab[8];  w[1] = ab[9];  w[2] = ab[10]; w[3] = hB;  w[4] = 0.0;
                w[5] = ab[11]; w[6] = ab[12]; w[7] = ab[13]; w[8] = 0.0; w[9] = hB;
                w[10] = 1.0; w[11] = 0.0; w[12] = 0.0; w[13] = 0.0; w[14] = 0.0;
                w[15] = 0.0; w[16] = 1.0; w[17] = 0.0; w[18] = 0.0; w[19] = 0.0;
                w[20] = ab[0]; w[21] = ab[3]; w[22] = 1.0; w[23] = 0.0; w[24] = 0.0;
                w[25] = ab[1]; w[26] = ab[4]; w[27] = ab[6]; w[28] = 1.0; w[29] = 0.0;
                w[30] = ab[2]; w[31] = ab[5]; w[32] = ab[7]; w[33] = 0.0; w[34] = 1.0;
#pragma unroll
                for (int q = 0; q < NZ; ++q) w[35 + q] = (q == 2) ? in.Ht[2] + in.hxx : (q == 3) ? in.Ht[3] + in.hyy : in.Ht[q];
                w[42] = in.hxy;
                w[43] = 0.0;
#pragma unroll
                for (int i = 0; i < NX; ++i) w[44 + i] = in.rb[i];
#pragma unroll
                for (int q = 0; q < NZ; ++q) w[49 + q] = in.gt[q];
            }
            __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
            __builtin_amdgcn_wave_barrier();
            const char *base = (const char *)rec + (size_t)(khi - klo) * REC * 8;
            double bc[NX], br[NX], ht;
#pragma unroll
            for (int i = 0; i < NX; ++i) { bc[i] = *(const double *)(base + off_c + 8 * i); br[i] = *(const double *)(base + off_r + 8 * i); }
            ht = *(const double *)(base + off_h);
#pragma unroll((VAR & 4) ? 2 : 1)
            for (int k = khi; k >= klo; --k) {
                double nbc[NX], nbr[NX], nht;
                const char *nb = (k > klo) ? base - REC * 8 : base;
#pragma unroll
                for (int i = 0; i < NX; ++i) { nbc[i] = *(const double *)(nb + off_c + 8 * i); nbr[i] = *(const double *)(nb + off_r + 8 * i); }
                nht = *(const double *)(nb + off_h);
                // P(k+1), p(k+1): wave-uniform
                double P[15], pn[NX];
                if (VAR & 2) {
                    if (pslot >= 0) pbc[pslot] = pnew;
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                    __builtin_amdgcn_wave_barrier();
#pragma unroll
                    for (int i = 0; i < 15; ++i) P[i] = *(volatile double *)&pbc[i];
#pragma unroll
                    for (int i = 0; i < NX; ++i) pn[i] = *(volatile double *)&pbc[15 + i];
                } else {
#pragma unroll
                    for (int i = 0; i < NX; ++i) {
#pragma unroll
                        for (int j = 0; j <= i; ++j) P[tri(i, j)] = rdlane(pnew, 8 * (NU + i) + (NU + j));
                        pn[i] = rdlane(pnew, 8 * (NU + i) + 7);
                    }
                }
                // column c of P [B A] (right-hand-side lanes: P rb, then + p: adding -0 changes no number)
                double pba[NX];
#pragma unroll
                for (int i = 0; i < NX; ++i) {
                    double acc = psym(P, i, 0) * bc[0];
#pragma unroll
                    for (int j = 1; j < NX; ++j) acc = fma(psym(P, i, j), bc[j], acc);
                    pba[i] = (rhs ? pn[i] : -0.0) + acc;
                }
                double m = ht;
#pragma unroll
                for (int i = 0; i < NX; ++i) m = fma(br[i], pba[i], m);
                const double m00 = rdlane(m, 0), m10 = rdlane(m, 8), m11 = rdlane(m, 9);
                double mr0, mr1, mc0, mc1;
                if (VAR & 1) {
                    mr0 = bperm(src_r0, m); mr1 = bperm(src_r1, m); mc0 = bperm(src_c0, m); mc1 = bperm(src_c1, m);
                    __builtin_amdgcn_sched_barrier(0);
                }
                const double d0 = 1.0 / m00;
                const double l10 = m10 * d0;
                const double d1 = 1.0 / fma(-l10, m10, m11);
                if (!(VAR & 1)) { mr0 = bperm(src_r0, m); mr1 = bperm(src_r1, m); mc0 = bperm(src_c0, m); mc1 = bperm(src_c1, m); }
                const double y1 = fma(-l10, mc0, mc1);
                const double z0 = mc0 * d0, z1 = y1 * d1;
                const double s0 = fma(-l10, z1, z0);
                const double k0j = -s0, k1j = -z1;
                const double e = fma(mr0, k0j, m);
                pnew = fma(mr1, k1j, e);
                {
                    double *ob = rec + (k - klo) * REC;
                    const int back = (k - klo) * REC;
                    ob[(oA >= 0) ? oA : spare - back] = diag ? k0j : kvl ? m : l10;
                    ob[(oB >= 0) ? oB : spare - back] = diag ? k1j : d0;
                    if (lane == 0) ob[12] = d1;
                }
#pragma unroll
                for (int i = 0; i < NX; ++i) { bc[i] = nbc[i]; br[i] = nbr[i]; }
                ht = nht;
                base = nb;
            }
            __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
            __builtin_amdgcn_wave_barrier();
            if (lane >= klo && lane <= khi) {
                const double *w = rec + (lane - klo) * REC;
#pragma unroll
                for (int i = 0; i < NX; ++i) { ro.K[0][i] = w[i]; ro.K[1][i] = w[5 + i]; }
                ro.l10 = w[10]; ro.d0 = w[11]; ro.d1 = w[12]; ro.kv[0] = w[13]; ro.kv[1] = w[14];
            }
        }
        pnew = fresh(pnew);
    }
    long long t1 = __builtin_amdgcn_s_memtime();
    if (lane <= N) out[(size_t)blockIdx.x * 64 + lane] = ro;
#pragma unroll
    for (int i = 0; i < NX; ++i) {
#pragma unroll
        for (int j = 0; j <= i; ++j)
            if (lane == 8 * (NU + i) + (NU + j)) P0[(size_t)blockIdx.x * 20 + tri(i, j)] = pnew;
        if (lane == 8 * (NU + i) + 7) P0[(size_t)blockIdx.x * 20 + 15 + i] = pnew;
    }
    if (lane == 0) cycles[blockIdx.x] = t1 - t0;
}

// ---------------------------------------------------------------------------------------------------
// SPREAD4 = SPREAD3 (exchanges before the divisions) on COMPACT staging records, so that more stages fit the
// LDS the solver has to spare: a record holds the 14 pattern entries of [B A], then hB, 0, 1, and every lane
// reads its five column and five row operands through per-lane word indices.  Record (doubles): [0,14) ab;
// 14 hB; 15 zero; 16 one; [17,24) diagonal of the barrier Hessian; 24 hxy; [25,30) rb; [30,37) gt.  Results of
// a stage go into its record as in SPREAD3.
// ---------------------------------------------------------------------------------------------------
#define REC4 37
DEV int ba_word(int c, int j)      // word of entry (j, c) of [B A | rb] in a compact record
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

template <int CH>
__global__ void __launch_bounds__(64) k_spread4(const StageIn *stages, double hB, int N, int reps, StageOut *out, double *P0,
                                                long long *cycles)
{
    __shared__ double rec[CH * REC4 + 1];
    const int lane = (int)threadIdx.x;
    const int r = lane >> 3, c = lane & 7;
    const bool ent = r < NZ;
    const bool rhs = ent && c == 7;
    const int rr = ent ? r : 0, cc = ent ? c : 0;
    const StageIn in = stages[(size_t)blockIdx.x * 64 + lane];
    StageOut ro = {};
    unsigned oc[NX], orr[NX];
#pragma unroll
    for (int j = 0; j < NX; ++j) { oc[j] = 8u * (unsigned)ba_word(cc, j); orr[j] = 8u * (unsigned)ba_word(rr, j); }
    const unsigned off_h = (cc == 7) ? 8u * (unsigned)(30 + rr)
                                     : 8u * (unsigned)((rr == cc) ? 17 + rr : (rr == 3 && cc == 2) ? 24 : 15);
    const bool diag = ent && cc < 7 && rr == cc && rr >= NU;
    const bool kvl = rhs && rr < NU;
    const int spare = CH * REC4;
    const int oA = diag ? (cc - NU) : (lane == 0) ? 10 : kvl ? 13 + rr : -1;
    const int oB = diag ? (5 + cc - NU) : (lane == 0) ? 11 : -1;
    const int src_r0 = 4 * (rhs ? 8 * 0 + 7 : 8 * rr + 0), src_r1 = 4 * (rhs ? 8 * 1 + 7 : 8 * rr + 1);
    const int src_c0 = 4 * (rhs ? 8 * rr + 0 : 8 * cc + 0), src_c1 = 4 * (rhs ? 8 * rr + 1 : 8 * cc + 1);
    double pnew = 0.0;
    long long t0 = __builtin_amdgcn_s_memtime();
    for (int rep = 0; rep < reps; ++rep) {
        {
            double e = 0.0;
#pragma unroll
            for (int i = 0; i < NX; ++i) {
#pragma unroll
                for (int j = 0; j <= i; ++j) {
                    const double v = rdlane(ht_entry(in, i, j), N);
                    e = (rr == NU + i && cc == NU + j) ? v : e;
                }
                const double g = rdlane(in.gt[NU + i], N);
                e = (rr == NU + i && cc == 7) ? g : e;
            }
            pnew = e;
        }
        for (int khi = N - 1; khi >= 0; khi -= CH) {
            const int klo = (khi - CH + 1 > 0) ? khi - CH + 1 : 0;
            __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
            __builtin_amdgcn_wave_barrier();
            if (lane >= klo && lane <= khi) {
                double *w = rec + (lane - klo) * REC4;
#pragma unroll
                for (int q = 0; q < 14; ++q) w[q] = in.ab[q];
                w[14] = hB; w[15] = 0.0; w[16] = 1.0;
#pragma unroll
                for (int q = 0; q < NZ; ++q) w[17 + q] = (q == 2) ? in.Ht[2] + in.hxx : (q == 3) ? in.Ht[3] + in.hyy : in.Ht[q];
                w[24] = in.hxy;
#pragma unroll
                for (int i = 0; i < NX; ++i) w[25 + i] = in.rb[i];
#pragma unroll
                for (int q = 0; q < NZ; ++q) w[30 + q] = in.gt[q];
            }
            __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
            __builtin_amdgcn_wave_barrier();
            const char *base = (const char *)rec + (size_t)(khi - klo) * REC4 * 8;
            double bc[NX], br[NX], ht;
#pragma unroll
            for (int i = 0; i < NX; ++i) { bc[i] = *(const double *)(base + oc[i]); br[i] = *(const double *)(base + orr[i]); }
            ht = *(const double *)(base + off_h);
            for (int k = khi; k >= klo; --k) {
                double nbc[NX], nbr[NX], nht;
                const char *nb = (k > klo) ? base - REC4 * 8 : base;
#pragma unroll
                for (int i = 0; i < NX; ++i) { nbc[i] = *(const double *)(nb + oc[i]); nbr[i] = *(const double *)(nb + orr[i]); }
                nht = *(const double *)(nb + off_h);
                double P[15], pn[NX];
#pragma unroll
                for (int i = 0; i < NX; ++i) {
#pragma unroll
                    for (int j = 0; j <= i; ++j) P[tri(i, j)] = rdlane(pnew, 8 * (NU + i) + (NU + j));
                    pn[i] = rdlane(pnew, 8 * (NU + i) + 7);
                }
                double pba[NX];
#pragma unroll
                for (int i = 0; i < NX; ++i) {
                    double acc = psym(P, i, 0) * bc[0];
#pragma unroll
                    for (int j = 1; j < NX; ++j) acc = fma(psym(P, i, j), bc[j], acc);
                    pba[i] = (rhs ? pn[i] : -0.0) + acc;
                }
                double m = ht;
#pragma unroll
                for (int i = 0; i < NX; ++i) m = fma(br[i], pba[i], m);
                const double m00 = rdlane(m, 0), m10 = rdlane(m, 8), m11 = rdlane(m, 9);
                const double mr0 = bperm(src_r0, m), mr1 = bperm(src_r1, m), mc0 = bperm(src_c0, m), mc1 = bperm(src_c1, m);
                __builtin_amdgcn_sched_barrier(0);
                const double d0 = 1.0 / m00;
                const double l10 = m10 * d0;
                const double d1 = 1.0 / fma(-l10, m10, m11);
                const double y1 = fma(-l10, mc0, mc1);
                const double z0 = mc0 * d0, z1 = y1 * d1;
                const double s0 = fma(-l10, z1, z0);
                const double k0j = -s0, k1j = -z1;
                const double e = fma(mr0, k0j, m);
                pnew = fma(mr1, k1j, e);
                {
                    double *ob = rec + (k - klo) * REC4;
                    const int back = (k - klo) * REC4;
                    ob[(oA >= 0) ? oA : spare - back] = diag ? k0j : kvl ? m : l10;
                    ob[(oB >= 0) ? oB : spare - back] = diag ? k1j : d0;
                    if (lane == 0) ob[12] = d1;
                }
#pragma unroll
                for (int i = 0; i < NX; ++i) { bc[i] = nbc[i]; br[i] = nbr[i]; }
                ht = nht;
                base = nb;
            }
            __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
            __builtin_amdgcn_wave_barrier();
            if (lane >= klo && lane <= khi) {
                const double *w = rec + (lane - klo) * REC4;
#pragma unroll
                for (int i = 0; i < NX; ++i) { ro.K[0][i] = w[i]; ro.K[1][i] = w[5 + i]; }
                ro.l10 = w[10]; ro.d0 = w[11]; ro.d1 = w[12]; ro.kv[0] = w[13]; ro.kv[1] = w[14];
            }
        }
        pnew = fresh(pnew);
    }
    long long t1 = __builtin_amdgcn_s_memtime();
    if (lane <= N) out[(size_t)blockIdx.x * 64 + lane] = ro;
#pragma unroll
    for (int i = 0; i < NX; ++i) {
#pragma unroll
        for (int j = 0; j <= i; ++j)
            if (lane == 8 * (NU + i) + (NU + j)) P0[(size_t)blockIdx.x * 20 + tri(i, j)] = pnew;
        if (lane == 8 * (NU + i) + 7) P0[(size_t)blockIdx.x * 20 + 15 + i] = pnew;
    }
    if (lane == 0) cycles[blockIdx.x] = t1 - t0;
}

// ---------------------------------------------------------------------------------------------------
// SPREAD5 = SPREAD4 with fewer instructions a step (inside the solver the loop is bound by instruction issue):
// no double buffer for the operands (they are read at the top of the step; the broadcast of P covers the LDS
// latency); p(k+1) reaches the right-hand-side lanes by ds_bpermute, the other lanes read -0 from lane 63;
// every lane has a fixed word of the record for each of its two result stores (word 37 for nothing); lane 1
// writes the second pivot.  VAR bit 0: P(k+1) is broadcast through LDS (one write, ten 16-byte reads) instead of
// 30 v_readlane.
// ---------------------------------------------------------------------------------------------------
#define REC5 38
typedef double __attribute__((ext_vector_type(2))) dbl2;

template <int CH, int VAR>
__global__ void __launch_bounds__(64) k_spread5(const StageIn *stages, double hB, int N, int reps, StageOut *out, double *P0,
                                                long long *cycles)
{
    __shared__ double rec[CH * REC5];
    __shared__ __attribute__((aligned(16))) double pbc[16];
    const int lane = (int)threadIdx.x;
    const int r = lane >> 3, c = lane & 7;
    const bool ent = r < NZ;
    const bool rhs = ent && c == 7;
    const int rr = ent ? r : 0, cc = ent ? c : 0;
    const StageIn in = stages[(size_t)blockIdx.x * 64 + lane];
    StageOut ro = {};
    unsigned oc[NX], orr[NX];
#pragma unroll
    for (int j = 0; j < NX; ++j) { oc[j] = 8u * (unsigned)ba_word(cc, j); orr[j] = 8u * (unsigned)ba_word(rr, j); }
    const unsigned off_h = (cc == 7) ? 8u * (unsigned)(30 + rr)
                                     : 8u * (unsigned)((rr == cc) ? 17 + rr : (rr == 3 && cc == 2) ? 24 : 15);
    const bool diag = ent && cc < 7 && rr == cc && rr >= NU;
    const bool kvl = rhs && rr < NU;
    const unsigned oA = 8u * (unsigned)(diag ? (cc - NU) : (lane == 0) ? 10 : kvl ? 13 + rr : 37);
    const unsigned oB = 8u * (unsigned)(diag ? (5 + cc - NU) : (lane == 0) ? 11 : (lane == 1) ? 12 : 37);
    const int src_r0 = 4 * (rhs ? 8 * 0 + 7 : 8 * rr + 0), src_r1 = 4 * (rhs ? 8 * 1 + 7 : 8 * rr + 1);
    const int src_c0 = 4 * (rhs ? 8 * rr + 0 : 8 * cc + 0), src_c1 = 4 * (rhs ? 8 * rr + 1 : 8 * cc + 1);
    int src_p[NX];      // where p(k+1)[i] comes from: its lane for the right-hand-side lanes, lane 63 (-0) for the others
#pragma unroll
    for (int i = 0; i < NX; ++i) src_p[i] = 4 * (rhs ? 8 * (NU + i) + 7 : 63);
    int pslot = -1;
#pragma unroll
    for (int i = 0; i < NX; ++i)
#pragma unroll
        for (int j = 0; j <= i; ++j)
            if (lane == 8 * (NU + i) + (NU + j)) pslot = tri(i, j);
    double pnew = 0.0;
    long long t0 = __builtin_amdgcn_s_memtime();
    for (int rep = 0; rep < reps; ++rep) {
        {
            double e = 0.0;
#pragma unroll
            for (int i = 0; i < NX; ++i) {
#pragma unroll
                for (int j = 0; j <= i; ++j) {
                    const double v = rdlane(ht_entry(in, i, j), N);
                    e = (rr == NU + i && cc == NU + j) ? v : e;
                }
                const double g = rdlane(in.gt[NU + i], N);
                e = (rr == NU + i && cc == 7) ? g : e;
            }
            pnew = (lane == 63) ? -0.0 : e;
        }
        for (int khi = N - 1; khi >= 0; khi -= CH) {
            const int klo = (khi - CH + 1 > 0) ? khi - CH + 1 : 0;
            __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
            __builtin_amdgcn_wave_barrier();
            if (lane >= klo && lane <= khi) {
                double *w = rec + (lane - klo) * REC5;
#pragma unroll
                for (int q = 0; q < 14; ++q) w[q] = in.ab[q];
                w[14] = hB; w[15] = 0.0; w[16] = 1.0;
#pragma unroll
                for (int q = 0; q < NZ; ++q) w[17 + q] = (q == 2) ? in.Ht[2] + in.hxx : (q == 3) ? in.Ht[3] + in.hyy : in.Ht[q];
                w[24] = in.hxy;
#pragma unroll
                for (int i = 0; i < NX; ++i) w[25 + i] = in.rb[i];
#pragma unroll
                for (int q = 0; q < NZ; ++q) w[30 + q] = in.gt[q];
            }
            __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
            __builtin_amdgcn_wave_barrier();
            const char *base = (const char *)rec + (size_t)(khi - klo) * REC5 * 8;
            for (int k = khi; k >= klo; --k) {
                if (VAR & 1) {
                    if (pslot >= 0) pbc[pslot] = pnew;
                }
                double bc[NX], br[NX], ht;
#pragma unroll
                for (int i = 0; i < NX; ++i) { bc[i] = *(const double *)(base + oc[i]); br[i] = *(const double *)(base + orr[i]); }
                ht = *(const double *)(base + off_h);
                double pn[NX];
#pragma unroll
                for (int i = 0; i < NX; ++i) pn[i] = bperm(src_p[i], pnew);
                double P[16];
                if (VAR & 1) {
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                    __builtin_amdgcn_wave_barrier();
#pragma unroll
                    for (int i = 0; i < 8; ++i) {
                        const dbl2 v = *(const dbl2 *)&pbc[2 * i];
                        P[2 * i] = v.x; P[2 * i + 1] = v.y;
                    }
                } else {
#pragma unroll
                    for (int i = 0; i < NX; ++i)
#pragma unroll
                        for (int j = 0; j <= i; ++j) P[tri(i, j)] = rdlane(pnew, 8 * (NU + i) + (NU + j));
                }
                double pba[NX];
#pragma unroll
                for (int i = 0; i < NX; ++i) {
                    double acc = psym(P, i, 0) * bc[0];
#pragma unroll
                    for (int j = 1; j < NX; ++j) acc = fma(psym(P, i, j), bc[j], acc);
                    pba[i] = pn[i] + acc;
                }
                double m = ht;
#pragma unroll
                for (int i = 0; i < NX; ++i) m = fma(br[i], pba[i], m);
                const double m00 = rdlane(m, 0), m10 = rdlane(m, 8), m11 = rdlane(m, 9);
                const double mr0 = bperm(src_r0, m), mr1 = bperm(src_r1, m), mc0 = bperm(src_c0, m), mc1 = bperm(src_c1, m);
                __builtin_amdgcn_sched_barrier(0);
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
                *(double *)((char *)base + oA) = diag ? k0j : kvl ? m : l10;
                *(double *)((char *)base + oB) = diag ? k1j : (lane == 1) ? d1 : d0;
                base -= REC5 * 8;
            }
            __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
            __builtin_amdgcn_wave_barrier();
            if (lane >= klo && lane <= khi) {
                const double *w = rec + (lane - klo) * REC5;
#pragma unroll
                for (int i = 0; i < NX; ++i) { ro.K[0][i] = w[i]; ro.K[1][i] = w[5 + i]; }
                ro.l10 = w[10]; ro.d0 = w[11]; ro.d1 = w[12]; ro.kv[0] = w[13]; ro.kv[1] = w[14];
            }
        }
        pnew = fresh(pnew);
    }
    long long t1 = __builtin_amdgcn_s_memtime();
    if (lane <= N) out[(size_t)blockIdx.x * 64 + lane] = ro;
#pragma unroll
    for (int i = 0; i < NX; ++i) {
#pragma unroll
        for (int j = 0; j <= i; ++j)
            if (lane == 8 * (NU + i) + (NU + j)) P0[(size_t)blockIdx.x * 20 + tri(i, j)] = pnew;
        if (lane == 8 * (NU + i) + 7) P0[(size_t)blockIdx.x * 20 + 15 + i] = pnew;
    }
    if (lane == 0) cycles[blockIdx.x] = t1 - t0;
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
    std::vector<StageIn> st((size_t)WG * 64);
    unsigned long long seed = 12345;
    for (auto &s : st) {
        const double v = 2.0 * urand(seed) - 0.5, th = 6.0 * urand(seed) - 3.0;
        s.ab[0] = -h * v * sin(th); s.ab[1] = h * cos(th); s.ab[2] = 1e-3 * (urand(seed) - 0.5);
        s.ab[3] = h * v * cos(th);  s.ab[4] = h * sin(th); s.ab[5] = 1e-3 * (urand(seed) - 0.5);
        s.ab[6] = h * 0.1 * (urand(seed) - 0.5); s.ab[7] = h * v / 2.7 * (1.0 + urand(seed));
        s.ab[8] = 0.005 * cos(th); s.ab[9] = 0.005 * sin(th); s.ab[10] = 1e-4 * (urand(seed) - 0.5);
        s.ab[11] = 1e-5 * (urand(seed) - 0.5); s.ab[12] = 1e-5 * (urand(seed) - 0.5); s.ab[13] = 0.005 * v / 2.7;
        for (int j = 0; j < NZ; ++j) s.Ht[j] = (j < NU ? 10.0 : 1e-3) + pow(10.0, 12.0 * urand(seed) - 6.0);
        const double w = pow(10.0, 8.0 * urand(seed) - 4.0), jx = urand(seed) - 0.5, jy = urand(seed) - 0.5;
        s.hxx = w * jx * jx; s.hxy = w * jy * jx; s.hyy = w * jy * jy;
        for (int j = 0; j < NZ; ++j) s.gt[j] = pow(10.0, 6.0 * urand(seed) - 3.0) * (urand(seed) - 0.5);
        for (int j = 0; j < NX; ++j) s.rb[j] = 1e-2 * (urand(seed) - 0.5);
    }
    StageIn *d_st; StageOut *d_o1, *d_o2; double *d_p1, *d_p2; long long *d_cy;
    hipMalloc(&d_st, st.size() * sizeof(StageIn));
    hipMalloc(&d_o1, st.size() * sizeof(StageOut)); hipMalloc(&d_o2, st.size() * sizeof(StageOut));
    hipMalloc(&d_p1, (size_t)WG * 20 * 8); hipMalloc(&d_p2, (size_t)WG * 20 * 8);
    hipMalloc(&d_cy, (size_t)WG * 8);
    hipMemcpy(d_st, st.data(), st.size() * sizeof(StageIn), hipMemcpyHostToDevice);
    std::vector<StageOut> o1(st.size()), o2(st.size());
    std::vector<double> p1((size_t)WG * 20), p2((size_t)WG * 20);
    std::vector<long long> cy(WG);
    auto report = [&](const char *name, float ms) {
        hipMemcpy(cy.data(), d_cy, (size_t)WG * 8, hipMemcpyDeviceToHost);
        double sum = 0.0;
        for (long long v : cy) sum += (double)v;
        printf("%-28s %8.1f cycles per stage (mean over %d wavefronts), %7.3f ms for %d recursions of %d stages each\n", name,
               sum / WG / ((double)reps * N), WG, ms, reps, N);
    };
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    float ms = 0.f;
    for (int pass = 0; pass < 2; ++pass) {
        hipMemset(d_o1, 0, st.size() * sizeof(StageOut));
        hipEventRecord(e0);
        hipLaunchKernelGGL(k_pipe, dim3(WG), dim3(64), 0, 0, d_st, hB, N, reps, d_o1, d_p1, d_cy);
        hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
        if (pass) report("PIPE (two lane pipelines)", ms);
    }
    hipMemcpy(o1.data(), d_o1, st.size() * sizeof(StageOut), hipMemcpyDeviceToHost);
    hipMemcpy(p1.data(), d_p1, p1.size() * 8, hipMemcpyDeviceToHost);
    auto check = [&](const char *name) {
        hipMemcpy(o2.data(), d_o2, st.size() * sizeof(StageOut), hipMemcpyDeviceToHost);
        hipMemcpy(p2.data(), d_p2, p2.size() * 8, hipMemcpyDeviceToHost);
        size_t bad = 0;
        for (int w = 0; w < WG; ++w) {
            for (int k = 0; k < N; ++k)
                if (memcmp(&o1[(size_t)w * 64 + k], &o2[(size_t)w * 64 + k], sizeof(StageOut)) != 0) ++bad;
            if (memcmp(&p1[(size_t)w * 20], &p2[(size_t)w * 20], 20 * 8) != 0) ++bad;
        }
        printf("%-28s factors, right-hand sides, P(0), p(0) of %d problems %s the lane pipelines' (%zu records differ)\n", name, WG,
               bad ? "DIFFER from" : "equal, bit for bit,", bad);
        return bad == 0;
    };
    bool ok = true;
#define RUN_SPREAD(CH_, VAR_)                                                                                      \
    for (int pass = 0; pass < 2; ++pass) {                                                                   \
        hipMemset(d_o2, 0, st.size() * sizeof(StageOut));                                                    \
        hipEventRecord(e0);                                                                                  \
        hipLaunchKernelGGL((k_spread3<CH_, VAR_>), dim3(WG), dim3(64), 0, 0, d_st, hB, N, reps, d_o2, d_p2, d_cy);   \
        hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);                       \
        if (pass) { report("SPREAD3 var " #VAR_ ", " #CH_ " staged", ms); ok = check("SPREAD3 var " #VAR_ ", " #CH_ " staged") && ok; } \
    }
    RUN_SPREAD(64, 0)
    RUN_SPREAD(64, 1)
    RUN_SPREAD(10, 1)
#define RUN_SPREAD4(CH_)                                                                                     \
    for (int pass = 0; pass < 2; ++pass) {                                                                   \
        hipMemset(d_o2, 0, st.size() * sizeof(StageOut));                                                    \
        hipEventRecord(e0);                                                                                  \
        hipLaunchKernelGGL(k_spread4<CH_>, dim3(WG), dim3(64), 0, 0, d_st, hB, N, reps, d_o2, d_p2, d_cy);   \
        hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);                       \
        if (pass) { report("SPREAD4, " #CH_ " staged", ms); ok = check("SPREAD4, " #CH_ " staged") && ok; } \
    }
    RUN_SPREAD4(64)
    RUN_SPREAD4(25)
#define RUN_SPREAD5(CH_, VAR_)                                                                               \
    for (int pass = 0; pass < 2; ++pass) {                                                                   \
        hipMemset(d_o2, 0, st.size() * sizeof(StageOut));                                                    \
        hipEventRecord(e0);                                                                                  \
        hipLaunchKernelGGL((k_spread5<CH_, VAR_>), dim3(WG), dim3(64), 0, 0, d_st, hB, N, reps, d_o2, d_p2, d_cy); \
        hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);                       \
        if (pass) { report("SPREAD5 var " #VAR_ ", " #CH_ " staged", ms); ok = check("SPREAD5 var " #VAR_ ", " #CH_ " staged") && ok; } \
    }
    RUN_SPREAD5(64, 0)
    RUN_SPREAD5(25, 0)
    RUN_SPREAD5(64, 1)
    RUN_SPREAD5(25, 1)
    if (hipDeviceSynchronize() != hipSuccess) { printf("HIP error\n"); return 2; }
    return ok ? 0 : 1;
}
