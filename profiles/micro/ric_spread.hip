// The backward Riccati factorisation of ONE problem (nx = 5, nu = 2, N stages) by one wavefront, three ways
// (SPREAD2, which keeps P(k+1) out of the scalar registers and pays three LDS-crossbar round trips a step
// instead, is described at its kernel):
//
//   PIPE    the lane pipeline of the shipped single-problem mode: lane k owns stage k, every step every
//           lane recomputes its stage from what the lane above produced the step before (DPP wave_shl:1),
//           N steps, one useful lane per step (~300 FP64 instructions + 30 moves a step);
//   SPREAD  a serial loop over the stages with the 7 x 7 algebra of the current stage spread over the
//           lanes: lane 8 r + c owns entry (r, c) of M = Ht + [B A]' P [B A]; P(k+1) travels as 15
//           wave-uniform values (v_readlane into scalar registers), every lane forms column c of
//           P [B A] (25 fused multiply-adds), its entry of M (5), the LDL' of M_uu redundantly, the
//           feedback column it needs and its entry of P(k); the operands of the stage come from an LDS
//           staging area that the stage's lane filled, CH stages at a time.
//
// Both perform, for every number, the operations of the CPU restatement's riccati_factor() in its order
// (dense chains; multiplying by a structural 0 or 1 leaves a finite chain unchanged), so the factors must
// agree bit for bit: the program checks that, then times both with every SIMD of the GPU busy (4
// one-wavefront workgroups per CU, as the solver runs).
//
// Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fno-fast-math ric_spread.hip -o ric_spread
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#define NX 5
#define NU 2
#define NZ 7
#define DEV __device__ __forceinline__

struct StageIn {          // what the row part of sweep B0 leaves for the factorisation
    double ab[14];        // a02 a03 a04 a12 a13 a14 a23 a24 b00 b10 b20 b01 b11 b21
    double Ht[NZ], hxx, hxy, hyy;
};
struct StageOut { double K[2][NX], l10, d0, d1; };

DEV constexpr int tri(int i, int j) { return i * (i + 1) / 2 + j; }
DEV double psym(const double *P, int i, int j) { return (i >= j) ? P[tri(i, j)] : P[tri(j, i)]; }

DEV double lane_above0(double v)      // value of lane + 1, zero at the end of the wavefront
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
DEV double fresh(double x) { asm volatile("" : "+v"(x)); return x; }

// one stage of the factorisation on the sparsity pattern of A and B (what the shipped kernel executes)
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

DEV double ht_entry(const StageIn &in, int i, int j)      // entry (NU+i, NU+j), i >= j, of the barrier Hessian
{
    if (i == j) return (i == 0) ? in.Ht[2] + in.hxx : (i == 1) ? in.Ht[3] + in.hyy : in.Ht[NU + i];
    return (i == 1 && j == 0) ? in.hxy : 0.0;
}

// ---------------------------------------------------------------------------------------------------
// PIPE
// ---------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(64) k_pipe(const StageIn *stages, double hB, int N, int reps, StageOut *out, double *P0,
                                             long long *cycles)
{
    const int kj = (int)threadIdx.x;
    const StageIn *mine = stages + (size_t)blockIdx.x * 64;
    StageIn in = mine[kj];
    StageOut ro = {};
    double Po[15];
    long long t0 = __builtin_amdgcn_s_memtime();
    for (int rep = 0; rep < reps; ++rep) {
#pragma unroll
        for (int i = 0; i < NX; ++i)
#pragma unroll
            for (int j = 0; j <= i; ++j) Po[tri(i, j)] = (kj == N) ? ht_entry(in, i, j) : 0.0;
#pragma unroll
        for (int q = 0; q < 14; ++q) in.ab[q] = fresh(in.ab[q]);
        for (int s = 0; s < N; ++s) {
            double Pi[15];
#pragma unroll
            for (int i = 0; i < 15; ++i) Pi[i] = lane_above0(Po[i]);
            if (kj < N) ric_stage(in, hB, Pi, Po, ro);
        }
        in.Ht[0] = fresh(in.Ht[0]);      // (keeps the repetitions from being merged)
    }
    long long t1 = __builtin_amdgcn_s_memtime();
    if (kj <= N) out[(size_t)blockIdx.x * 64 + kj] = ro;
    if (kj == 0) {
        for (int i = 0; i < 15; ++i) P0[(size_t)blockIdx.x * 15 + i] = Po[i];
        cycles[blockIdx.x] = t1 - t0;
    }
}

// ---------------------------------------------------------------------------------------------------
// SPREAD.  Staging record of a stage (doubles): [0,35) [B A] dense by column, column c at 5 c;
// [35,42) the seven diagonal entries of the barrier Hessian (the (x, y) block's already added);
// 42 its one off-diagonal entry hxy; 43 zero.  When the stage has been factorised its record receives the
// results: [0,5) K row 0, [5,10) K row 1, 10 l10, 11 d0, 12 d1 (read back by the stage's lane).
// ---------------------------------------------------------------------------------------------------
#define REC 44

template <int CH>
__global__ void __launch_bounds__(64) k_spread(const StageIn *stages, double hB, int N, int reps, StageOut *out, double *P0,
                                               long long *cycles)
{
    __shared__ double rec[CH * REC + 8];
    const int lane = (int)threadIdx.x;
    const int r = lane >> 3, c = lane & 7;
    const bool ent = (r < NZ) && (c < NZ);                 // the lane owns an entry of M
    const int rr = ent ? r : 0, cc = ent ? c : 0;          // (lanes without an entry run along on entry (0,0))
    const StageIn *mine = stages + (size_t)blockIdx.x * 64;
    const StageIn in = mine[lane];                          // the lane's OWN stage (lane = stage), as the row code leaves it
    StageOut ro = {};
    // byte offsets inside a record of what this lane reads each step
    const unsigned off_c = 8u * 5u * (unsigned)cc, off_r = 8u * 5u * (unsigned)rr;
    const unsigned off_h = 8u * (unsigned)((rr == cc) ? 35 + rr : (rr == 3 && cc == 2) ? 42 : 43);
    // where this lane's results of a step go in the record (a spare word for lanes that have none)
    //   diagonal lanes (NU+j, NU+j): K[0][j] -> j, K[1][j] -> 5 + j;  lane (0,0): l10, d0, d1 -> 10, 11, 12
    const bool diag = ent && rr == cc && rr >= NU;
    const unsigned spare = 8u * (unsigned)(CH * REC);
    // sources of the cross-lane reads: M[r][0], M[r][1] of the lane's row; M[c][0], M[c][1] of its column
    const int src_r0 = 4 * (8 * rr + 0), src_r1 = 4 * (8 * rr + 1), src_c0 = 4 * (8 * cc + 0), src_c1 = 4 * (8 * cc + 1);
    double pnew = 0.0;
    long long t0 = __builtin_amdgcn_s_memtime();
    for (int rep = 0; rep < reps; ++rep) {
        // P(N) = the (x-part of the) barrier Hessian of the last stage: entry (NU+i, NU+j) in lane (NU+i, NU+j)
        {
            const StageIn &sN = in;      // valid in lane N only; broadcast entry by entry
            double e = 0.0;
#pragma unroll
            for (int i = 0; i < NX; ++i)
#pragma unroll
                for (int j = 0; j <= i; ++j) {
                    const double v = rdlane(ht_entry(sN, i, j), N);
                    e = (rr == NU + i && cc == NU + j) ? v : e;
                }
            pnew = e;
        }
        for (int khi = N - 1; khi >= 0; khi -= CH) {
            const int klo = (khi - CH + 1 > 0) ? khi - CH + 1 : 0;
            // ---- staging: the lanes of stages klo..khi write their records ----
            __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
            __builtin_amdgcn_wave_barrier();
            if (lane >= klo && lane <= khi) {
                double *w = rec + (lane - klo) * REC;
                const double *ab = in.ab;
                // column 0, 1 = B; 2..6 = A
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
            }
            __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
            __builtin_amdgcn_wave_barrier();
            // ---- the chain over the staged stages ----
            const char *base = (const char *)rec + (size_t)(khi - klo) * REC * 8;
            double bc[NX], br[NX], ht;
#pragma unroll
            for (int i = 0; i < NX; ++i) { bc[i] = *(const double *)(base + off_c + 8 * i); br[i] = *(const double *)(base + off_r + 8 * i); }
            ht = *(const double *)(base + off_h);
            for (int k = khi; k >= klo; --k) {
                // operands of the NEXT step are fetched while this one computes
                double nbc[NX], nbr[NX], nht;
                const char *nb = (k > klo) ? base - REC * 8 : base;
#pragma unroll
                for (int i = 0; i < NX; ++i) { nbc[i] = *(const double *)(nb + off_c + 8 * i); nbr[i] = *(const double *)(nb + off_r + 8 * i); }
                nht = *(const double *)(nb + off_h);
                // P(k+1), wave-uniform (scalar registers)
                double P[15];
#pragma unroll
                for (int i = 0; i < NX; ++i)
#pragma unroll
                    for (int j = 0; j <= i; ++j) P[tri(i, j)] = rdlane(pnew, 8 * (NU + i) + (NU + j));
                // column c of P [B A], then entry (r, c) of M
                double pba[NX];
#pragma unroll
                for (int i = 0; i < NX; ++i) {
                    double acc = psym(P, i, 0) * bc[0];
#pragma unroll
                    for (int j = 1; j < NX; ++j) acc = fma(psym(P, i, j), bc[j], acc);
                    pba[i] = acc;
                }
                double m = ht;
#pragma unroll
                for (int i = 0; i < NX; ++i) m = fma(br[i], pba[i], m);
                // LDL' of M_uu, in every lane
                const double m00 = rdlane(m, 0), m10 = rdlane(m, 8), m11 = rdlane(m, 9);
                const double d0 = 1.0 / m00;
                const double l10 = m10 * d0;
                const double d1 = 1.0 / fma(-l10, m10, m11);
                // M[r][0], M[r][1] of the lane's row and M[c][0], M[c][1] of its column
                const int mlo = __double2loint(m), mhi = __double2hiint(m);
                const double mr0 = __hiloint2double(__builtin_amdgcn_ds_bpermute(src_r0, mhi), __builtin_amdgcn_ds_bpermute(src_r0, mlo));
                const double mr1 = __hiloint2double(__builtin_amdgcn_ds_bpermute(src_r1, mhi), __builtin_amdgcn_ds_bpermute(src_r1, mlo));
                const double mc0 = __hiloint2double(__builtin_amdgcn_ds_bpermute(src_c0, mhi), __builtin_amdgcn_ds_bpermute(src_c0, mlo));
                const double mc1 = __hiloint2double(__builtin_amdgcn_ds_bpermute(src_c1, mhi), __builtin_amdgcn_ds_bpermute(src_c1, mlo));
                // feedback column j = c - NU
                const double y1 = fma(-l10, mc0, mc1);
                const double z0 = mc0 * d0, z1 = y1 * d1;
                const double s0 = fma(-l10, z1, z0);
                const double k0j = -s0, k1j = -z1;
                // entry (r - NU, c - NU) of P(k)
                const double e = fma(mr0, k0j, m);
                pnew = fma(mr1, k1j, e);
                // results of stage k into its record
                {
                    char *ob = (char *)rec + (size_t)(k - klo) * REC * 8;
                    const unsigned o0 = diag ? 8u * (unsigned)(cc - NU) : (lane == 0) ? 8u * 10u : spare - (unsigned)((k - klo) * REC * 8);
                    const unsigned o1 = diag ? 8u * (unsigned)(5 + cc - NU) : (lane == 0) ? 8u * 11u : spare - (unsigned)((k - klo) * REC * 8);
                    *(double *)(ob + o0) = diag ? k0j : l10;
                    *(double *)(ob + o1) = diag ? k1j : d0;
                    if (lane == 0) *(double *)(ob + 8 * 12) = d1;
                }
#pragma unroll
                for (int i = 0; i < NX; ++i) { bc[i] = nbc[i]; br[i] = nbr[i]; }
                ht = nht;
                base = nb;
            }
            // ---- the stages of the chunk take their factors back ----
            __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
            __builtin_amdgcn_wave_barrier();
            if (lane >= klo && lane <= khi) {
                const double *w = rec + (lane - klo) * REC;
#pragma unroll
                for (int i = 0; i < NX; ++i) { ro.K[0][i] = w[i]; ro.K[1][i] = w[5 + i]; }
                ro.l10 = w[10]; ro.d0 = w[11]; ro.d1 = w[12];
            }
        }
        pnew = fresh(pnew);
    }
    long long t1 = __builtin_amdgcn_s_memtime();
    if (lane <= N) out[(size_t)blockIdx.x * 64 + lane] = ro;
#pragma unroll
    for (int i = 0; i < NX; ++i)
#pragma unroll
        for (int j = 0; j <= i; ++j)
            if (lane == 8 * (NU + i) + (NU + j)) P0[(size_t)blockIdx.x * 15 + tri(i, j)] = pnew;
    if (lane == 0) cycles[blockIdx.x] = t1 - t0;
}


// ---------------------------------------------------------------------------------------------------
// SPREAD2.  Lane 8 g + c: group g = 0..4 owns row NU+g of M (an x row), groups 5, 6 own rows 0, 1 (the u
// rows); c is the column of M.  Per step:
//   A  every lane of group g < 5 takes row g of P(k+1) from the lanes (g, NU+j) of its own group
//      (ds_swizzle, pattern fixed at compile time) and forms entry (g, c) of P [B A] (5 multiply-adds);
//      the same row gives entry g of P(k+1) rb;
//   B  column c of P [B A] is gathered from the lanes (0..4, c) (ds_bpermute) and the lane forms its entry
//      of M (5 multiply-adds);
//   C  the three entries of M_uu go to scalar registers, the LDL' is formed in every lane; meanwhile the
//      lane gathers M[r][0..1] of its row (swizzle), of the row that belongs to its column, and the entry of
//      M mirrored at the diagonal (bpermute);
//   D  the lanes (g, NU+j), g, j < 5, form entry (g, j) of P(k): below and on the diagonal from their own
//      entry of M, above it as the mirror entry (j, g) -- the same operations on the same operands the
//      lane (j, g) performs, so row g of P(k) is complete in group g for the next step.
// Staging record (doubles): [0,35) [B A] by column; [35,42) diagonal of the barrier Hessian; 42 hxy; 43 zero;
// [44,49) rb.  Results of a stage into its record: [0,5) K row 0, [5,10) K row 1, 10 l10, 11 d0, 12 d1,
// [13,18) P(k+1) rb.
// ---------------------------------------------------------------------------------------------------
#define REC2 50

template <int PAT>
DEV double swz(double v)
{
    const int lo = __builtin_amdgcn_ds_swizzle(__double2loint(v), PAT);
    const int hi = __builtin_amdgcn_ds_swizzle(__double2hiint(v), PAT);
    return __hiloint2double(hi, lo);
}
DEV double bperm(int addr, double v)
{
    const int lo = __builtin_amdgcn_ds_bpermute(addr, __double2loint(v));
    const int hi = __builtin_amdgcn_ds_bpermute(addr, __double2hiint(v));
    return __hiloint2double(hi, lo);
}
// BitMode swizzle: lane' = ((lane & and) | or) ^ xor inside each half of the wavefront; and = 0x18 keeps the
// group, or = the column to read from
#define SWZ_GROUP_COL(col) ((0x18) | ((col) << 5))

template <int CH>
__global__ void __launch_bounds__(64) k_spread2(const StageIn *stages, const double *rbs, double hB, int N, int reps,
                                                StageOut *out, double *prb_out, double *P0, long long *cycles)
{
    __shared__ double rec[CH * REC2 + 8];
    const int lane = (int)threadIdx.x;
    const int g = lane >> 3, c = lane & 7;
    const int r = (g < NX) ? NU + g : (g - NX);            // row of M this group owns (groups 5, 6: rows 0, 1; group 7: none)
    const bool ent = (g < NZ) && (c < NZ);
    const int rr = ent ? r : 0, cc = ent ? c : 0, gg = ent ? g : 5;
    const StageIn *mine = stages + (size_t)blockIdx.x * 64;
    const StageIn in = mine[lane];
    double rb[NX];
#pragma unroll
    for (int i = 0; i < NX; ++i) rb[i] = rbs[((size_t)blockIdx.x * 64 + lane) * NX + i];
    StageOut ro = {};
    double prb[NX] = {0.0, 0.0, 0.0, 0.0, 0.0};
    const unsigned off_c = 8u * 5u * (unsigned)cc, off_r = 8u * 5u * (unsigned)rr;
    const unsigned off_h = 8u * (unsigned)((rr == cc) ? 35 + rr : (rr == 3 && cc == 2) ? 42 : 43);
    const bool pl = ent && gg < NX && cc >= NU;             // the lane holds an entry of P
    const int j = pl ? cc - NU : 0;                         // ... entry (gg, j)
    const bool lower = gg >= j;
    // cross-lane sources
    const int a_col = 4 * cc;                                // column gather: lanes (i, c), + 32 i
    const int a_row0 = 4 * (8 * (pl ? j : 0) + 0);           // M[NU+j][0], M[NU+j][1]: lanes (j, 0), (j, 1)
    const int a_mirror = 4 * (8 * (pl ? j : 0) + NU + (pl ? gg : 0));      // M[NU+j][NU+g]: lane (j, NU+g)
    const unsigned spare = 8u * (unsigned)(CH * REC2);
    double pnew = 0.0;
    long long t0 = __builtin_amdgcn_s_memtime();
    for (int rep = 0; rep < reps; ++rep) {
        {
            double e = 0.0;
#pragma unroll
            for (int i = 0; i < NX; ++i)
#pragma unroll
                for (int jj = 0; jj < NX; ++jj) {
                    const double v = rdlane(ht_entry(in, (i >= jj) ? i : jj, (i >= jj) ? jj : i), N);
                    e = (pl && gg == i && j == jj) ? v : e;
                }
            pnew = e;
        }
        for (int khi = N - 1; khi >= 0; khi -= CH) {
            const int klo = (khi - CH + 1 > 0) ? khi - CH + 1 : 0;
            __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
            __builtin_amdgcn_wave_barrier();
            if (lane >= klo && lane <= khi) {
                double *w = rec + (lane - klo) * REC2;
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
                for (int i = 0; i < NX; ++i) w[44 + i] = rb[i];
            }
            __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
            __builtin_amdgcn_wave_barrier();
            const char *base = (const char *)rec + (size_t)(khi - klo) * REC2 * 8;
            double bc[NX], br[NX], ht, rbk[NX];
#pragma unroll
            for (int i = 0; i < NX; ++i) {
                bc[i] = *(const double *)(base + off_c + 8 * i); br[i] = *(const double *)(base + off_r + 8 * i);
                rbk[i] = *(const double *)(base + 8 * (44 + i));
            }
            ht = *(const double *)(base + off_h);
            for (int k = khi; k >= klo; --k) {
                double nbc[NX], nbr[NX], nht, nrb[NX];
                const char *nb = (k > klo) ? base - REC2 * 8 : base;
#pragma unroll
                for (int i = 0; i < NX; ++i) {
                    nbc[i] = *(const double *)(nb + off_c + 8 * i); nbr[i] = *(const double *)(nb + off_r + 8 * i);
                    nrb[i] = *(const double *)(nb + 8 * (44 + i));
                }
                nht = *(const double *)(nb + off_h);
                // A: row g of P(k+1), from the lanes (g, NU + jj) of the lane's own group
                const double p0 = swz<SWZ_GROUP_COL(2)>(pnew), p1 = swz<SWZ_GROUP_COL(3)>(pnew), p2 = swz<SWZ_GROUP_COL(4)>(pnew),
                             p3 = swz<SWZ_GROUP_COL(5)>(pnew), p4 = swz<SWZ_GROUP_COL(6)>(pnew);
                double pba = p0 * bc[0];
                pba = fma(p1, bc[1], pba); pba = fma(p2, bc[2], pba); pba = fma(p3, bc[3], pba); pba = fma(p4, bc[4], pba);
                double pr = p0 * rbk[0];
                pr = fma(p1, rbk[1], pr); pr = fma(p2, rbk[2], pr); pr = fma(p3, rbk[3], pr); pr = fma(p4, rbk[4], pr);
                // B: column c of P [B A] from the lanes (0..4, c), then the lane's entry of M
                const int plo = __double2loint(pba), phi = __double2hiint(pba);
                double m = ht;
#define COLTERM(i_)                                                                                                    \
                m = fma(br[i_], __hiloint2double(__builtin_amdgcn_ds_bpermute(a_col + 32 * (i_), phi),                  \
                                                 __builtin_amdgcn_ds_bpermute(a_col + 32 * (i_), plo)), m);
                COLTERM(0) COLTERM(1) COLTERM(2) COLTERM(3) COLTERM(4)
#undef COLTERM
                // C: LDL' of M_uu in every lane; the exchanges of phase D meanwhile
                const double m00 = rdlane(m, 8 * 5 + 0), m10 = rdlane(m, 8 * 6 + 0), m11 = rdlane(m, 8 * 6 + 1);
                const double mr0 = swz<SWZ_GROUP_COL(0)>(m), mr1 = swz<SWZ_GROUP_COL(1)>(m);      // M[NU+g][0..1]
                const double mc0 = bperm(a_row0, m), mc1 = bperm(a_row0 + 4, m);                   // M[NU+j][0..1]
                const double mT = bperm(a_mirror, m);                                              // M[NU+j][NU+g]
                const double d0 = 1.0 / m00;
                const double l10 = m10 * d0;
                const double d1 = 1.0 / fma(-l10, m10, m11);
                // D: entry (g, j) of P(k): feedback column of the smaller index, row of the larger
                const double y0 = lower ? mc0 : mr0, y1r = lower ? mc1 : mr1;
                const double x0 = lower ? mr0 : mc0, x1 = lower ? mr1 : mc1, mm = lower ? m : mT;
                const double y1 = fma(-l10, y0, y1r);
                const double z0 = y0 * d0, z1 = y1 * d1;
                const double s0 = fma(-l10, z1, z0);
                const double k0 = -s0, k1 = -z1;
                const double e = fma(x0, k0, mm);
                pnew = fma(x1, k1, e);
                // results of stage k into its record: K column j from the diagonal lanes, the pivots from lane 0,
                // P(k+1) rb from the lanes (i, 0)
                {
                    char *ob = (char *)rec + (size_t)(k - klo) * REC2 * 8;
                    const unsigned back = (unsigned)((k - klo) * REC2 * 8);
                    const bool dg = pl && gg == j;
                    const unsigned o0 = dg ? 8u * (unsigned)j : (lane == 0) ? 8u * 10u : spare - back;
                    const unsigned o1 = dg ? 8u * (unsigned)(5 + j) : (lane == 0) ? 8u * 11u : spare - back;
                    const unsigned o2 = (ent && gg < NX && cc == 0) ? 8u * (unsigned)(13 + gg) : (lane == 1) ? 8u * 12u : spare - back;
                    *(double *)(ob + o0) = dg ? k0 : l10;
                    *(double *)(ob + o1) = dg ? k1 : d0;
                    *(double *)(ob + o2) = (lane == 1) ? d1 : pr;
                }
#pragma unroll
                for (int i = 0; i < NX; ++i) { bc[i] = nbc[i]; br[i] = nbr[i]; rbk[i] = nrb[i]; }
                ht = nht;
                base = nb;
            }
            __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
            __builtin_amdgcn_wave_barrier();
            if (lane >= klo && lane <= khi) {
                const double *w = rec + (lane - klo) * REC2;
#pragma unroll
                for (int i = 0; i < NX; ++i) { ro.K[0][i] = w[i]; ro.K[1][i] = w[5 + i]; prb[i] = w[13 + i]; }
                ro.l10 = w[10]; ro.d0 = w[11]; ro.d1 = w[12];
            }
        }
        pnew = fresh(pnew);
    }
    long long t1 = __builtin_amdgcn_s_memtime();
    if (lane <= N) {
        out[(size_t)blockIdx.x * 64 + lane] = ro;
#pragma unroll
        for (int i = 0; i < NX; ++i) prb_out[((size_t)blockIdx.x * 64 + lane) * NX + i] = prb[i];
    }
#pragma unroll
    for (int i = 0; i < NX; ++i)
#pragma unroll
        for (int jj = 0; jj <= i; ++jj)
            if (lane == 8 * i + (NU + jj)) P0[(size_t)blockIdx.x * 15 + tri(i, jj)] = pnew;
    if (lane == 0) cycles[blockIdx.x] = t1 - t0;
}

// P(k+1) rb of every stage by the lane pipeline's formula (reference for SPREAD2's product)
__global__ void __launch_bounds__(64) k_pipe_prb(const StageIn *stages, const double *rbs, double hB, int N, double *prb_out)
{
    const int kj = (int)threadIdx.x;
    const StageIn in = stages[(size_t)blockIdx.x * 64 + kj];
    StageOut ro = {};
    double Po[15];
#pragma unroll
    for (int i = 0; i < NX; ++i)
#pragma unroll
        for (int j = 0; j <= i; ++j) Po[tri(i, j)] = (kj == N) ? ht_entry(in, i, j) : 0.0;
    for (int s = 0; s < N; ++s) {
        double Pi[15];
#pragma unroll
        for (int i = 0; i < 15; ++i) Pi[i] = lane_above0(Po[i]);
        if (kj < N) ric_stage(in, hB, Pi, Po, ro);
    }
    double Pi[15];
#pragma unroll
    for (int i = 0; i < 15; ++i) Pi[i] = lane_above0(Po[i]);
#pragma unroll
    for (int i = 0; i < NX; ++i) {
        double acc = psym(Pi, i, 0) * rbs[((size_t)blockIdx.x * 64 + kj) * NX + 0];
#pragma unroll
        for (int j = 1; j < NX; ++j) acc = fma(psym(Pi, i, j), rbs[((size_t)blockIdx.x * 64 + kj) * NX + j], acc);
        if (kj < N) prb_out[((size_t)blockIdx.x * 64 + kj) * NX + i] = acc;
    }
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
        // sensitivities of an RK4 step of the bicycle at a random state, roughly; barrier Hessians over many decades
        const double v = 2.0 * urand(seed) - 0.5, th = 6.0 * urand(seed) - 3.0;
        s.ab[0] = -h * v * sin(th); s.ab[1] = h * cos(th); s.ab[2] = 1e-3 * (urand(seed) - 0.5);
        s.ab[3] = h * v * cos(th);  s.ab[4] = h * sin(th); s.ab[5] = 1e-3 * (urand(seed) - 0.5);
        s.ab[6] = h * 0.1 * (urand(seed) - 0.5); s.ab[7] = h * v / 2.7 * (1.0 + urand(seed));
        s.ab[8] = 0.005 * cos(th); s.ab[9] = 0.005 * sin(th); s.ab[10] = 1e-4 * (urand(seed) - 0.5);
        s.ab[11] = 1e-5 * (urand(seed) - 0.5); s.ab[12] = 1e-5 * (urand(seed) - 0.5); s.ab[13] = 0.005 * v / 2.7;
        for (int j = 0; j < NZ; ++j) s.Ht[j] = (j < NU ? 10.0 : 1e-3) + pow(10.0, 12.0 * urand(seed) - 6.0);
        const double w = pow(10.0, 8.0 * urand(seed) - 4.0), jx = urand(seed) - 0.5, jy = urand(seed) - 0.5;
        s.hxx = w * jx * jx; s.hxy = w * jy * jx; s.hyy = w * jy * jy;
    }
    std::vector<double> rbs(st.size() * NX);
    for (auto &v : rbs) v = 1e-2 * (urand(seed) - 0.5);
    double *d_rb, *d_prb1, *d_prb2;
    hipMalloc(&d_rb, rbs.size() * 8); hipMalloc(&d_prb1, rbs.size() * 8); hipMalloc(&d_prb2, rbs.size() * 8);
    hipMemcpy(d_rb, rbs.data(), rbs.size() * 8, hipMemcpyHostToDevice);
    StageIn *d_st; StageOut *d_o1, *d_o2; double *d_p1, *d_p2; long long *d_cy;
    hipMalloc(&d_st, st.size() * sizeof(StageIn));
    hipMalloc(&d_o1, st.size() * sizeof(StageOut)); hipMalloc(&d_o2, st.size() * sizeof(StageOut));
    hipMalloc(&d_p1, (size_t)WG * 15 * 8); hipMalloc(&d_p2, (size_t)WG * 15 * 8);
    hipMalloc(&d_cy, (size_t)WG * 8);
    hipMemcpy(d_st, st.data(), st.size() * sizeof(StageIn), hipMemcpyHostToDevice);
    std::vector<StageOut> o1(st.size()), o2(st.size());
    std::vector<double> p1((size_t)WG * 15), p2((size_t)WG * 15);
    std::vector<long long> cy(WG);
    auto report = [&](const char *name, float ms) {
        hipMemcpy(cy.data(), d_cy, (size_t)WG * 8, hipMemcpyDeviceToHost);
        double sum = 0.0;
        for (long long v : cy) sum += (double)v;
        printf("%-28s %8.1f cycles per stage step (mean over %d wavefronts), %7.3f ms for %d chains of %d stages each\n", name,
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
        if (pass) report("PIPE (lane pipeline)", ms);
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
            if (memcmp(&p1[(size_t)w * 15], &p2[(size_t)w * 15], 15 * 8) != 0) ++bad;
        }
        printf("%-28s factors and P(0) of %d problems %s the lane pipeline's (%zu records differ)\n", name, WG,
               bad ? "DIFFER from" : "equal, bit for bit,", bad);
        return bad == 0;
    };
    bool ok = true;
#define RUN_SPREAD(CH_)                                                                                      \
    for (int pass = 0; pass < 2; ++pass) {                                                                   \
        hipMemset(d_o2, 0, st.size() * sizeof(StageOut));                                                    \
        hipEventRecord(e0);                                                                                  \
        hipLaunchKernelGGL(k_spread<CH_>, dim3(WG), dim3(64), 0, 0, d_st, hB, N, reps, d_o2, d_p2, d_cy);    \
        hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);                       \
        if (pass) { report("SPREAD, " #CH_ " stages staged", ms); ok = check("SPREAD, " #CH_ " stages staged") && ok; } \
    }
    RUN_SPREAD(64)
    RUN_SPREAD(13)
    // second variant, with P(k+1) rb riding along
    hipMemset(d_prb1, 0, rbs.size() * 8);
    hipLaunchKernelGGL(k_pipe_prb, dim3(WG), dim3(64), 0, 0, d_st, d_rb, hB, N, d_prb1);
    std::vector<double> q1(rbs.size()), q2(rbs.size());
    hipMemcpy(q1.data(), d_prb1, rbs.size() * 8, hipMemcpyDeviceToHost);
#define RUN_SPREAD2(CH_)                                                                                     \
    for (int pass = 0; pass < 2; ++pass) {                                                                   \
        hipMemset(d_o2, 0, st.size() * sizeof(StageOut));                                                    \
        hipMemset(d_prb2, 0, rbs.size() * 8);                                                                \
        hipEventRecord(e0);                                                                                  \
        hipLaunchKernelGGL(k_spread2<CH_>, dim3(WG), dim3(64), 0, 0, d_st, d_rb, hB, N, reps, d_o2, d_prb2, d_p2, d_cy); \
        hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);                       \
        if (pass) {                                                                                          \
            report("SPREAD2, " #CH_ " stages staged", ms);                                                   \
            ok = check("SPREAD2, " #CH_ " stages staged") && ok;                                             \
            hipMemcpy(q2.data(), d_prb2, rbs.size() * 8, hipMemcpyDeviceToHost);                             \
            size_t badq = 0;                                                                                 \
            for (int w = 0; w < WG; ++w)                                                                     \
                for (int k = 0; k < N; ++k)                                                                  \
                    if (memcmp(&q1[((size_t)w * 64 + k) * NX], &q2[((size_t)w * 64 + k) * NX], NX * 8) != 0) ++badq; \
            printf("%-28s P(k+1) rb: %zu stages differ\n", "", badq);                                       \
            ok = ok && badq == 0;                                                                            \
        }                                                                                                    \
    }
    RUN_SPREAD2(64)
    RUN_SPREAD2(11)
    if (hipDeviceSynchronize() != hipSuccess) { printf("HIP error\n"); return 2; }
    return ok ? 0 : 1;
}
