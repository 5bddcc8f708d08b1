// Micro-benchmark: the memory pattern of one-lane-per-problem mode without the arithmetic.
// Each wavefront walks its own workspace ([stage][field][lane], F fields of 512 B per stage),
// loads NL fields of a stage, then stores NS fields of it, stage after stage, `reps` passes.
// 1024 wavefronts of 64 lanes, launch bounds as the solver's (one wavefront per SIMD).
// Prints aggregate GB/s for plain and for non-temporal accesses, and with a dependent
// "compute" delay between the loads and the stores.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

template <int NL, int NS, bool NT, int WORK>
__global__ void __launch_bounds__(64) k_rw(double *ws, size_t wave_doubles, int F, int stages, int reps, double *out)
{
    double *base = ws + (size_t)blockIdx.x * wave_doubles;
    double acc = 0.0;
    for (int r = 0; r < reps; ++r) {
        for (int k = 0; k < stages; ++k) {
            double *sp = base + (size_t)k * F * 64 + threadIdx.x;
            double v[NL > 0 ? NL : 1] = {0.0};
#pragma unroll
            for (int f = 0; f < NL; ++f) v[f] = NT ? __builtin_nontemporal_load(sp + f * 64) : sp[f * 64];
            __builtin_amdgcn_sched_barrier(0);
            double s = acc;
#pragma unroll
            for (int f = 0; f < NL; ++f) s = fma(v[f], 1.0000001, s);
#pragma unroll 8
            for (int w = 0; w < WORK; ++w) s = fma(s, 1.0000001, 1e-9);      // dependent chain, 4+ cycles each
            acc = s;
#pragma unroll
            for (int f = 0; f < NS; ++f) {
                const double o = s + (double)f;
                if (NT) __builtin_nontemporal_store(o, sp + (F - 1 - f) * 64);
                else sp[(F - 1 - f) * 64] = o;
            }
        }
    }
    out[blockIdx.x * 64 + threadIdx.x] = acc;
}

// The same walk with fields stored in pairs, [stage][field pair][lane][2]: 16 bytes per lane and access.
typedef double d2 __attribute__((ext_vector_type(2)));
template <int NL, int NS, bool NT, int WORK>
__global__ void __launch_bounds__(64) k_rw2(double *ws, size_t wave_doubles, int F, int stages, int reps, double *out)
{
    double *base = ws + (size_t)blockIdx.x * wave_doubles;
    double acc = 0.0;
    for (int r = 0; r < reps; ++r) {
        for (int k = 0; k < stages; ++k) {
            d2 *sp = (d2 *)(base + (size_t)k * F * 64) + threadIdx.x;
            d2 v[NL / 2 > 0 ? NL / 2 : 1] = {};
#pragma unroll
            for (int f = 0; f < NL / 2; ++f) v[f] = NT ? __builtin_nontemporal_load(sp + f * 64) : sp[f * 64];
            __builtin_amdgcn_sched_barrier(0);
            double s = acc;
#pragma unroll
            for (int f = 0; f < NL / 2; ++f) { s = fma(v[f].x, 1.0000001, s); s = fma(v[f].y, 1.0000001, s); }
#pragma unroll 8
            for (int w = 0; w < WORK; ++w) s = fma(s, 1.0000001, 1e-9);
            acc = s;
#pragma unroll
            for (int f = 0; f < NS / 2; ++f) {
                d2 o; o.x = s + (double)f; o.y = s - (double)f;
                if (NT) __builtin_nontemporal_store(o, sp + (F / 2 - 1 - f) * 64);
                else sp[(F / 2 - 1 - f) * 64] = o;
            }
        }
    }
    out[blockIdx.x * 64 + threadIdx.x] = acc;
}

template <int NL, int NS, bool NT, int WORK, bool PAIR = false>
void run(const char *name, double *ws, size_t wave_doubles, int F, int stages, double *out, int waves)
{
    const int reps = 20;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    auto kern = PAIR ? k_rw2<NL, NS, NT, WORK> : k_rw<NL, NS, NT, WORK>;
    hipLaunchKernelGGL(kern, dim3(waves), dim3(64), 0, 0, ws, wave_doubles, F, stages, 2, out);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(kern, dim3(waves), dim3(64), 0, 0, ws, wave_doubles, F, stages, reps, out);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    const double bytes = (double)waves * reps * stages * (NL + NS) * 512.0;
    printf("%-44s waves %5d  %8.2f ms  %8.1f GB/s  (%.2f us per stage visit)\n", name, waves, ms, bytes / ms / 1e6,
           ms * 1e3 / (reps * stages));
}

int main()
{
    const int F = 164, stages = 51;
    const size_t wave_doubles = (size_t)stages * F * 64;
    const int maxw = 1024;
    double *ws, *out;
    if (hipMalloc(&ws, maxw * wave_doubles * 8) != hipSuccess) { printf("alloc failed\n"); return 1; }
    hipMalloc(&out, maxw * 64 * 8);
    hipMemset(ws, 0, maxw * wave_doubles * 8);
    run<96, 35, false, 0>("96 loads + 35 stores, no work", ws, wave_doubles, F, stages, out, 1024);
    run<96, 35, true, 0>("96 loads + 35 stores, no work, nt", ws, wave_doubles, F, stages, out, 1024);
    run<96, 35, false, 300>("96 loads + 35 stores, 300 dependent fma", ws, wave_doubles, F, stages, out, 1024);
    run<96, 35, true, 300>("96 loads + 35 stores, 300 dependent fma, nt", ws, wave_doubles, F, stages, out, 1024);
    run<96, 0, true, 0>("96 loads, nt", ws, wave_doubles, F, stages, out, 1024);
    run<0, 96, true, 0>("96 stores, nt", ws, wave_doubles, F, stages, out, 1024);
    run<96, 36, true, 300, true>("pairs: 48 x 16 B loads + 18 x 16 B stores, 300 fma, nt", ws, wave_doubles, F, stages, out, 1024);
    run<96, 36, false, 300, true>("pairs: 48 x 16 B loads + 18 x 16 B stores, 300 fma", ws, wave_doubles, F, stages, out, 1024);
    run<96, 0, true, 0, true>("pairs: 48 x 16 B loads, nt", ws, wave_doubles, F, stages, out, 1024);
    run<0, 96, true, 0, true>("pairs: 48 x 16 B stores, nt", ws, wave_doubles, F, stages, out, 1024);
    run<96, 35, true, 300>("96 loads + 35 stores, 300 fma, nt, 512 waves", ws, wave_doubles, F, stages, out, 512);
    run<96, 35, true, 300>("96 loads + 35 stores, 300 fma, nt, 1 wave", ws, wave_doubles, F, stages, out, 1);
    return 0;
}
