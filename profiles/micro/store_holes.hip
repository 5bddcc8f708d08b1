// Micro-benchmark: what lanes that sit out cost a one-lane-per-problem sweep.  The walk of stream_rw
// (96 loads + 35 non-temporal stores of 512 B per stage visit, [stage][field][lane]) with H of the 64
// lanes switched off for a whole pass, as the solver's sweeps run with the lanes that wait for service
// or whose problem lives in a region of its own switched off: the stores of the other lanes then
// cover their 128-byte lines only in part.
#include <hip/hip_runtime.h>
#include <cstdio>

template <int NL, int NS, bool NT = true>
__global__ void __launch_bounds__(64) k_rw(double *ws, size_t wave_doubles, int F, int stages, int reps, int holes,
                                           double *out)
{
    double *base = ws + (size_t)blockIdx.x * wave_doubles;
    double acc = 0.0;
    // the lanes that sit out: spread over the wavefront (every 64/H-th lane)
    const bool off = holes > 0 && (threadIdx.x % (64 / holes)) == 0 && (int)(threadIdx.x / (64 / holes)) < holes;
    if (!off) {
        for (int r = 0; r < reps; ++r) {
            for (int k = 0; k < stages; ++k) {
                double *sp = base + (size_t)k * F * 64 + threadIdx.x;
                double v[NL > 0 ? NL : 1] = {0.0};
#pragma unroll
                for (int f = 0; f < NL; ++f) v[f] = __builtin_nontemporal_load(sp + f * 64);
                __builtin_amdgcn_sched_barrier(0);
                double s = acc;
#pragma unroll
                for (int f = 0; f < NL; ++f) s = fma(v[f], 1.0000001, s);
                acc = s;
#pragma unroll
                for (int f = 0; f < NS; ++f) {
                    if (NT) __builtin_nontemporal_store(s + (double)f, sp + (F - 1 - f) * 64);
                    else sp[(F - 1 - f) * 64] = s + (double)f;
                }
            }
        }
    }
    out[blockIdx.x * 64 + threadIdx.x] = acc;
}

template <int NL, int NS, bool NT = true>
void run(const char *name, double *ws, size_t wave_doubles, int F, int stages, double *out, int waves, int holes)
{
    const int reps = 20;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k_rw<NL, NS, NT>), dim3(waves), dim3(64), 0, 0, ws, wave_doubles, F, stages, 2, holes, out);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL((k_rw<NL, NS, NT>), dim3(waves), dim3(64), 0, 0, ws, wave_doubles, F, stages, reps, holes, out);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    printf("%-28s lanes off %2d  %8.2f ms  (%.2f us per stage visit)\n", name, holes, ms, ms * 1e3 / (reps * stages));
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
    for (int h : {0, 1, 2, 4, 8}) run<96, 35>("96 loads + 35 stores", ws, wave_doubles, F, stages, out, 1024, h);
    for (int h : {0, 1, 2, 4, 8}) run<0, 96>("96 stores", ws, wave_doubles, F, stages, out, 1024, h);
    for (int h : {0, 1, 2, 4, 8}) run<93, 95>("93 loads + 95 stores (B0)", ws, wave_doubles, F, stages, out, 1024, h);
    for (int h : {0, 1, 2, 4}) run<0, 96, false>("96 stores, plain", ws, wave_doubles, F, stages, out, 1024, h);
    for (int h : {0, 1, 2, 4}) run<93, 95, false>("93 ld + 95 st (B0), plain st", ws, wave_doubles, F, stages, out, 1024, h);
    return 0;
}
