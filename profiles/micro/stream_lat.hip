// Micro-benchmark: how long does ONE wavefront take to load NF fields of 512 B
// (8 B per lane) from its own workspace region per "stage", all loads issued
// before the first use?  Run with 1 wave and with 1024 waves.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

template <int NF, int W>   // W = doubles per lane per load (1 or 2)
__global__ void __launch_bounds__(64) k_stream(const double *ws, size_t wave_doubles, int F, int stages, int reps, double *out)
{
    const double *base = ws + (size_t)blockIdx.x * wave_doubles;
    double acc = 0.0;
    for (int r = 0; r < reps; ++r) {
        for (int k = 0; k < stages; ++k) {
            const double *sp = base + (size_t)k * F * 64;
            double v[NF * W];
#pragma unroll
            for (int f = 0; f < NF; ++f) {
                if (W == 1) v[f] = sp[f * 64 + threadIdx.x];
                else {
                    double2 t = reinterpret_cast<const double2 *>(sp + f * 128)[threadIdx.x];
                    v[2 * f] = t.x; v[2 * f + 1] = t.y;
                }
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int f = 0; f < NF * W; ++f) acc = fma(v[f], 1.0000001, acc);
        }
    }
    out[blockIdx.x * 64 + threadIdx.x] = acc;
}

int main(int argc, char **argv)
{
    const int F = 145, stages = 51;
    const size_t wave_doubles = (size_t)stages * F * 64;
    const int maxw = 2048;
    double *ws, *out;
    hipMalloc(&ws, maxw * wave_doubles * 8);
    hipMalloc(&out, maxw * 64 * 8);
    hipMemset(ws, 0, maxw * wave_doubles * 8);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int nw : {1, 4, 256, 1024, 2048}) {
        for (int variant = 0; variant < 3; ++variant) {
            int reps = 20;
            auto launch = [&]() {
                if (variant == 0) hipLaunchKernelGGL((k_stream<96, 1>), dim3(nw), dim3(64), 0, 0, ws, wave_doubles, F, stages, reps, out);
                if (variant == 1) hipLaunchKernelGGL((k_stream<48, 2>), dim3(nw), dim3(64), 0, 0, ws, wave_doubles, F, stages, reps, out);
                if (variant == 2) hipLaunchKernelGGL((k_stream<24, 1>), dim3(nw), dim3(64), 0, 0, ws, wave_doubles, F, stages, reps, out);
            };
            launch();
            hipDeviceSynchronize();
            hipEventRecord(e0);
            launch();
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float ms;
            hipEventElapsedTime(&ms, e0, e1);
            double per_stage_us = ms * 1e3 / (reps * stages);
            int nf = variant == 2 ? 24 : 96;
            double gbs = (double)nw * reps * stages * nf * 512 / (ms * 1e-3) / 1e9;
            printf("waves %4d  %s  %.2f us per stage  %.0f GB/s\n", nw,
                   variant == 0 ? "96 x dwordx2" : variant == 1 ? "48 x dwordx4" : "24 x dwordx2", per_stage_us, gbs);
        }
    }
    return 0;
}
