// Cost of the cross-lane moves the single-problem recursions can be built from, one wavefront,
// dependent chains and independent streams.  Build: hipcc --offload-arch=gfx950 -O3 dpp_rate.hip -o dpp_rate
#include <hip/hip_runtime.h>
#include <cstdio>

template <int CTRL>
__device__ inline int dpp(int v, int old) { return __builtin_amdgcn_update_dpp(old, v, CTRL, 0xf, 0xf, false); }

template <int KIND>
__device__ inline int mv(int v)
{
    if (KIND == 0) return dpp<0x130>(v, 7);                    // wave_shl:1
    if (KIND == 1) return dpp<0x138>(v, 7);                    // wave_shr:1
    if (KIND == 2) return dpp<0x101>(v, 7);                    // row_shl:1
    if (KIND == 3) return dpp<0x111>(v, 7);                    // row_shr:1
    if (KIND == 4) return dpp<0x111>(v, dpp<0x142>(v, 7));     // row_bcast:15 then row_shr:1 = shift by one lane
    if (KIND == 5) return __builtin_amdgcn_ds_bpermute((int)((threadIdx.x + 1) & 63) << 2, v);
    if (KIND == 6) return __builtin_amdgcn_readlane(v, 5) + (int)threadIdx.x;
    if (KIND == 8) { double x = __hiloint2double(v, v); x = fma(x, 1.0000001, 0.5); return __double2hiint(x) ^ __double2loint(x); }
    return v * (int)(threadIdx.x | 3) + 1;                     // plain VALU (v_mad / v_mul_lo)
}

template <int KIND, bool DEP>
__global__ void k(long long *out, int *sink, int iters)
{
    int a = threadIdx.x, b = a * 3, c = a * 5, d = a * 7;
    long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            if (DEP) { a = mv<KIND>(a); a = mv<KIND>(a); a = mv<KIND>(a); a = mv<KIND>(a); }
            else { a = mv<KIND>(a); b = mv<KIND>(b); c = mv<KIND>(c); d = mv<KIND>(d); }
        }
    }
    long long t1 = __builtin_amdgcn_s_memtime();
    sink[threadIdx.x] = a + b + c + d;
    if (threadIdx.x == 0) out[0] = t1 - t0;
}

template <int KIND>
void run(const char *name, long long *d_out, int *d_sink)
{
    const int iters = 20000;
    long long h[2];
    for (int rep = 0; rep < 2; ++rep) {
        hipLaunchKernelGGL((k<KIND, true>), dim3(1), dim3(64), 0, 0, d_out, d_sink, iters);
        hipMemcpy(&h[0], d_out, 8, hipMemcpyDeviceToHost);
        hipLaunchKernelGGL((k<KIND, false>), dim3(1), dim3(64), 0, 0, d_out, d_sink, iters);
        hipMemcpy(&h[1], d_out, 8, hipMemcpyDeviceToHost);
    }
    printf("%-34s dependent %6.1f  independent %6.1f  (s_memtime ticks per move)\n", name,
           (double)h[0] / (iters * 32.0), (double)h[1] / (iters * 32.0));
}

int main()
{
    long long *d_out; int *d_sink;
    hipMalloc(&d_out, 16); hipMalloc(&d_sink, 256);
    run<7>("v_mul_lo + add (reference)", d_out, d_sink);
    run<7>("v_mul_lo + add (again, warm)", d_out, d_sink);
    run<8>("v_fma_f64 + pack/unpack", d_out, d_sink);
    run<0>("dpp wave_shl:1", d_out, d_sink);
    run<1>("dpp wave_shr:1", d_out, d_sink);
    run<2>("dpp row_shl:1", d_out, d_sink);
    run<3>("dpp row_shr:1", d_out, d_sink);
    run<4>("dpp row_bcast:15 + row_shr:1", d_out, d_sink);
    run<5>("ds_bpermute", d_out, d_sink);
    run<6>("v_readlane + v_add", d_out, d_sink);
    return 0;
}
