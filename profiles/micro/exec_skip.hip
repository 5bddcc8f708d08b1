// Does a double-precision vector instruction cost fewer issue cycles when only part of the wavefront is
// active?  One wavefront per workgroup, one workgroup; ILP independent fma chains per lane, lanes
// selected by a mask.  Prints cycles per fma instruction (s_memtime ticks at 100 MHz converted with the
// wall clock of a full-wavefront run is avoided: clock64() counts shader cycles on gfx9).
// build: hipcc --offload-arch=gfx950 -O3 -o exec_skip exec_skip.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

template <int ILP>
__global__ void k(double *out, long long *cyc, uint64_t mask, int iters)
{
    int lane = threadIdx.x;
    double a[ILP];
    for (int i = 0; i < ILP; ++i) a[i] = 1.0 + lane * 1e-3 + i;
    double m = 1.0000001, c = 1e-9;
    long long t0 = 0, t1 = 0;
    if ((mask >> lane) & 1) {
        t0 = wall_clock64();
        long long c0 = clock64();
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < ILP; ++i) a[i] = __builtin_fma(a[i], m, c);
        }
        long long c1 = clock64();
        t1 = wall_clock64();
        double s = 0;
        for (int i = 0; i < ILP; ++i) s += a[i];
        out[lane] = s;
        cyc[lane] = c1 - c0;
        cyc[64 + lane] = t1 - t0;
    }
}

template <int ILP>
void run(const char *name, uint64_t mask)
{
    double *out; long long *cyc;
    hipMalloc(&out, 64 * 8); hipMalloc(&cyc, 128 * 8);
    hipMemset(cyc, 0, 128 * 8);
    int iters = 20000;
    k<ILP><<<1, 64>>>(out, cyc, mask, iters);
    k<ILP><<<1, 64>>>(out, cyc, mask, iters);
    hipDeviceSynchronize();
    long long h[128];
    hipMemcpy(h, cyc, sizeof h, hipMemcpyDeviceToHost);
    int l = __builtin_ctzll(mask);
    printf("%-28s ILP %2d: %6.2f clock64 ticks, %6.3f wall ticks (100 MHz) per fma\n", name, ILP,
           (double)h[l] / ((double)iters * ILP), (double)h[64 + l] / ((double)iters * ILP));
    hipFree(out); hipFree(cyc);
}

int main()
{
    struct { const char *n; uint64_t m; } ms[] = {
        {"all 64 lanes", ~0ull}, {"lanes 0-31", 0xffffffffull}, {"lanes 0-15", 0xffffull},
        {"lane 0", 1ull}, {"lane 37", 1ull << 37}, {"lanes 0,16,32,48", 0x0001000100010001ull},
        {"lanes 32-63", 0xffffffff00000000ull}};
    for (auto &m : ms) { run<1>(m.n, m.m); run<4>(m.n, m.m); run<16>(m.n, m.m); }
    return 0;
}
