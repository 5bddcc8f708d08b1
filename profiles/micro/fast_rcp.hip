// Is 1/x by v_rcp_f64 + two Newton steps + one residual correction (the device's own division sequence with
// the scaling and the special-case fix-up left out: seven dependent operations instead of ten) the same bits as
// the IEEE division 1.0 / x for every x whose exponent is far from the ends of the range?  Random mantissas,
// exponents -900 .. 900, both signs; also values next to powers of two.
// build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fno-fast-math fast_rcp.hip -o fast_rcp
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

__device__ __forceinline__ double fast_rcp(double x)
{
    double y = __builtin_amdgcn_rcp(x);
    double e = __builtin_fma(-x, y, 1.0);
    y = __builtin_fma(y, e, y);
    e = __builtin_fma(-x, y, 1.0);
    y = __builtin_fma(y, e, y);
    const double r = __builtin_fma(-x, y, 1.0);
    return __builtin_fma(r, y, y);
}

__global__ void k(unsigned long long seed, unsigned long long n_per_thread, unsigned long long *bad, double *example)
{
    unsigned long long s = seed + 0x9E3779B97F4A7C15ull * (blockIdx.x * (unsigned long long)blockDim.x + threadIdx.x + 1);
    unsigned long long nb = 0;
    for (unsigned long long i = 0; i < n_per_thread; ++i) {
        s = s * 6364136223846793005ull + 1442695040888963407ull;
        unsigned long long m = s >> 12;                                   // 52 mantissa bits
        s = s * 6364136223846793005ull + 1442695040888963407ull;
        const unsigned long long r = s >> 33;
        long long ex = (long long)(r % 1801) - 900;                       // -900 .. 900
        if ((r & 0xf0000) == 0) m = (r & 1) ? (m & 0xff) : (0xfffffffffffffull - (m & 0xff));      // next to a power of two
        const unsigned long long bits = ((unsigned long long)(ex + 1023) << 52) | m | ((r >> 20 & 1) << 63);
        const double x = __longlong_as_double((long long)bits);
        const double a = 1.0 / x, b = fast_rcp(x);
        if (__double_as_longlong(a) != __double_as_longlong(b)) { ++nb; *example = x; }
    }
    if (nb) atomicAdd(bad, nb);
}

int main()
{
    unsigned long long *d_bad; double *d_ex;
    hipMalloc(&d_bad, 8); hipMalloc(&d_ex, 8);
    hipMemset(d_bad, 0, 8); hipMemset(d_ex, 0, 8);
    const unsigned long long per = 1 << 14;
    hipLaunchKernelGGL(k, dim3(4096), dim3(256), 0, 0, 12345ull, per, d_bad, d_ex);
    unsigned long long bad = 0; double ex = 0;
    hipMemcpy(&bad, d_bad, 8, hipMemcpyDeviceToHost); hipMemcpy(&ex, d_ex, 8, hipMemcpyDeviceToHost);
    printf("%llu reciprocals compared, %llu differ from the IEEE division", 4096ull * 256ull * per, bad);
    if (bad) printf(" (e.g. x = %a)", ex);
    printf("\n");
    return bad ? 1 : 0;
}
