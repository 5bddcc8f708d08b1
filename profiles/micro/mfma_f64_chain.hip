// Is v_mfma_f64_16x16x4_f64 a k-ordered chain of IEEE fused multiply-adds, bit for bit?
//   D[r][c] = fma(A[r][3],B[3][c], fma(A[r][2],B[2][c], fma(A[r][1],B[1][c], fma(A[r][0],B[0][c], C[r][c]))))
// Operand layout assumed: A[r][k] in lane r + 16k, B[k][c] in lane c + 16k, C/D[r][c] in lane c + 16(r%4),
// register r/4.  Random operands with heavy cancellation, so that a wider internal accumulator or another
// summation order would show.  Prints the number of mismatching elements for the sequential chain and for
// three other orders.  Build: hipcc --offload-arch=gfx950 -O2 -ffp-contract=off mfma_f64_chain.hip -o mfma_f64_chain
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

typedef double v4d __attribute__((ext_vector_type(4)));

__global__ void k_mfma(const double *A, const double *B, const double *C, double *D, int n)
{
    const int l = threadIdx.x;
    for (int t = blockIdx.x; t < n; t += gridDim.x) {
        const double a = A[(size_t)t * 64 + l], b = B[(size_t)t * 64 + l];
        v4d c;
        for (int q = 0; q < 4; ++q) c[q] = C[((size_t)t * 4 + q) * 64 + l];
        v4d d = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
        for (int q = 0; q < 4; ++q) D[((size_t)t * 4 + q) * 64 + l] = d[q];
    }
}

static double rnd(unsigned long long &s)
{
    s = s * 6364136223846793005ull + 1442695040888963407ull;
    const double u = (double)(s >> 11) / 9007199254740992.0;   // [0,1)
    s = s * 6364136223846793005ull + 1442695040888963407ull;
    const int e = (int)((s >> 40) % 41) - 20;                   // exponents 2^-20 .. 2^20
    return ldexp(2.0 * u - 1.0, e);
}

int main()
{
    const int n = 20000;
    std::vector<double> A((size_t)n * 64), B((size_t)n * 64), C((size_t)n * 256), D((size_t)n * 256);
    unsigned long long s = 12345;
    for (auto &v : A) v = rnd(s);
    for (auto &v : B) v = rnd(s);
    for (auto &v : C) v = rnd(s);
    // a fifth of the tiles: exact zeros and ones among the operands, and C = 0 (the shapes the solver feeds)
    for (int t = 0; t < n; t += 5)
        for (int l = 0; l < 64; ++l) {
            if (l % 3 == 0) A[(size_t)t * 64 + l] = 0.0;
            if (l % 7 == 0) B[(size_t)t * 64 + l] = 1.0;
            for (int q = 0; q < 4; ++q) C[((size_t)t * 4 + q) * 64 + l] = 0.0;
        }
    double *dA, *dB, *dC, *dD;
    hipMalloc(&dA, A.size() * 8); hipMalloc(&dB, B.size() * 8); hipMalloc(&dC, C.size() * 8); hipMalloc(&dD, D.size() * 8);
    hipMemcpy(dA, A.data(), A.size() * 8, hipMemcpyHostToDevice);
    hipMemcpy(dB, B.data(), B.size() * 8, hipMemcpyHostToDevice);
    hipMemcpy(dC, C.data(), C.size() * 8, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k_mfma, dim3(256), dim3(64), 0, 0, dA, dB, dC, dD, n);
    if (hipDeviceSynchronize() != hipSuccess) { printf("kernel failed\n"); return 1; }
    hipMemcpy(D.data(), dD, D.size() * 8, hipMemcpyDeviceToHost);
    long bad_seq = 0, bad_rev = 0, bad_sum = 0, bad_pair = 0, total = 0;
    for (int t = 0; t < n; ++t)
        for (int r = 0; r < 16; ++r)
            for (int c = 0; c < 16; ++c) {
                const double *a = &A[(size_t)t * 64], *b = &B[(size_t)t * 64];
                const double c0 = C[((size_t)t * 4 + r / 4) * 64 + c + 16 * (r % 4)];
                const double got = D[((size_t)t * 4 + r / 4) * 64 + c + 16 * (r % 4)];
                auto p = [&](int k) { return std::pair<double, double>(a[r + 16 * k], b[c + 16 * k]); };
                double seq = c0;
                for (int k = 0; k < 4; ++k) seq = fma(p(k).first, p(k).second, seq);
                double rev = c0;
                for (int k = 3; k >= 0; --k) rev = fma(p(k).first, p(k).second, rev);
                double sum = 0.0;
                for (int k = 0; k < 4; ++k) sum = fma(p(k).first, p(k).second, sum);
                sum = sum + c0;
                double pair = fma(p(1).first, p(1).second, p(0).first * p(0).second) + fma(p(3).first, p(3).second, p(2).first * p(2).second) + c0;
                ++total;
                bad_seq += memcmp(&got, &seq, 8) != 0 && !(got == seq);
                bad_rev += !(got == rev);
                bad_sum += !(got == sum);
                bad_pair += !(got == pair);
            }
    printf("elements %ld  mismatches: k-ordered fma chain %ld | reverse chain %ld | sum then + C %ld | pairwise %ld\n",
           total, bad_seq, bad_rev, bad_sum, bad_pair);
    return bad_seq == 0 ? 0 : 2;
}
