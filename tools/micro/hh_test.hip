// Unit test of few::householder_tridiag_wave (few_roots.hpp) against a host Householder: build_micro/hh_test [m] [MR]
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include "common.hpp"
namespace evc {
#include "few_roots.hpp"
template <typename T, int MR>
__global__ void hh_kernel(const T *A, int m, T *dd_out, T *ee_out, T *dbg) {
    __shared__ __align__(16) T scr[5 * 32 + 32 * 32 + 64];
    T *vb = scr, *wb = vb + 32, *dd = wb + 32, *ee = dd + 32, *bb = ee + 32, *Vh = bb + 32;
    const int lane = threadIdx.x & 63, j = lane & 31, h = lane >> 5;
    T a[MR];
#pragma unroll
    for (int c = 0; c < MR; ++c) a[c] = (j < m && c < m) ? A[j * m + c] : (T)0;
    few::householder_tridiag_wave<T, MR>(a, m, vb, wb, dd, ee, bb, Vh, (T)1e-30, (T)0);
    (void)h;
    (void)dbg;
    if (lane < m) {
        dd_out[lane] = dd[lane];
        ee_out[lane] = ee[lane];
    }
}
}
template <typename T, int MR>
int run(int m) {
    std::vector<T> A(m * m);
    unsigned long long s = 12345;
    auto rnd = [&]() { s = s * 6364136223846793005ull + 1442695040888963407ull; return (double)(s >> 11) / 9007199254740992.0 - 0.5; };
    for (int i = 0; i < m; ++i) for (int j = 0; j <= i; ++j) A[i * m + j] = A[j * m + i] = (T)rnd();
    T *dA, *dd, *de;
    hipMalloc(&dA, sizeof(T) * m * m); hipMalloc(&dd, sizeof(T) * 32); hipMalloc(&de, sizeof(T) * 32);
    hipMemcpy(dA, A.data(), sizeof(T) * m * m, hipMemcpyHostToDevice);
    T *dg; hipMalloc(&dg, sizeof(T) * 32 * 64 * 8); hipMemset(dg, 0, sizeof(T) * 32 * 64 * 8);
    hipLaunchKernelGGL((evc::hh_kernel<T, MR>), dim3(1), dim3(64), 0, 0, dA, m, dd, de, dg);
    std::vector<T> d(32), e(32);
    hipMemcpy(d.data(), dd, sizeof(T) * 32, hipMemcpyDeviceToHost);
    hipMemcpy(e.data(), de, sizeof(T) * 32, hipMemcpyDeviceToHost);
    // host reference (same algorithm, double)
    std::vector<double> B(m * m);
    for (int i = 0; i < m * m; ++i) B[i] = A[i];
    std::vector<double> rd(m), re(m, 0.0);
    for (int k = 0; k + 2 < m; ++k) {
        double sig = 0; for (int j = k + 1; j < m; ++j) sig += B[j * m + k] * B[j * m + k];
        double xk1 = B[(k + 1) * m + k], alpha = xk1 > 0 ? -sqrt(sig) : sqrt(sig), beta = 1.0 / (sig - xk1 * alpha);
        std::vector<double> v(m, 0.0), p(m, 0.0), w(m, 0.0);
        for (int j = k + 1; j < m; ++j) v[j] = B[j * m + k];
        v[k + 1] -= alpha;
        for (int j = 0; j < m; ++j) { double t = 0; for (int c = 0; c < m; ++c) t += B[j * m + c] * v[c]; p[j] = beta * t; }
        double K = 0; for (int j = 0; j < m; ++j) K += p[j] * v[j]; K *= 0.5 * beta;
        for (int j = 0; j < m; ++j) w[j] = p[j] - K * v[j];
        rd[k] = B[k * m + k]; re[k] = alpha;
        for (int j = 0; j < m; ++j) for (int c = 0; c < m; ++c) B[j * m + c] -= v[j] * w[c] + w[j] * v[c];
    }
    if (m >= 2) { rd[m - 2] = B[(m - 2) * m + m - 2]; rd[m - 1] = B[(m - 1) * m + m - 1]; re[m - 2] = B[(m - 1) * m + m - 2]; } else rd[0] = B[0];
    double worst = 0;
    for (int i = 0; i < m; ++i) { worst = fmax(worst, fabs(rd[i] - d[i])); if (i + 1 < m) worst = fmax(worst, fabs(fabs(re[i]) - fabs((double)e[i]))); }
    printf("m=%d MR=%d sizeof(T)=%d: max |d - d_ref|, ||e| - |e_ref|| = %.3e\n", m, MR, (int)sizeof(T), worst);
    if (worst > (sizeof(T) == 8 ? 1e-12 : 1e-4) && m == 4) {
        std::vector<T> g(32 * 64 * 8);
        hipMemcpy(g.data(), dg, sizeof(T) * g.size(), hipMemcpyDeviceToHost);
        for (int kk = 0; kk < 2; ++kk)
            for (int l : {0, 1, 2, 3, 32, 33, 34, 35}) {
                const T *o = g.data() + (kk * 64 + l) * 8;
                printf("   step %d lane %2d: x %.6f v %.6f p %.6f w %.6f a0 %.6f a1 %.6f sig %.6f K %.6f\n", kk, l, (double)o[0], (double)o[1], (double)o[2], (double)o[3], (double)o[4], (double)o[5], (double)o[6], (double)o[7]);
            }
    }
    if (worst > (sizeof(T) == 8 ? 1e-12 : 1e-4)) {
        for (int i = 0; i < m; ++i) printf("  %2d: d %.6f (ref %.6f)  e %.6f (ref %.6f)\n", i, (double)d[i], rd[i], (double)e[i], re[i]);
        return 1;
    }
    return 0;
}
int main(int argc, char **argv) {
    int bad = 0;
    for (int m : {1, 2, 3, 4, 5, 7, 8, 12, 16, 17, 20, 24, 30, 31, 32}) {
        bad += run<double, 32>(m);
        bad += run<float, 32>(m);
        if (m <= 24) bad += run<double, 24>(m);
        if (m <= 16) bad += run<double, 16>(m), bad += run<float, 16>(m);
        if (m <= 8) bad += run<double, 8>(m);
    }
    printf("%s\n", bad ? "FAILED" : "all good");
    return bad;
}
