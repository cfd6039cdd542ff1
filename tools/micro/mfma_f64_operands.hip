// Does the FP64 MFMA rate depend on where its operands come from?  Same loop as mfma_f64_peak.hip with (a) one fixed
// A/B register pair, (b) 8 different A and 8 different B registers in rotation, (c) B taken from another MFMA's
// accumulator (as the pair transform does for X^T (M X)).  2 waves per SIMD, 8 accumulators.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef double d4 __attribute__((ext_vector_type(4)));
template <int MODE>
__global__ __launch_bounds__(256) void k(double *out, int iters, double a0) {
    d4 acc[8];
    double a[8], b[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) { acc[i] = (d4){0, 0, 0, 0}; a[i] = a0 + threadIdx.x * 1e-9 + i; b[i] = 0.5 + i * 1e-3; }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            if (MODE == 0) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[0], b[0], acc[i], 0, 0, 0);
            if (MODE == 1) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[i], b[(i + 3) & 7], acc[i], 0, 0, 0);
            if (MODE == 2) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[i], acc[(i + 4) & 7][i & 3], acc[i], 0, 0, 0);
        }
    }
    double s = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int MODE>
void run(const char *name) {
    const int blocks = 512, iters = 5000;
    double *out;
    (void)hipMalloc(&out, sizeof(double) * blocks * 256);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    float ms = 0;
    for (int rep = 0; rep < 3; ++rep) {
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, out, iters, 1.0);
        (void)hipEventRecord(e1);
        (void)hipEventSynchronize(e1);
        (void)hipEventElapsedTime(&ms, e0, e1);
    }
    const double nm = (double)blocks * 4 * iters * 8;
    printf("%-44s %.3f ms  %.1f TFLOP/s  (%.0f cycles per MFMA per SIMD at 2.4 GHz)\n", name, ms,
           nm * 2048.0 / (ms * 1e-3) / 1e12, ms * 1e-3 * 2.4e9 / (nm / 1024.0));
    (void)hipFree(out);
}
int main() {
    run<0>("fixed A/B registers");
    run<1>("8 A x 8 B registers in rotation");
    run<2>("B from another accumulator");
    return 0;
}
