// v_mfma_f64_16x16x4_f64 issue rate by register class of its operands (inline asm pins the classes):
// accumulator in ArchVGPRs ("v") or AccVGPRs ("a"), A/B in ArchVGPRs or AccVGPRs.  One wave per SIMD, 8 accumulators.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef double d4 __attribute__((ext_vector_type(4)));
#define LOOP(ACC_C, AB_C)                                                                                      \
    for (int it = 0; it < iters; ++it) {                                                                      \
        _Pragma("unroll") for (int i = 0; i < 8; ++i)                                                         \
            asm volatile("v_mfma_f64_16x16x4_f64 %0, %1, %2, %0" : "+" ACC_C(acc[i]) : AB_C(a), AB_C(b));     \
    }
template <int MODE>
__global__ __launch_bounds__(256) void k(double *out, int iters, double a0) {
    d4 acc[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) acc[i] = (d4){0, 0, 0, 0};
    double a = a0 + threadIdx.x * 1e-9, b = 0.5 + threadIdx.x * 1e-7;
    if (MODE == 0) LOOP("v", "v")
    if (MODE == 1) LOOP("v", "a")
    if (MODE == 2) LOOP("a", "v")
    if (MODE == 3) LOOP("a", "a")
    double s = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int MODE>
void run(const char *name, int blocks) {
    const int iters = 4000;
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
    printf("%-34s %d waves/SIMD: %.3f ms  %.1f TFLOP/s  (%.0f cycles per MFMA per SIMD at 2.4 GHz)\n", name, blocks / 256,
           ms, nm * 2048.0 / (ms * 1e-3) / 1e12, ms * 1e-3 * 2.4e9 / (nm / 1024.0));
    (void)hipFree(out);
}
int main() {
    for (int blocks = 256; blocks <= 512; blocks *= 2) {
        run<0>("acc ArchVGPR, A/B ArchVGPR", blocks);
        run<1>("acc ArchVGPR, A/B AccVGPR", blocks);
        run<2>("acc AccVGPR,  A/B ArchVGPR", blocks);
        run<3>("acc AccVGPR,  A/B AccVGPR", blocks);
    }
    return 0;
}
