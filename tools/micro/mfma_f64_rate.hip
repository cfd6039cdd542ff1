// Microbenchmark: issue rate of v_mfma_f64_16x16x4_f64 (and f64 FMA) on gfx950.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef double d4 __attribute__((ext_vector_type(4)));
template <int NACC>
__global__ __launch_bounds__(256) void mfma_loop(double *out, int iters, double a0, double b0) {
    d4 acc[NACC];
    for (int i = 0; i < NACC; ++i) acc[i] = (d4){0, 0, 0, 0};
    double a = a0 + threadIdx.x * 1e-9, b = b0;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
    }
    double s = 0;
    for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int NACC>
__global__ __launch_bounds__(256) void fma_loop(double *out, int iters, double a0, double b0) {
    double acc[NACC];
    for (int i = 0; i < NACC; ++i) acc[i] = i;
    double a = a0 + threadIdx.x * 1e-9, b = b0;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) acc[i] = fma(a, acc[i], b);
    }
    double s = 0;
    for (int i = 0; i < NACC; ++i) s += acc[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <typename K>
void run(const char *name, K k, int blocks, int iters, int per_iter, double flop_per) {
    double *out;
    hipMalloc(&out, sizeof(double) * blocks * 256);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, out, 10, 1.0, 0.5);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, out, iters, 1.0, 0.5);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    double waves = blocks * 4.0, ninstr = waves * iters * per_iter;
    // per-SIMD instruction interval assuming waves spread evenly over 1024 SIMDs
    double ns_per_instr_per_simd = ms * 1e6 / (ninstr / 1024.0);
    printf("%-28s blocks=%d  %.3f ms  %.1f ns per instr per SIMD (%.0f cycles @2.4GHz)  %.1f TFLOP/s\n", name, blocks, ms,
           ns_per_instr_per_simd, ns_per_instr_per_simd * 2.4, ninstr * flop_per / (ms * 1e-3) / 1e12);
    hipFree(out);
}
int main() {
    run("mfma_f64 1 acc, 1 wave/SIMD", mfma_loop<1>, 256, 20000, 1, 2048.0);
    run("mfma_f64 4 acc, 1 wave/SIMD", mfma_loop<4>, 256, 5000, 4, 2048.0);
    run("mfma_f64 4 acc, 4 waves/SIMD", mfma_loop<4>, 1024, 5000, 4, 2048.0);
    run("fma_f64 8 acc, 1 wave/SIMD", fma_loop<8>, 256, 20000, 8, 128.0);
    run("fma_f64 8 acc, 4 waves/SIMD", fma_loop<8>, 1024, 20000, 8, 128.0);
    return 0;
}
