// Microbenchmark: what FP64 matrix rate does gfx950 actually sustain?  v_mfma_f64_16x16x4_f64 on NACC independent
// accumulators per wave, 1 / 2 / 4 waves per SIMD on every CU, operands in registers; reports TFLOP/s from the
// kernel's wall time (hipEvents) and the in-kernel clock (s_memtime / s_memrealtime) of wave 0 of every block.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef double d4 __attribute__((ext_vector_type(4)));

template <int NACC>
__global__ __launch_bounds__(256) void mfma_loop(double *out, long long *clk, int iters, double a0, double b0) {
    d4 acc[NACC];
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = (d4){0, 0, 0, 0};
    double a = a0 + threadIdx.x * 1e-9, b = b0 + threadIdx.x * 1e-7;
    const long long c0 = clock64(), w0 = wall_clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
    }
    double s = 0;
#pragma unroll
    for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    const long long c1 = clock64(), w1 = wall_clock64();
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0) {
        clk[2 * blockIdx.x] = c1 - c0;
        clk[2 * blockIdx.x + 1] = w1 - w0;
    }
}

template <int NACC>
static void run(int blocks, int iters) {
    double *out;
    long long *clk, *h = (long long *)malloc(sizeof(long long) * 2 * blocks);
    hipMalloc(&out, sizeof(double) * blocks * 256);
    hipMalloc(&clk, sizeof(long long) * 2 * blocks);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    float ms = 0;
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(mfma_loop<NACC>, dim3(blocks), dim3(256), 0, 0, out, clk, iters, 1.0, 0.5);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1);
    }
    hipMemcpy(h, clk, sizeof(long long) * 2 * blocks, hipMemcpyDeviceToHost);
    double c = 0, w = 0;
    for (int b = 0; b < blocks; ++b) { c += h[2 * b]; w += h[2 * b + 1]; }
    const double nm = (double)blocks * 4 * iters * NACC;
    printf("NACC=%2d blocks=%4d (%d waves/SIMD): kernel %.3f ms  %.1f TFLOP/s  | per wave: %.1f shader cycles per MFMA, "
           "clock %.2f GHz\n", NACC, blocks, blocks / 256, ms, nm * 2048.0 / (ms * 1e-3) / 1e12,
           c / blocks / ((double)iters * NACC), c / w * 0.1);
    hipFree(out);
    hipFree(clk);
    free(h);
}

int main() {
    const int it = 20000;
    run<1>(256, it); run<2>(256, it); run<3>(256, it); run<3>(512, it);   // dependent chains: how many hide the latency
    run<4>(256, it); run<4>(512, it); run<4>(1024, it);
    run<8>(256, it / 2); run<8>(512, it / 2); run<8>(1024, it / 2);
    run<16>(256, it / 4); run<16>(512, it / 4);
    return 0;
}
