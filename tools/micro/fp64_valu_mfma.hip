// FP64 on the VECTOR pipe beside the MATRIX pipe: v_fma_f64 alone, v_mfma_f64_16x16x4 alone, both interleaved in one
// wave, and both in different waves of a workgroup (waves 0-3 MFMA, 4-7 FMA: one of each per SIMD).
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef double d4 __attribute__((ext_vector_type(4)));
template <int MODE>   // 0 fma, 1 mfma, 2 interleaved, 3 split by wave
__global__ __launch_bounds__(512) void k(double *out, int iters, double a0) {
    d4 acc[4];
    double f[16];
#pragma unroll
    for (int i = 0; i < 4; ++i) acc[i] = (d4){0, 0, 0, 0};
#pragma unroll
    for (int i = 0; i < 16; ++i) f[i] = i * 1e-3;
    const double a = a0 + threadIdx.x * 1e-9, b = 0.999999;
    const bool mf = MODE == 1 || MODE == 2 || (MODE == 3 && (threadIdx.x >> 6) < 4);
    const bool vf = MODE == 0 || MODE == 2 || (MODE == 3 && (threadIdx.x >> 6) >= 4);
    for (int it = 0; it < iters; ++it) {
        if (mf) {
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
        }
        if (vf) {
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int i = 0; i < 16; ++i) f[i] = fma(f[i], b, a);
        }
    }
    double s = 0;
#pragma unroll
    for (int i = 0; i < 4; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
#pragma unroll
    for (int i = 0; i < 16; ++i) s += f[i];
    out[blockIdx.x * 512 + threadIdx.x] = s;
}
template <int MODE>
void run(const char *name, int threads) {
    const int blocks = 512, iters = 4000;
    double *out;
    (void)hipMalloc(&out, sizeof(double) * blocks * 512);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    float ms = 0;
    for (int rep = 0; rep < 3; ++rep) {
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(threads), 0, 0, out, iters, 1.0);
        (void)hipEventRecord(e1);
        (void)hipEventSynchronize(e1);
        (void)hipEventElapsedTime(&ms, e0, e1);
    }
    const double waves = (double)blocks * threads / 64;
    double wm = waves, wv = waves;
    if (MODE == 0) wm = 0;
    if (MODE == 1) wv = 0;
    if (MODE == 3) { wm = waves / 2; wv = waves / 2; }
    const double fl_m = wm * iters * 4 * 2048.0, fl_v = wv * iters * 64 * 128.0;
    printf("%-46s %.3f ms   MFMA %.1f TFLOP/s + VALU %.1f TFLOP/s = %.1f\n", name, ms, fl_m / (ms * 1e-3) / 1e12,
           fl_v / (ms * 1e-3) / 1e12, (fl_m + fl_v) / (ms * 1e-3) / 1e12);
    (void)hipFree(out);
}
int main() {
    run<0>("v_fma_f64 only, 4 waves/CU x2 blocks", 256);
    run<0>("v_fma_f64 only, 8 waves per block", 512);
    run<1>("v_mfma_f64 only, 8 waves per block", 512);
    run<1>("v_mfma_f64 only, 4 waves per block", 256);
    run<1>("v_mfma_f64 only, 2 waves per block", 128);
    run<2>("interleaved in every wave (8 waves)", 512);
    run<3>("waves 0-3 MFMA, waves 4-7 FMA", 512);
    return 0;
}
