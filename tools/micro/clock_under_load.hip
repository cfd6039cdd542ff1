// Microbenchmark: does the shader clock hold under FP64 matrix load, under a pure HBM stream, and
// under both at once?  The same per-wave instruction sequence (a fixed number of
// v_mfma_f64_16x16x4_f64 on 4 independent accumulators, one wave per SIMD) is timed (i) on ONE
// workgroup, the rest of the chip idle, and (ii) on every CU; optionally a streaming read runs beside
// it in other waves.  MFMAs on different CUs share nothing but the power/clock domain, so a longer
// per-wave time at full occupancy is clock throttling, not contention.
// In-kernel: clock64() (s_memtime) and wall_clock64() (constant-rate counter) deltas of wave 0.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef double d4 __attribute__((ext_vector_type(4)));

// waves [0, mfma_waves) of a block run MFMAs, the others stream `src`
__global__ __launch_bounds__(512) void load_kernel(double *out, long long *clk, int iters, int mfma_waves,
                                                   const double2 *__restrict__ src, long n2, int passes) {
    const int wave = threadIdx.x >> 6;
    if (wave < mfma_waves) {
        d4 acc[4];
        for (int i = 0; i < 4; ++i) acc[i] = (d4){0, 0, 0, 0};
        double a = 1.0 + threadIdx.x * 1e-9, b = 0.5;
        const long long c0 = clock64(), w0 = wall_clock64();
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
        }
        double s = 0;
        for (int i = 0; i < 4; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
        const long long c1 = clock64(), w1 = wall_clock64();
        out[blockIdx.x * 512 + threadIdx.x] = s;
        if (threadIdx.x == 0) {
            clk[2 * blockIdx.x] = c1 - c0;
            clk[2 * blockIdx.x + 1] = w1 - w0;
        }
    } else if (src) {
        const int sw = wave - mfma_waves, nsw = (blockDim.x >> 6) - mfma_waves;
        const long stride = (long)gridDim.x * nsw * 64;
        double s = 0;
        for (int p = 0; p < passes; ++p)
            for (long i = ((long)blockIdx.x * nsw + sw) * 64 + (threadIdx.x & 63); i + 7 * stride < n2; i += 8 * stride) {
                double2 v[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) v[u] = src[i + u * stride];
#pragma unroll
                for (int u = 0; u < 8; ++u) s += v[u].x + v[u].y;
            }
        out[blockIdx.x * 512 + threadIdx.x] = s;
    }
}

static void run(const char *name, int blocks, int threads, int mfma_waves, int iters, const double2 *src, long n2,
                int passes, int wall_khz) {
    double *out;
    long long *clk, *h = (long long *)malloc(sizeof(long long) * 2 * blocks);
    hipMalloc(&out, sizeof(double) * blocks * 512);
    hipMalloc(&clk, sizeof(long long) * 2 * blocks);
    hipMemset(clk, 0, sizeof(long long) * 2 * blocks);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    for (int rep = 0; rep < 2; ++rep) {  // rep 0 warms up
        hipEventRecord(e0);
        hipLaunchKernelGGL(load_kernel, dim3(blocks), dim3(threads), 0, 0, out, clk, iters, mfma_waves, src, n2, passes);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
    }
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    hipMemcpy(h, clk, sizeof(long long) * 2 * blocks, hipMemcpyDeviceToHost);
    double c = 0, w = 0;
    int nb = 0;
    for (int b = 0; b < blocks; ++b)
        if (h[2 * b + 1] > 0) { c += h[2 * b]; w += h[2 * b + 1]; ++nb; }
    if (nb) { c /= nb; w /= nb; }
    const double us = w / (wall_khz * 1e-3);             // wall-clock microseconds of the MFMA loop
    const double nmfma = 4.0 * iters;
    printf("%-44s kernel %.3f ms | mfma loop %.1f us, %.1f ns per MFMA (= %.0f cycles at 2.4 GHz), clock64/wall = %.3f",
           name, ms, us, nmfma ? us * 1e3 / nmfma : 0.0, nmfma ? us * 1e3 / nmfma * 2.4 : 0.0, w ? c / w : 0.0);
    if (src) printf(" | stream %.2f TB/s", (double)n2 * 16 * passes / (ms * 1e-3) / 1e12);
    printf("\n");
    hipFree(out);
    hipFree(clk);
    free(h);
}

int main() {
    int wall_khz = 0;
    hipDeviceGetAttribute(&wall_khz, hipDeviceAttributeWallClockRate, 0);
    int clk_khz = 0;
    hipDeviceGetAttribute(&clk_khz, hipDeviceAttributeClockRate, 0);
    printf("wall clock rate %d kHz, advertised shader clock %d kHz\n", wall_khz, clk_khz);
    const long n2 = (long)1 << 26;  // 1 GiB of double2
    double2 *src;
    hipMalloc(&src, n2 * 16);
    hipMemset(src, 0, n2 * 16);
    const int it = 40000;
    run("mfma only, 1 block (4 waves, 1 per SIMD)", 1, 256, 4, it, nullptr, 0, 0, wall_khz);
    run("mfma only, 256 blocks (1 wave per SIMD)", 256, 256, 4, it, nullptr, 0, 0, wall_khz);
    run("mfma only, 512 blocks (2 waves per SIMD)", 512, 256, 4, it / 2, nullptr, 0, 0, wall_khz);
    run("mfma only, 128 blocks (half the CUs)", 128, 256, 4, it, nullptr, 0, 0, wall_khz);
    run("mfma only, 64 blocks (quarter of the CUs)", 64, 256, 4, it, nullptr, 0, 0, wall_khz);
    run("stream only, 1024 blocks x 4 waves", 1024, 256, 0, 0, src, n2, 4, wall_khz);
    // 4 MFMA waves + 4 streaming waves per block, one block per CU x 2
    run("mfma + stream, 512 blocks x (4+4) waves", 512, 512, 4, it / 4, src, n2, 2, wall_khz);
    run("mfma + stream, 256 blocks x (4+4) waves", 256, 512, 4, it / 2, src, n2, 2, wall_khz);
    run("mfma(1/4 duty) + stream, 512 x (1+7) waves", 512, 512, 1, it / 4, src, n2, 2, wall_khz);
    hipFree(src);
    return 0;
}
