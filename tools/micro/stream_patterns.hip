// Microbenchmark: HBM read bandwidth of a (rows x cols) f64 matrix for different per-instruction
// access shapes (all 16 B per lane, 16 loads in flight per lane, 256-thread blocks).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
// shape 0: a wave-instruction reads 1 KiB contiguous of ONE row (lane -> 16 B)
// shape 1: 16 rows x 64 B (lane (l15,l4): row l15, 16 B at l4)          [MFMA A/B operand shape]
// shape 2: 4 rows x 256 B (lane (l4,l15): row l4, 16 B at l15)          [cols-kernel shape]
template <int SHAPE>
__global__ __launch_bounds__(256) void rd(const double *A, int64_t rows, int64_t ld, int64_t cols, double *out) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int l15 = lane & 15, l4 = lane >> 4;
    double s = 0;
    const int64_t nw = (int64_t)gridDim.x * 4, w = (int64_t)blockIdx.x * 4 + wave;
    if (SHAPE == 0) {
        // work item: (row, 2048-column segment) ; 16 loads of 1 KiB
        const int64_t segs = cols / 2048, items = rows * segs;
        for (int64_t it = w; it < items; it += nw) {
            const double *p = A + (it / segs) * ld + (it % segs) * 2048 + lane * 2;
            double2 v[16];
#pragma unroll
            for (int u = 0; u < 16; ++u) v[u] = *(const double2 *)(p + u * 128);
#pragma unroll
            for (int u = 0; u < 16; ++u) s += v[u].x + v[u].y;
        }
    } else if (SHAPE == 1) {
        // work item: (16-row tile, 128-column segment): 16 loads, each 16 rows x 64 B
        const int64_t segs = cols / 128, items = (rows / 16) * segs;
        for (int64_t it = w; it < items; it += nw) {
            const double *p = A + ((it / segs) * 16 + l15) * ld + (it % segs) * 128 + l4 * 2;
            double2 v[16];
#pragma unroll
            for (int u = 0; u < 16; ++u) v[u] = *(const double2 *)(p + u * 8);
#pragma unroll
            for (int u = 0; u < 16; ++u) s += v[u].x + v[u].y;
        }
    } else {
        // work item: (16 rows as 4 k-steps of 4 rows, 128 columns): 16 loads, each 4 rows x 256 B
        const int64_t segs = cols / 128, items = (rows / 16) * segs;
        for (int64_t it = w; it < items; it += nw) {
            const double *p = A + ((it / segs) * 16 + l4) * ld + (it % segs) * 128 + l15 * 2;
            double2 v[16];
#pragma unroll
            for (int k = 0; k < 4; ++k)
#pragma unroll
                for (int t = 0; t < 4; ++t) v[k * 4 + t] = *(const double2 *)(p + (int64_t)k * 4 * ld + t * 32);
#pragma unroll
            for (int u = 0; u < 16; ++u) s += v[u].x + v[u].y;
        }
    }
    if (s == 12345.678) out[0] = s;
}
template <typename K>
void run(const char *name, K k, const double *A, int64_t rows, int64_t ld, int64_t cols, double *out, int blocks) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, A, rows, ld, cols, out);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int i = 0; i < 5; ++i) hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, A, rows, ld, cols, out);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 5;
    printf("%-34s blocks=%5d  %.3f ms  %.0f GB/s\n", name, blocks, ms, rows * cols * 8.0 / (ms * 1e-3) / 1e9);
}
int main() {
    const int64_t rows = 208, cols = 405504, ld = cols;   // 675 MB
    double *A, *out;
    hipMalloc(&A, rows * ld * 8); hipMalloc(&out, 8);
    hipMemset(A, 0, rows * ld * 8);
    for (int blocks : {512, 1024, 2048, 4096}) {
        run("1 row x 1 KiB", rd<0>, A, rows, ld, cols, out, blocks);
        run("16 rows x 64 B (MFMA operand)", rd<1>, A, rows, ld, cols, out, blocks);
        run("4 rows x 256 B", rd<2>, A, rows, ld, cols, out, blocks);
    }
    return 0;
}
