// Microbenchmark: same bytes (208 x 405504 f64 = 675 MB), MFMA-operand load shape (16 rows x 64 B per
// instruction), different work decompositions.  Finds which structure keeps HBM busy.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef double d4 __attribute__((ext_vector_type(4)));
// mode 0: grid-stride over (tile, 128-col segment) items; 16 loads then use
// mode 1: block = (row group of TILES tiles, span of `spanc` cols); waves interleave 32-col chunks; per chunk TILES*4 loads
// mode 2: as 1 + MFMAs (2 per load) with 2*TILES chains
template <int TILES, int MODE>
__global__ __launch_bounds__(256) void k(const double *A, int64_t rows, int64_t ld, int64_t cols, int64_t spanc, double *out) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, l15 = lane & 15, l4 = lane >> 4;
    double s = 0;
    d4 acc[TILES], acd[TILES];
    for (int t = 0; t < TILES; ++t) { acc[t] = (d4){0,0,0,0}; acd[t] = (d4){0,0,0,0}; }
    const int nspans = (int)((cols + spanc - 1) / spanc);
    const int span = blockIdx.x % nspans, rg = blockIdx.x / nspans;
    const int64_t cbeg = (int64_t)span * spanc, cend = cbeg + spanc < cols ? cbeg + spanc : cols;
    const double *Ar[TILES];
    for (int t = 0; t < TILES; ++t) Ar[t] = A + ((int64_t)rg * 16 * TILES + t * 16 + l15) * ld;
    for (int64_t c = cbeg + wave * 32; c < cend; c += 128) {
        double2 a[TILES][4];
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int t = 0; t < TILES; ++t) a[t][u] = *(const double2 *)(Ar[t] + c + 2 * l4 + 8 * u);
        if (MODE == 2) {
#pragma unroll
            for (int u = 0; u < 4; ++u) {
#pragma unroll
                for (int t = 0; t < TILES; ++t) acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[t][u].x, 1.0, acc[t], 0, 0, 0);
#pragma unroll
                for (int t = 0; t < TILES; ++t) acd[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[t][u].y, 2.0, acd[t], 0, 0, 0);
            }
        } else {
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int t = 0; t < TILES; ++t) s += a[t][u].x + a[t][u].y;
        }
    }
    for (int t = 0; t < TILES; ++t) s += acc[t][0] + acd[t][0];
    if (s == 12345.678) out[0] = s;
}
template <typename K>
void run(const char *name, K kern, const double *A, int64_t rows, int64_t ld, int64_t cols, int64_t spanc, int tiles, double *out) {
    const int nspans = (int)((cols + spanc - 1) / spanc);
    const int blocks = (int)(rows / (16 * tiles)) * nspans;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, A, rows, ld, cols, spanc, out);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int i = 0; i < 5; ++i) hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, A, rows, ld, cols, spanc, out);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 5;
    printf("%-28s span=%6ld blocks=%5d  %.3f ms  %.0f GB/s\n", name, (long)spanc, blocks, ms, rows * cols * 8.0 / (ms * 1e-3) / 1e9);
}
int main() {
    const int64_t rows = 192, cols = 405504, ld = cols;
    double *A, *out; hipMalloc(&A, rows * ld * 8); hipMalloc(&out, 8); hipMemset(A, 0, rows * ld * 8);
    for (int64_t spanc : {512, 1024, 2048, 5120}) {
        run("T4 loads only", k<4, 1>, A, rows, ld, cols, spanc, 4, out);
        run("T4 loads+mfma", k<4, 2>, A, rows, ld, cols, spanc, 4, out);
        run("T2 loads only", k<2, 1>, A, rows, ld, cols, spanc, 2, out);
        run("T2 loads+mfma", k<2, 2>, A, rows, ld, cols, spanc, 2, out);
    }
    return 0;
}
