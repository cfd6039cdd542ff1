// Microbenchmark 2: add the real kernel's features one by one to the structure that streams at ~5 TB/s.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <stdlib.h>
typedef double d4 __attribute__((ext_vector_type(4)));
// FEAT bit0: V fragments loaded from 16 strided vectors; bit1: `full` guard branch; bit2: lane predicate l15<G on V
template <int TILES, int FEAT>
__global__ __launch_bounds__(256) void k(const double *A, int64_t rows, int64_t ld, int64_t cols, int64_t spanc,
                                         const double *V, int64_t vs, int G, double *out) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, l15 = lane & 15, l4 = lane >> 4;
    d4 acc[TILES], acd[TILES];
    for (int t = 0; t < TILES; ++t) { acc[t] = (d4){0,0,0,0}; acd[t] = (d4){0,0,0,0}; }
    const int nspans = (int)((cols + spanc - 1) / spanc);
    const int span = blockIdx.x % nspans, rg = blockIdx.x / nspans;
    const int64_t cbeg = (int64_t)span * spanc, cend = cbeg + spanc < cols ? cbeg + spanc : cols;
    const double *Ar[TILES];
    for (int t = 0; t < TILES; ++t) Ar[t] = A + ((int64_t)rg * 16 * TILES + t * 16 + l15) * ld;
    const bool gok = (FEAT & 4) ? (l15 < G) : true;
    const double *vr = V + (int64_t)(gok ? l15 : 0) * vs;
    for (int64_t c = cbeg + wave * 32; c < cend; c += 128) {
        double2 a[TILES][4], b[4];
        const int64_t cc = c + 2 * l4;
        const bool full = (FEAT & 2) ? (c + 32 <= cols) : true;
        if (full) {
#pragma unroll
            for (int u = 0; u < 4; ++u) {
#pragma unroll
                for (int t = 0; t < TILES; ++t) a[t][u] = *(const double2 *)(Ar[t] + cc + 8 * u);
                if (FEAT & 1) b[u] = gok ? *(const double2 *)(vr + cc + 8 * u) : make_double2(0.0, 0.0);
                else b[u] = make_double2(1.0, 2.0);
            }
        } else {
#pragma unroll
            for (int u = 0; u < 4; ++u) {
#pragma unroll
                for (int t = 0; t < TILES; ++t) { int64_t x = cc + 8 * u; a[t][u] = (x + 1 < cols) ? *(const double2 *)(Ar[t] + x) : make_double2(x < cols ? Ar[t][x] : 0.0, 0.0); }
                b[u] = make_double2(1.0, 2.0);
            }
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
#pragma unroll
            for (int t = 0; t < TILES; ++t) acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[t][u].x, b[u].x, acc[t], 0, 0, 0);
#pragma unroll
            for (int t = 0; t < TILES; ++t) acd[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[t][u].y, b[u].y, acd[t], 0, 0, 0);
        }
    }
    double s = 0;
    for (int t = 0; t < TILES; ++t) s += acc[t][0] + acd[t][0] + acc[t][1] + acd[t][3];
    if (s == 12345.678) out[0] = s;
}
template <typename K>
void run(const char *name, K kern, const double *A, int64_t rows, int64_t ld, int64_t cols, int64_t spanc, int tiles,
         const double *V, int64_t vs, double *out) {
    const int nspans = (int)((cols + spanc - 1) / spanc);
    const int blocks = (int)(rows / (16 * tiles)) * nspans;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, A, rows, ld, cols, spanc, V, vs, 16, out);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int i = 0; i < 5; ++i) hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, A, rows, ld, cols, spanc, V, vs, 16, out);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 5;
    printf("%-34s span=%6ld blocks=%5d  %.3f ms  %.0f GB/s\n", name, (long)spanc, blocks, ms, rows * cols * 8.0 / (ms * 1e-3) / 1e9);
}
__global__ void fill(double *p, int64_t n, unsigned seed) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    for (; i < n; i += (int64_t)gridDim.x * blockDim.x) { unsigned x = (unsigned)i * 2654435761u + seed; x ^= x >> 13; x *= 1274126177u; p[i] = (double)(int)(x) * 4.6e-10; }
}
int main() {
    const int64_t rows = 192, cols = 405504, ld = cols;
    const int64_t vs = 3400000;   // ~27 MB apart like the per-geometry workspaces
    double *A, *out, *V; hipMalloc(&A, rows * ld * 8); hipMalloc(&out, 8); hipMalloc(&V, 16 * vs * 8);
    for (int pass = 0; pass < 2; ++pass) {
        if (pass == 0) { hipMemset(A, 0, rows * ld * 8); hipMemset(V, 0, 16 * vs * 8); printf("--- zero data\n"); }
        else { hipLaunchKernelGGL(fill, dim3(4096), dim3(256), 0, 0, A, rows * ld, 1u); hipLaunchKernelGGL(fill, dim3(4096), dim3(256), 0, 0, V, 16 * vs, 7u); hipDeviceSynchronize(); printf("--- random data\n"); }
        for (int64_t spanc : {1024, 5120}) {
            run("T4 const B", k<4, 0>, A, rows, ld, cols, spanc, 4, V, vs, out);
            run("T4 +V loads", k<4, 1>, A, rows, ld, cols, spanc, 4, V, vs, out);
            run("T4 +V +guard", k<4, 3>, A, rows, ld, cols, spanc, 4, V, vs, out);
            run("T4 +V +guard +lanepred", k<4, 7>, A, rows, ld, cols, spanc, 4, V, vs, out);
            run("T2 +V +guard +lanepred", k<2, 7>, A, rows, ld, cols, spanc, 2, V, vs, out);
        }
    }
    return 0;
}
