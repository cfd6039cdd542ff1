// Microbenchmark 3: the PRODUCT batched rows kernels (csrc/gemv_mfma.hip, csrc/gemv_stream.hip compiled
// in) on a bare matrix, to bisect what separates them from a plain stream: rows 192/208/210, with and
// without the small second problem, with the V vectors close together or a workspace apart.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I../../include -I../../evcont_amd/csrc rows_insitu.hip \
//         ../../evcont_amd/csrc/gemv_mfma.hip ../../evcont_amd/csrc/gemv_stream.hip -o /tmp/rows_insitu
#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>

#include "kernels.hpp"

namespace evc {
void set_error(const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vfprintf(stderr, fmt, ap);
    va_end(ap);
    fputc('\n', stderr);
}
}  // namespace evc

__global__ void fill(double *p, int64_t n, unsigned seed) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    for (; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        unsigned x = (unsigned)i * 2654435761u + seed;
        x ^= x >> 13;
        x *= 1274126177u;
        p[i] = (double)(int)(x)*4.6e-10;
    }
}

static double *dalloc(int64_t n, unsigned seed) {
    double *p;
    if (hipMalloc(&p, n * 8) != hipSuccess) {
        fprintf(stderr, "hipMalloc %lld failed\n", (long long)n);
        exit(1);
    }
    hipLaunchKernelGGL(fill, dim3(4096), dim3(256), 0, 0, p, n, seed);
    return p;
}

int main(int argc, char **argv) {
    const int G = 16;
    const int64_t cols = 405450, ld = 405456;
    const int64_t rows_list[] = {192, 208, 210, 224};
    const int64_t wsstride = 3800000;  // doubles between the per-geometry workspaces (~30 MB)
    double *A = dalloc(224 * ld, 1u);
    double *V = dalloc(G * wsstride, 7u);
    double *part = dalloc(G * wsstride, 9u);
    double *A1 = dalloc(400 * 900, 3u);
    hipDeviceSynchronize();
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    for (int with_p1 = 0; with_p1 < 2; ++with_p1)
        for (int far = 1; far >= 0; --far)
            for (int64_t rows : rows_list) {
                evc::RowProblem p0{}, p1{};
                p0.A = A;
                p0.v = V;
                p0.partial = part + 2000000;
                p0.rows = rows;
                p0.cols = cols;
                p0.ld = ld;
                p0.vstride = far ? wsstride : ld;
                p0.pstride = wsstride;
                evc::plan_rows(p0, true);
                p1.A = A1;
                p1.v = V + 1000000;
                p1.partial = part + 1500000;
                p1.rows = 400;
                p1.cols = 900;
                p1.ld = 900;
                p1.vstride = far ? wsstride : ld;
                p1.pstride = wsstride;
                evc::plan_rows(p1, true);
                if (!with_p1) p1.nblocks = 0;
                for (int i = 0; i < 3; ++i) evc::launch_gemv_rows(p0, p1, G, 0);
                hipDeviceSynchronize();
                const int reps = 20;
                hipEventRecord(e0);
                for (int i = 0; i < reps; ++i) evc::launch_gemv_rows(p0, p1, G, 0);
                hipEventRecord(e1);
                hipEventSynchronize(e1);
                float ms;
                hipEventElapsedTime(&ms, e0, e1);
                ms /= reps;
                const double bytes = rows * cols * 8.0;
                printf("rows %3lld  p1 %d  V %s  spans %4d cps %d : %.1f us  %.0f GB/s (A only)\n", (long long)rows,
                       with_p1, far ? "far " : "near", p0.nspans, p0.cps, ms * 1e3, bytes / (ms * 1e-3) / 1e9);
            }
    return 0;
}
