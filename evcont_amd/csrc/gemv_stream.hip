// Streaming t-RDM contractions (HBM-bandwidth bound):
//   K5/K4  rows GEMV   y[r]   = sum_c A[r,c] v[c]      (H_ab build,      evcont.py:38-68)
//   K8/K7  cols GEMV   out[c] = sum_r w[r] A[r,c]      (predicted RDMs,  gradients_loewdin.py:343-356)
// Each launch carries TWO problems (the two-body and the one-body t-RDM) so the small
// one rides along with the big one instead of costing a kernel boundary.
#include "common.hpp"
#include "kernels.hpp"

namespace evc {

// ------------------------------------------------------------------ rows GEMV
// Workgroup = 256 lanes x 16 B = one 512-column chunk of RB rows per step; a block owns a
// span of `cps` chunks, keeps v for the chunk in registers and RB accumulators per lane.
// Partials go to ws[span][row]; the consumer sums the spans in fixed order (deterministic).
constexpr int kRB = 8;
constexpr int kChunk = 512;  // columns per workgroup step

__device__ __forceinline__ double2 ld_stream(const double *p) {
    return *reinterpret_cast<const double2 *>(p);
}

__global__ __launch_bounds__(256) void gemv_rows_kernel(GemvRowsLaunch L) {
    __shared__ double red[kRB][4];
    int bid = blockIdx.x;
    const int which = bid >= L.nblk0 ? 1 : 0;
    if (which) bid -= L.nblk0;
    const RowProblem &P = L.p[which];
    const int span = bid % P.nspans;
    const int rb = bid / P.nspans;
    const int64_t row0 = (int64_t)rb * kRB;
    const int nrows = (int)min((int64_t)kRB, P.rows - row0);
    const int tid = threadIdx.x;

    int64_t c = ((int64_t)span * P.cps) * kChunk + tid * 2;
    const int64_t cend = min(P.cols, ((int64_t)(span + 1) * P.cps) * kChunk);
    const double *__restrict__ A = P.A + row0 * P.ld;
    const double *__restrict__ v = P.v;
    const int64_t ld = P.ld;

    double acc[kRB];
#pragma unroll
    for (int r = 0; r < kRB; ++r) acc[r] = 0.0;

    if (nrows == kRB) {
        // full row block, whole chunks: no predicates in the steady state
        for (; c + 1 < cend; c += kChunk) {
            const double2 vv = ld_stream(v + c);
            double2 a[kRB];
#pragma unroll
            for (int r = 0; r < kRB; ++r) a[r] = ld_stream(A + r * ld + c);
#pragma unroll
            for (int r = 0; r < kRB; ++r) acc[r] = fma(a[r].y, vv.y, fma(a[r].x, vv.x, acc[r]));
        }
        if (c < cend) {  // odd `cols`: last single column
            const double vx = v[c];
#pragma unroll
            for (int r = 0; r < kRB; ++r) acc[r] = fma(A[r * ld + c], vx, acc[r]);
        }
    } else {
        for (; c < cend; c += kChunk) {
            const bool two = c + 1 < cend;
            const double vx = v[c], vy = two ? v[c + 1] : 0.0;
            for (int r = 0; r < nrows; ++r) {
                const double ax = A[r * ld + c], ay = two ? A[r * ld + c + 1] : 0.0;
                acc[r] = fma(ay, vy, fma(ax, vx, acc[r]));
            }
        }
    }

    const int lane = tid & 63, wave = tid >> 6;
#pragma unroll
    for (int r = 0; r < kRB; ++r) {
        const double s = wave_sum(acc[r]);
        if (lane == 0) red[r][wave] = s;
    }
    __syncthreads();
    if (tid < nrows) {
        const double s = (red[tid][0] + red[tid][1]) + (red[tid][2] + red[tid][3]);
        P.partial[(int64_t)span * P.rows + row0 + tid] = s;
    }
}

// y[r] = alpha * sum_span partial[r][span]
__global__ void gemv_rows_reduce_kernel(const double *partial, int64_t rows, int nspans, double alpha,
                                        double *y) {
    const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= rows) return;
    double s = 0.0;
    for (int k = 0; k < nspans; ++k) s += partial[(int64_t)k * rows + r];
    y[r] = alpha * s;
}

void plan_rows(RowProblem &P) {
    const int64_t nchunks = ceil_div(P.cols, kChunk);
    const int64_t nrb = ceil_div(P.rows, kRB);
    // aim at >= ~2048 workgroups (8 per CU) while keeping spans >= 4 chunks when possible
    int64_t want_spans = ceil_div(2048, nrb);
    int64_t cps = nchunks / want_spans;
    if (cps < 4) cps = nchunks < 4 ? nchunks : 4;
    if (cps < 1) cps = 1;
    P.cps = (int)cps;
    P.nspans = (int)ceil_div(nchunks, cps);
    P.nblocks = (int)(nrb * P.nspans);
}

size_t rows_ws_doubles(int64_t rows, int64_t cols) {
    RowProblem P{};
    P.rows = rows;
    P.cols = cols;
    plan_rows(P);
    return (size_t)rows * P.nspans;
}

int launch_gemv_rows(RowProblem p0, RowProblem p1, hipStream_t st) {
    GemvRowsLaunch L;
    L.p[0] = p0;
    L.p[1] = p1;
    L.nblk0 = p0.nblocks;
    const int total = p0.nblocks + p1.nblocks;
    if (total == 0) return 0;
    hipLaunchKernelGGL(gemv_rows_kernel, dim3(total), dim3(256), 0, st, L);
    EVC_LAUNCH_CHECK("gemv_rows");
    return 0;
}

// ------------------------------------------------------------------ cols GEMV
// Each lane owns two adjacent columns and walks down the rows with kU independent
// 16-byte loads in flight; the row weights are wave-uniform (scalar loads).
constexpr int kU = 16;

__global__ __launch_bounds__(256) void gemv_cols_kernel(GemvColsLaunch L) {
    int bid = blockIdx.x;
    const int which = bid >= L.nblk0 ? 1 : 0;
    if (which) bid -= L.nblk0;
    const ColProblem &P = L.p[which];
    const int64_t c = (int64_t)bid * kChunk + threadIdx.x * 2;
    if (c >= P.cols) return;
    const double *__restrict__ A = P.A + c;
    const double *__restrict__ w = P.w;
    const int64_t ld = P.ld;
    const int64_t rows = P.rows;
    double sx = 0.0, sy = 0.0;
    if (c + 1 < P.cols) {
        int64_t r = 0;
        for (; r + kU <= rows; r += kU) {
            double2 a[kU];
#pragma unroll
            for (int u = 0; u < kU; ++u) a[u] = ld_stream(A + (r + u) * ld);
#pragma unroll
            for (int u = 0; u < kU; ++u) {
                const double wr = w[r + u];
                sx = fma(wr, a[u].x, sx);
                sy = fma(wr, a[u].y, sy);
            }
        }
        for (; r < rows; ++r) {
            const double2 a = ld_stream(A + r * ld);
            const double wr = w[r];
            sx = fma(wr, a.x, sx);
            sy = fma(wr, a.y, sy);
        }
        *reinterpret_cast<double2 *>(P.out + c) = make_double2(sx, sy);
    } else {
        for (int64_t r = 0; r < rows; ++r) sx = fma(w[r], A[r * ld], sx);
        P.out[c] = sx;
    }
}

int launch_gemv_cols(ColProblem p0, ColProblem p1, hipStream_t st) {
    GemvColsLaunch L;
    L.p[0] = p0;
    L.p[1] = p1;
    L.nblk0 = (int)ceil_div(p0.cols, kChunk);
    const int total = L.nblk0 + (int)ceil_div(p1.cols, kChunk);
    if (total == 0) return 0;
    hipLaunchKernelGGL(gemv_cols_kernel, dim3(total), dim3(256), 0, st, L);
    EVC_LAUNCH_CHECK("gemv_cols");
    return 0;
}

}  // namespace evc

// ------------------------------------------------------------------ C ABI
using namespace evc;

extern "C" size_t evc_gemv_rows_ws_bytes(int64_t rows, int64_t cols) {
    if (rows <= 0 || cols <= 0) return 0;
    return rows_ws_doubles(rows, cols) * sizeof(double);
}

extern "C" int evc_gemv_rows(const double *A, int64_t rows, int64_t cols, int64_t ld, const double *v,
                             double alpha, double *y, void *ws, size_t ws_bytes, void *stream) {
    EVC_REQUIRE(rows > 0 && cols > 0, "evc_gemv_rows: rows=%lld cols=%lld must be positive",
                (long long)rows, (long long)cols);
    EVC_REQUIRE(A && v && y && ws, "evc_gemv_rows: null pointer");
    EVC_REQUIRE(ld >= cols && (ld % 2) == 0, "evc_gemv_rows: ld=%lld must be even and >= cols=%lld",
                (long long)ld, (long long)cols);
    EVC_REQUIRE(aligned16(A) && aligned16(v), "evc_gemv_rows: A and v must be 16-byte aligned");
    EVC_REQUIRE(ws_bytes >= evc_gemv_rows_ws_bytes(rows, cols), "evc_gemv_rows: workspace too small");
    RowProblem P{};
    P.A = A;
    P.v = v;
    P.partial = static_cast<double *>(ws);
    P.rows = rows;
    P.cols = cols;
    P.ld = ld;
    plan_rows(P);
    RowProblem none{};
    int rc = launch_gemv_rows(P, none, as_stream(stream));
    if (rc) return rc;
    hipLaunchKernelGGL(gemv_rows_reduce_kernel, dim3((unsigned)ceil_div(rows, 256)), dim3(256), 0,
                       as_stream(stream), P.partial, rows, P.nspans, alpha, y);
    EVC_LAUNCH_CHECK("gemv_rows_reduce");
    return 0;
}

extern "C" int evc_gemv_cols(const double *A, int64_t rows, int64_t cols, int64_t ld, const double *w,
                             double *out, void *stream) {
    EVC_REQUIRE(rows > 0 && cols > 0, "evc_gemv_cols: rows=%lld cols=%lld must be positive",
                (long long)rows, (long long)cols);
    EVC_REQUIRE(A && w && out, "evc_gemv_cols: null pointer");
    EVC_REQUIRE(ld >= cols && (ld % 2) == 0, "evc_gemv_cols: ld=%lld must be even and >= cols=%lld",
                (long long)ld, (long long)cols);
    EVC_REQUIRE(aligned16(A) && aligned16(out), "evc_gemv_cols: A and out must be 16-byte aligned");
    ColProblem P{};
    P.A = A;
    P.w = w;
    P.out = out;
    P.rows = rows;
    P.cols = cols;
    P.ld = ld;
    ColProblem none{};
    return launch_gemv_cols(P, none, as_stream(stream));
}
