// Streaming t-RDM contractions (HBM-bandwidth bound):
//   K5/K4  rows GEMV   y[g][r]   = sum_c A[r,c] v[g][c]     (H_ab build,      evcont.py:38-68)
//   K8/K7  cols GEMV   out[g][c] = sum_r w[g][r] A[r,c]     (predicted RDMs,  gradients_loewdin.py:343-356)
// A (the t-RDM) is shared by a batch of geometries, so one pass over the 0.68-2.6 GB matrix serves
// G evaluations: G <= 8 per pass with the VALU kernels of this file, 12..32 with the matrix-core
// variants of gemv_mfma.hip (the dispatch at the bottom of this file picks per group).
// Each launch carries TWO problems (the two-body and the one-body t-RDM) so the small one rides
// along with the big one instead of costing a kernel boundary.
#include <stdlib.h>

#include "common.hpp"
#include "kernels.hpp"

namespace evc {

constexpr int kChunk = 512;  // columns per workgroup step (256 lanes x 16 B)

__device__ __forceinline__ double2 ld_stream(const double *p) {
    return *reinterpret_cast<const double2 *>(p);
}

// Wave reduction of N (power of two <= 64) per-lane values with N-1 shuffles instead of 6N: at each
// butterfly step a lane sends the half of its values it does not keep.  On return every lane holds
// the wave-wide sum of value index lane >> (6 - log2 N).
template <int N, int M, int OFF>
__device__ __forceinline__ void multi_reduce_step(double (&v)[N], int lane) {
    if constexpr (M > 1) {
        const bool up = (lane & OFF) != 0;
#pragma unroll
        for (int i = 0; i < M / 2; ++i) {
            const double send = up ? v[i] : v[i + M / 2];
            const double keep = up ? v[i + M / 2] : v[i];
            v[i] = keep + __shfl_xor(send, OFF, kWave);
        }
        multi_reduce_step<N, M / 2, OFF / 2>(v, lane);
    } else if constexpr (OFF >= 1) {
        v[0] += __shfl_xor(v[0], OFF, kWave);
        multi_reduce_step<N, 1, OFF / 2>(v, lane);
    }
}

template <int N>
__device__ __forceinline__ double multi_reduce(double (&v)[N], int lane) {
    multi_reduce_step<N, N, 32>(v, lane);
    return v[0];
}

// ------------------------------------------------------------------ rows GEMV
// Workgroup = 256 lanes x 16 B = one 512-column chunk of RB rows per step; a block owns a span of
// `span_cols` columns, keeps v[g] for the chunk in registers and RB*G accumulators per lane.
// Partials go to ws[g][span][row]; the consumer sums the spans in fixed order (deterministic).
template <int RB, int G>
__global__ __launch_bounds__(256) void gemv_rows_kernel(GemvRowsLaunch L, int g0) {
    constexpr int NV = RB * G;
    __shared__ double red[NV][4];
    int bid = blockIdx.x;
    const int which = bid >= L.nblk0 ? 1 : 0;
    if (which) bid -= L.nblk0;
    const RowProblem &P = L.p[which];
    const int span = bid % P.nspans;
    const int rb = bid / P.nspans;
    const int64_t row0 = (int64_t)rb * RB;
    const int nrows = (int)min((int64_t)RB, P.rows - row0);
    const int tid = threadIdx.x;

    int64_t c = (int64_t)span * P.span_cols + tid * 2;
    const int64_t cend = min(P.cols, (int64_t)(span + 1) * P.span_cols);
    const int64_t ld = P.ld;
    const double *__restrict__ v = P.v + (int64_t)g0 * P.vstride;
    const int64_t vs = P.vstride;
    // rows past the end of a ragged last block re-read the last valid row (results discarded)
    const double *__restrict__ Ar[RB];
#pragma unroll
    for (int r = 0; r < RB; ++r) Ar[r] = P.A + (row0 + (r < nrows ? r : nrows - 1)) * ld;

    double acc[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i) acc[i] = 0.0;

    for (; c + 1 < cend; c += kChunk) {
        double2 a[RB], vv[G];
#pragma unroll
        for (int r = 0; r < RB; ++r) a[r] = ld_stream(Ar[r] + c);
#pragma unroll
        for (int g = 0; g < G; ++g) vv[g] = ld_stream(v + g * vs + c);
#pragma unroll
        for (int r = 0; r < RB; ++r)
#pragma unroll
            for (int g = 0; g < G; ++g) acc[r * G + g] = fma(a[r].y, vv[g].y, fma(a[r].x, vv[g].x, acc[r * G + g]));
    }
    if (c < cend) {  // odd `cols`: last single column
#pragma unroll
        for (int r = 0; r < RB; ++r) {
            const double ax = Ar[r][c];
#pragma unroll
            for (int g = 0; g < G; ++g) acc[r * G + g] = fma(ax, v[g * vs + c], acc[r * G + g]);
        }
    }

    const int lane = tid & 63, wave = tid >> 6;
    const double s = multi_reduce<NV>(acc, lane);
    constexpr int kShift = (NV >= 64) ? 0 : (NV == 32 ? 1 : NV == 16 ? 2 : NV == 8 ? 3 : NV == 4 ? 4 : NV == 2 ? 5 : 6);
    if ((lane & ((1 << kShift) - 1)) == 0) red[lane >> kShift][wave] = s;
    __syncthreads();
    if (tid < NV) {
        const int r = tid / G, g = tid - r * G;
        if (r < nrows) {
            const double t = (red[tid][0] + red[tid][1]) + (red[tid][2] + red[tid][3]);
            P.partial[(int64_t)(g0 + g) * P.pstride + (int64_t)span * P.rows + row0 + r] = t;
        }
    }
}

// Batched variant for G >= 4.  With G vectors the lane-private form above re-reads the v tile once per
// row block (G/RB times the bytes of A through L2 and the texture path) and its RB*G accumulators cut
// the occupancy, i.e. the bytes in flight.  Here the roles are turned: the four waves of a workgroup
// take DIFFERENT rows (RBW each) of the SAME column tile (SUB sub-steps of 128 columns), the v tile
// (G x SUB*128 doubles) is staged once per workgroup in LDS, and every lane issues all RBW*SUB = 16
// independent 16-byte loads of A for the tile before the staging barrier, so a wave keeps 16 KB of the
// stream in flight with only RBW*G accumulators.  Each wave reduces its own rows (no cross-wave sum).
constexpr int kStep = 128;  // columns per sub-step (64 lanes x 16 B)

template <int RBW, int G, int SUB>
__global__ __launch_bounds__(256, 3) void gemv_rows_wr_kernel(GemvRowsLaunch L, int g0) {
    constexpr int NV = RBW * G;
    constexpr int TILE = SUB * kStep;        // columns per iteration
    constexpr int VPT = G * TILE / 256;      // doubles of the v tile each thread stages
    static_assert(VPT % 2 == 0 && (TILE % VPT) == 0, "tile shape");
    __shared__ __align__(16) double vt[G][TILE];
    int bid = blockIdx.x;
    const int which = bid >= L.nblk0 ? 1 : 0;
    if (which) bid -= L.nblk0;
    const RowProblem &P = L.p[which];
    const int span = bid % P.nspans;
    const int rg = bid / P.nspans;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int64_t rows = P.rows, cols = P.cols, ld = P.ld;
    const int64_t row0 = (int64_t)rg * (4 * RBW) + wave * RBW;
    const int64_t cbeg = (int64_t)span * P.span_cols;
    const int64_t cend = min(cols, (int64_t)(span + 1) * P.span_cols);
    const double *__restrict__ v = P.v + (int64_t)g0 * P.vstride;
    const double *__restrict__ Ar[RBW];
#pragma unroll
    for (int r = 0; r < RBW; ++r) {
        int64_t rr = row0 + r;
        if (rr >= rows) rr = rows - 1;  // re-read a valid row, result discarded
        Ar[r] = P.A + rr * ld;
    }
    // staging role of this thread inside the v tile: VPT consecutive doubles of one vector
    const int sg = (tid * VPT) / TILE, sc = (tid * VPT) % TILE;
    const double *vsrc = v + (int64_t)sg * P.vstride;

    double acc[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i) acc[i] = 0.0;

    for (int64_t c0 = cbeg; c0 < cend; c0 += TILE) {
        double2 a[SUB][RBW];
        const bool full = c0 + TILE <= cend;   // (a span need not be a whole number of tiles)
        if (full) {
#pragma unroll
            for (int u = 0; u < SUB; ++u)
#pragma unroll
                for (int r = 0; r < RBW; ++r) a[u][r] = ld_stream(Ar[r] + c0 + u * kStep + lane * 2);
#pragma unroll
            for (int k = 0; k < VPT; k += 2)
                *reinterpret_cast<double2 *>(&vt[sg][sc + k]) = ld_stream(vsrc + c0 + sc + k);
        } else {
#pragma unroll
            for (int u = 0; u < SUB; ++u) {
                const int64_t c = c0 + u * kStep + lane * 2;
#pragma unroll
                for (int r = 0; r < RBW; ++r)
                    a[u][r] = (c + 1 < cend) ? ld_stream(Ar[r] + c) : make_double2(c < cend ? Ar[r][c] : 0.0, 0.0);
            }
#pragma unroll
            for (int k = 0; k < VPT; ++k) vt[sg][sc + k] = (c0 + sc + k < cend) ? vsrc[c0 + sc + k] : 0.0;
        }
        __syncthreads();  // v tile staged
#pragma unroll
        for (int u = 0; u < SUB; ++u)
#pragma unroll
            for (int g = 0; g < G; ++g) {
                const double2 vv = *reinterpret_cast<const double2 *>(&vt[g][u * kStep + lane * 2]);
#pragma unroll
                for (int r = 0; r < RBW; ++r)
                    acc[r * G + g] = fma(a[u][r].y, vv.y, fma(a[u][r].x, vv.x, acc[r * G + g]));
            }
        __syncthreads();  // everyone done with the tile before it is overwritten
    }
    const double s = multi_reduce<NV>(acc, lane);
    constexpr int kShift = (NV >= 64) ? 0 : (NV == 32 ? 1 : NV == 16 ? 2 : 3);
    if ((lane & ((1 << kShift) - 1)) == 0) {
        const int idx = lane >> kShift, r = idx / G, g = idx - r * G;
        if (idx < NV && row0 + r < rows)
            P.partial[(int64_t)(g0 + g) * P.pstride + (int64_t)span * rows + row0 + r] = s;
    }
}

// y[r] = alpha * sum_span partial[span][r]
__global__ void gemv_rows_reduce_kernel(const double *partial, int64_t rows, int nspans, double alpha,
                                        double *y) {
    const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= rows) return;
    double s = 0.0;
    for (int k = 0; k < nspans; ++k) s += partial[(int64_t)k * rows + r];
    y[r] = alpha * s;
}

constexpr int kRBPlan = 8;  // row-block height the workspace/partials are planned for (all variants use it)

// Span plan.  `batched` selects the finer decomposition the batched (G > 1) kernels want (they own whole
// row groups, so they need more spans for the same number of workgroups); the partial buffers are sized
// for the finer plan and the single-geometry kernels simply use fewer spans of the same layout.
void plan_rows(RowProblem &P, bool batched) {
    const int64_t nchunks = ceil_div(P.cols, kChunk);
    const int64_t nrb = ceil_div(P.rows, kRBPlan);
    // aim at `target` 8-row blocks (span count = target / row blocks) while keeping spans >= min_cps chunks
    const int target = batched ? 8192 : 2048;
    const int min_cps = batched ? 2 : 4;
    int64_t want_spans = ceil_div(target, nrb);
    int64_t cps = nchunks / want_spans;
    if (cps < min_cps) cps = nchunks < min_cps ? nchunks : min_cps;
    if (cps < 1) cps = 1;
    P.span_cols = cps * kChunk;
    P.nspans = (int)ceil_div(P.cols, P.span_cols);
    P.nblocks = (int)(nrb * P.nspans);
    P.lds_plan = 0;
}

size_t rows_ws_doubles(int64_t rows, int64_t cols) {
    RowProblem P{};
    P.rows = rows;
    P.cols = cols;
    return (size_t)rows * rows_max_spans(P, false);
}

// The span decomposition (span_cols, nspans -> layout of the partials) is fixed by plan_rows; the row-block
// height RB is a property of the kernel variant only: nblocks = ceil(rows/RB) * nspans.
template <int RB, int G>
static void rows_launch(GemvRowsLaunch L, int g0, hipStream_t st) {
    for (int k = 0; k < 2; ++k)
        L.p[k].nblocks = L.p[k].nblocks ? (int)(ceil_div(L.p[k].rows, RB) * L.p[k].nspans) : 0;
    L.nblk0 = L.p[0].nblocks;
    hipLaunchKernelGGL((gemv_rows_kernel<RB, G>), dim3(L.p[0].nblocks + L.p[1].nblocks), dim3(256), 0, st, L, g0);
    note_kernel(EVC_PROF_ROWS, "gemv_rows_kernel<%d,%d> G=%d", RB, G, G);
}

template <int RBW, int G>
static void rows_wr_launch(GemvRowsLaunch L, int g0, hipStream_t st) {
    for (int k = 0; k < 2; ++k)
        L.p[k].nblocks = L.p[k].nblocks ? (int)(ceil_div(L.p[k].rows, 4 * RBW) * L.p[k].nspans) : 0;
    L.nblk0 = L.p[0].nblocks;
    hipLaunchKernelGGL((gemv_rows_wr_kernel<RBW, G, 16 / RBW>), dim3(L.p[0].nblocks + L.p[1].nblocks), dim3(256), 0, st,
                       L, g0);
    note_kernel(EVC_PROF_ROWS, "gemv_rows_wr_kernel<%d,%d,%d> G=%d", RBW, G, 16 / RBW, G);
}

static constexpr int mfma_min_g() { return 12; }          // groups of >= this many geometries use the matrix cores
static constexpr int mfma_max_g() { return kMaxBatchG; }   // geometries per pass over the matrix
// the grouping of launch_gemv_rows below: true if no group of fewer than mfma_min geometries is left over
bool rows_groups_all_mfma(int count) {
    int left = count;
    while (left > 0) {
        if (left < mfma_min_g()) return false;
        left -= left < mfma_max_g() ? left : mfma_max_g();
    }
    return count > 0;
}

int launch_gemv_rows(RowProblem p0, RowProblem p1, int count, hipStream_t st) {
    GemvRowsLaunch L;
    L.p[0] = p0;
    L.p[1] = p1;
    L.nblk0 = p0.nblocks;
    if (p0.nblocks + p1.nblocks == 0 || count <= 0) return 0;
    const int mfma_min = mfma_min_g(), mfma_max = mfma_max_g();
    int g0 = 0;
    while (g0 < count) {
        const int left = count - g0;
        if (left >= mfma_min) {
            // (the LDS-staged kernel takes up to 64 geometries per pass over the matrix; 64 = 2 x 32 leaves the same
            //  remainder as the grouping rows_groups_all_mfma assumed when the span plan was chosen)
            int gmax = mfma_max;
            const bool lds = p0.nblocks ? p0.lds_plan : p1.lds_plan;
            if (lds && mfma_max == 32 && left > 32 && !(p0.nblocks && p1.nblocks && p1.lds_plan <= 0))
                gmax = rows_lds_max_g(p0, p1);
            const int G = left < gmax ? left : gmax;
            int rc = launch_gemv_rows_mfma(L, g0, G, 0, st);
            if (rc) return rc;
            g0 += G;
            continue;
        }
        if (left >= 8) {
            rows_wr_launch<4, 8>(L, g0, st);   // (wave-rows kernels for 4 / 8 geometries: the lane-private ones were slower)
            g0 += 8;
        } else if (left >= 4) {
            rows_wr_launch<8, 4>(L, g0, st);
            g0 += 4;
        } else if (left >= 2) {
            rows_launch<8, 2>(L, g0, st);
            g0 += 2;
        } else {
            rows_launch<8, 1>(L, g0, st);
            g0 += 1;
        }
        EVC_LAUNCH_CHECK("gemv_rows");
    }
    return 0;
}

// ------------------------------------------------------------------ cols GEMV
// Each lane owns two adjacent columns and walks down the rows with kU independent 16-byte loads in
// flight.  G = 1: the row weights are wave-uniform scalar loads.  G > 1: the weights of a 512-row
// tile are staged in LDS as wl[row][g] and read back as broadcasts.
constexpr int kU = 16;
constexpr int kWTile = 512;

template <int G>
__global__ __launch_bounds__(256) void gemv_cols_kernel(GemvColsLaunch L, int g0) {
    __shared__ double wl[(G > 1) ? kWTile * G : 1];
    int bid = blockIdx.x;
    const int which = bid >= L.nblk0 ? 1 : 0;
    if (which) bid -= L.nblk0;
    const ColProblem &P = L.p[which];
    const int64_t c = (int64_t)bid * kChunk + threadIdx.x * 2;
    const bool active = c < P.cols;
    const bool two = c + 1 < P.cols;
    const double *__restrict__ A = P.A + (active ? c : 0);
    const double *__restrict__ w = P.w + (int64_t)g0 * P.wstride;
    const int64_t ld = P.ld;
    const int64_t rows = P.rows;
    double sx[G], sy[G];
#pragma unroll
    for (int g = 0; g < G; ++g) sx[g] = sy[g] = 0.0;

    if constexpr (G == 1) {
        if (active && two) {
            int64_t r = 0;
            for (; r + kU <= rows; r += kU) {
                double2 a[kU];
#pragma unroll
                for (int u = 0; u < kU; ++u) a[u] = ld_stream(A + (r + u) * ld);
#pragma unroll
                for (int u = 0; u < kU; ++u) {
                    const double wr = w[r + u];
                    sx[0] = fma(wr, a[u].x, sx[0]);
                    sy[0] = fma(wr, a[u].y, sy[0]);
                }
            }
            for (; r < rows; ++r) {
                const double2 a = ld_stream(A + r * ld);
                const double wr = w[r];
                sx[0] = fma(wr, a.x, sx[0]);
                sy[0] = fma(wr, a.y, sy[0]);
            }
        } else if (active) {
            for (int64_t r = 0; r < rows; ++r) sx[0] = fma(w[r], A[r * ld], sx[0]);
        }
    } else {
        for (int64_t r0 = 0; r0 < rows; r0 += kWTile) {
            const int nr = (int)min((int64_t)kWTile, rows - r0);
            __syncthreads();
            // consecutive lanes read consecutive rows of one geometry's weight vector (coalesced)
            for (int idx = threadIdx.x; idx < nr * G; idx += 256) {
                const int g = idx / nr, r = idx - g * nr;
                wl[r * G + g] = w[(int64_t)g * P.wstride + r0 + r];
            }
            __syncthreads();
            if (active && two) {
                int r = 0;
                for (; r + kU <= nr; r += kU) {
                    double2 a[kU];
#pragma unroll
                    for (int u = 0; u < kU; ++u) a[u] = ld_stream(A + (r0 + r + u) * ld);
#pragma unroll
                    for (int u = 0; u < kU; ++u)
#pragma unroll
                        for (int g = 0; g < G; ++g) {
                            const double wr = wl[(r + u) * G + g];
                            sx[g] = fma(wr, a[u].x, sx[g]);
                            sy[g] = fma(wr, a[u].y, sy[g]);
                        }
                }
                for (; r < nr; ++r) {
                    const double2 a = ld_stream(A + (r0 + r) * ld);
#pragma unroll
                    for (int g = 0; g < G; ++g) {
                        const double wr = wl[r * G + g];
                        sx[g] = fma(wr, a.x, sx[g]);
                        sy[g] = fma(wr, a.y, sy[g]);
                    }
                }
            } else if (active) {
                for (int r = 0; r < nr; ++r) {
                    const double ax = A[(r0 + r) * ld];
#pragma unroll
                    for (int g = 0; g < G; ++g) sx[g] = fma(wl[r * G + g], ax, sx[g]);
                }
            }
        }
    }
    if (active) {
#pragma unroll
        for (int g = 0; g < G; ++g) {
            double *o = P.out + (int64_t)(g0 + g) * P.ostride + c;
            if (two) *reinterpret_cast<double2 *>(o) = make_double2(sx[g], sy[g]);
            else *o = sx[g];
        }
    }
}

// Row-split variant for matrices with few columns (the 8-fold compressed layout: 212 chunks of 512 columns do not
// fill 256 CUs): a block owns 128 columns, its four waves take every fourth group of 8 rows and the four partial
// sums are added in fixed order through LDS.  The row weights are wave-uniform loads.
constexpr int kColsRsMaxCols = 200000;
template <int G>
__global__ __launch_bounds__(256) void gemv_cols_rs_kernel(GemvColsLaunch L, int g0) {
    __shared__ double2 red[4][G][64];
    constexpr int U = 16;
    int bid = blockIdx.x;
    const int which = bid >= L.nblk0 ? 1 : 0;
    if (which) bid -= L.nblk0;
    const ColProblem &P = L.p[which];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t c = (int64_t)bid * 128 + lane * 2;
    const bool active = c < P.cols;
    const bool two = c + 1 < P.cols;
    const double *__restrict__ A = P.A + (active ? c : 0);
    const double *__restrict__ w = P.w + (int64_t)g0 * P.wstride;
    const int64_t ld = P.ld, rows = P.rows;
    double sx[G], sy[G];
#pragma unroll
    for (int g = 0; g < G; ++g) sx[g] = sy[g] = 0.0;
    if (active) {
        for (int64_t r = (int64_t)wave * U; r < rows; r += 4 * U) {
            double2 a[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int64_t rr = r + u < rows ? r + u : rows - 1;   // clamped rows get weight 0 below
                a[u] = two ? ld_stream(A + rr * ld) : make_double2(A[rr * ld], 0.0);
            }
#pragma unroll
            for (int u = 0; u < U; ++u)
#pragma unroll
                for (int g = 0; g < G; ++g) {
                    const double wr = r + u < rows ? w[(int64_t)g * P.wstride + r + u] : 0.0;
                    sx[g] = fma(wr, a[u].x, sx[g]);
                    sy[g] = fma(wr, a[u].y, sy[g]);
                }
        }
    }
#pragma unroll
    for (int g = 0; g < G; ++g) red[wave][g][lane] = make_double2(sx[g], sy[g]);
    __syncthreads();
    for (int idx = threadIdx.x; idx < G * 64; idx += 256) {
        const int g = idx >> 6, ln = idx & 63;
        const int64_t cc = (int64_t)bid * 128 + ln * 2;
        if (cc < P.cols) {
            const double2 p0 = red[0][g][ln], p1 = red[1][g][ln], p2 = red[2][g][ln], p3 = red[3][g][ln];
            const double vx = (p0.x + p1.x) + (p2.x + p3.x), vy = (p0.y + p1.y) + (p2.y + p3.y);
            double *o = P.out + (int64_t)(g0 + g) * P.ostride + cc;
            if (cc + 1 < P.cols) *reinterpret_cast<double2 *>(o) = make_double2(vx, vy);
            else *o = vx;
        }
    }
}

template <int G>
static void cols_launch(const GemvColsLaunch &Lin, int total, int g0, hipStream_t st) {
    if (Lin.p[0].cols <= kColsRsMaxCols) {
        GemvColsLaunch L = Lin;
        L.nblk0 = (int)ceil_div(L.p[0].cols, 128);
        const int tot = L.nblk0 + (int)ceil_div(L.p[1].cols, 128);
        hipLaunchKernelGGL((gemv_cols_rs_kernel<G>), dim3(tot), dim3(256), 0, st, L, g0);
        note_kernel(EVC_PROF_COLS, "gemv_cols_rs_kernel<%d>", G);
        return;
    }
    const GemvColsLaunch &L = Lin;
    hipLaunchKernelGGL((gemv_cols_kernel<G>), dim3(total), dim3(256), 0, st, L, g0);
    note_kernel(EVC_PROF_COLS, "gemv_cols_kernel<%d>", G);
}

// Row-slab form for a NARROW matrix with MANY rows -- the one-body t-RDM of a large training set, (T^2, N^2): 10 000 x 784
// at the Zundel shape with 100 training states.  The column-tiled kernels above give such a matrix two to seven
// workgroups that walk all its rows (0.3 ms at 0.2 TB/s, the tail of the whole K8 launch); here a workgroup takes a slab
// of rows x 512 columns (whole rows: 4-6 KB contiguous per row), eight geometries at a time, and leaves a partial row;
// a second, tiny launch adds the slabs in fixed order.
__global__ __launch_bounds__(256) void gemv_cols_slab_kernel(ColProblem P, int count, int nslab) {
    const int slab = blockIdx.x;
    const int64_t c = (int64_t)blockIdx.y * kChunk + threadIdx.x * 2;
    const bool active = c < P.cols, two = c + 1 < P.cols;
    const int64_t rps = (P.rows + nslab - 1) / nslab, r0 = (int64_t)slab * rps, r1 = min(P.rows, r0 + rps);
    const double *__restrict__ A = P.A + (active ? c : 0);
    for (int gb = 0; gb < count; gb += 8) {
        const int ng = min(8, count - gb);
        double2 acc[8];
#pragma unroll
        for (int g = 0; g < 8; ++g) acc[g] = make_double2(0.0, 0.0);
        if (active) {
            int64_t r = r0;
            for (; r + 8 <= r1; r += 8) {   // eight independent row loads in flight
                double2 a[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) a[u] = two ? ld_stream(A + (r + u) * P.ld) : make_double2(A[(r + u) * P.ld], 0.0);
#pragma unroll
                for (int g = 0; g < 8; ++g)
                    if (g < ng) {
                        const double *wp = P.w + (int64_t)(gb + g) * P.wstride + r;
#pragma unroll
                        for (int u = 0; u < 8; ++u) {
                            acc[g].x = fma(wp[u], a[u].x, acc[g].x);
                            acc[g].y = fma(wp[u], a[u].y, acc[g].y);
                        }
                    }
            }
            for (; r < r1; ++r) {
                const double2 a = two ? ld_stream(A + r * P.ld) : make_double2(A[r * P.ld], 0.0);
#pragma unroll
                for (int g = 0; g < 8; ++g)
                    if (g < ng) {
                        const double wr = P.w[(int64_t)(gb + g) * P.wstride + r];
                        acc[g].x = fma(wr, a.x, acc[g].x);
                        acc[g].y = fma(wr, a.y, acc[g].y);
                    }
            }
#pragma unroll
            for (int g = 0; g < 8; ++g)
                if (g < ng) {
                    double *o = P.part + (int64_t)(gb + g) * P.pstride + (int64_t)slab * P.ld + c;
                    if (two) *reinterpret_cast<double2 *>(o) = acc[g];
                    else *o = acc[g].x;
                }
        }
    }
}

__global__ __launch_bounds__(256) void gemv_cols_slab_reduce_kernel(ColProblem P, int nslab) {
    const int64_t c = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (c >= P.cols) return;
    const double *p = P.part + (int64_t)blockIdx.y * P.pstride + c;
    double s0 = 0.0, s1 = 0.0;
    int k = 0;
    for (; k + 2 <= nslab; k += 2) {
        s0 += p[(int64_t)k * P.ld];
        s1 += p[(int64_t)(k + 1) * P.ld];
    }
    if (k < nslab) s0 += p[(int64_t)k * P.ld];
    P.out[(int64_t)blockIdx.y * P.ostride + c] = s0 + s1;
}

int launch_gemv_cols(ColProblem p0, ColProblem p1, int count, hipStream_t st) {
    if (p1.part && p1.cols > 0 && p1.rows >= 1024 && p1.cols <= 16384 && count > 0 && count < mfma_min_g()) {
        // (groups of >= 12 geometries go through the matrix-core kernel, whose four waves split the rows of a tile)
        // the narrow second problem in row slabs (its own two launches), the wide one alone below
        const int nslab = (int)(p1.rows / 64 < kColSlabs ? p1.rows / 64 : kColSlabs);
        hipLaunchKernelGGL(gemv_cols_slab_kernel, dim3((unsigned)nslab, (unsigned)ceil_div(p1.cols, kChunk)), dim3(256), 0,
                           st, p1, count, nslab);
        EVC_LAUNCH_CHECK("gemv_cols_slab");
        hipLaunchKernelGGL(gemv_cols_slab_reduce_kernel, dim3((unsigned)ceil_div(p1.cols, 256), (unsigned)count),
                           dim3(256), 0, st, p1, nslab);
        EVC_LAUNCH_CHECK("gemv_cols_slab_reduce");
        p1.cols = 0;
    }
    GemvColsLaunch L;
    L.p[0] = p0;
    L.p[1] = p1;
    L.nblk0 = (int)ceil_div(p0.cols, kChunk);
    const int total = L.nblk0 + (int)ceil_div(p1.cols, kChunk);
    if (total == 0 || count <= 0) return 0;
    const int mfma_min = mfma_min_g(), mfma_max = mfma_max_g();
    int g0 = 0;
    while (g0 < count) {
        const int left = count - g0;
        if (left >= mfma_min) {
            const int G = left < mfma_max ? left : mfma_max;
            int rc = launch_gemv_cols_mfma(L, g0, G, st);
            if (rc) return rc;
            g0 += G;
            continue;
        }
        if (left >= 8) { cols_launch<8>(L, total, g0, st); g0 += 8; }
        else if (left >= 4) { cols_launch<4>(L, total, g0, st); g0 += 4; }
        else if (left >= 2) { cols_launch<2>(L, total, g0, st); g0 += 2; }
        else { cols_launch<1>(L, total, g0, st); g0 += 1; }
        EVC_LAUNCH_CHECK("gemv_cols");
    }
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
    plan_rows(P, false);
    RowProblem none{};
    int rc = launch_gemv_rows(P, none, 1, as_stream(stream));
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
    return launch_gemv_cols(P, none, 1, as_stream(stream));
}
