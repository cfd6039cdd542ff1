// Batched streaming contractions on the FP64 matrix cores (v_mfma_f64_16x16x4_f64), up to 16
// geometries per pass over the t-RDM:
//   K5  Y[row][g]  = sum_c A[row][c] v[g][c]     M <-> 16 rows,        N <-> 16 geometries, K <-> columns
//   K8  O[g][c]    = sum_r w[g][r] A[r][c]       M <-> 16 geometries,  N <-> 16 columns,    K <-> rows
// These are still HBM-bound streams (A is read once, 681 MB at H30/T=20); the matrix cores are
// used because the MFMA sums over K INSIDE the instruction: a lane keeps 4 accumulator doubles per
// 16x16 output tile instead of one private partial sum per (row, geometry), so 16 geometries cost
// no more registers than one and the per-geometry share of the stream drops to 1/16.  Measured on
// MI355X: the MFMA pipe is ~15 % busy in these kernels; what decides their speed is the number of
// 16-byte loads in flight (>= 16 per lane on >= 2 waves per SIMD reaches ~6 TB/s, tools/micro).
//
// Operand maps (cdna_hip_programming.md §3): A[i][k]: lane l holds i = l&15, k = l>>4;
// B[k][j]: k = l>>4, j = l&15; D[i][j]: j = l&15, i = (l>>4) + 4*reg.  The K slot of a lane can
// be ANY column as long as A and B agree, so both operands of K5 are fetched with the same
// coalesced 16-byte pattern: lane (l15,l4) loads columns c+8u+2*l4, +1 of "its" row (a t-RDM row
// for A, a geometry's vector for B) and feeds .x / .y to two MFMAs.
#include "common.hpp"
#include "kernels.hpp"

namespace evc {

typedef double d4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ d4 mfma_f64(double a, double b, d4 c) {
    return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ double2 ld2(const double *p) { return *reinterpret_cast<const double2 *>(p); }

// guarded 16-byte load of columns (c, c+1) of a row with `cols` valid columns
__device__ __forceinline__ double2 ld2_guard(const double *row, int64_t c, int64_t cols) {
    if (c + 1 < cols) return ld2(row + c);
    return make_double2(c < cols ? row[c] : 0.0, 0.0);
}

// ------------------------------------------------------------------ K5: rows GEMM
// Block = (row group of up to RT 16-row tiles, column span); bid = span*nrg + rg so the row groups
// of one span run back to back and re-read its V tile from cache.  The four waves interleave the
// 32-column chunks of the span.  Per chunk a wave loads its V fragments once (4 x 16 B per lane) and
// walks over the row tiles four at a time: 16 independent 16-byte loads of A per lane, then 32 MFMAs
// into RT independent accumulator tiles (a dependent f64 MFMA costs ~3x the issue interval).
// Row tiles past the end of the matrix are skipped (no loads, no MFMAs).  Partials: ws[g][span][row].
constexpr int kMC = 32;  // columns per wave chunk

template <int RT>
__global__ __launch_bounds__(256) void gemv_rows_mfma_kernel(GemvRowsLaunch L, int g0, int G) {
    constexpr int RH = (RT + 3) / 4;  // groups of 4 tiles
    __shared__ double red[4][4][4][64];  // [wave][tile in group][reg][lane]
    int bid = gridDim.x - 1 - blockIdx.x;  // the few blocks of the small second problem are dispatched first
    const int which = bid >= L.nblk0 ? 1 : 0;
    if (which) bid -= L.nblk0;
    const RowProblem &P = L.p[which];
    const int64_t rows = P.rows, cols = P.cols, ld = P.ld;
    const int nrg = (int)((rows + 16 * RT - 1) / (16 * RT));
    const int span = bid / nrg;
    const int rg = bid - span * nrg;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l15 = lane & 15, l4 = lane >> 4;
    const int64_t row_base = (int64_t)rg * 16 * RT;
    const int ntile = (int)min((int64_t)RT, (rows - row_base + 15) / 16);  // live tiles of this row group
    const int64_t cbeg = (int64_t)span * P.cps * 512;
    const int64_t cend = min(cols, (int64_t)(span + 1) * P.cps * 512);
    const bool gok = l15 < G;
    const double *__restrict__ vr = P.v + (int64_t)(g0 + (gok ? l15 : 0)) * P.vstride;
    // row of this lane inside tile t: row_base + 16 t + l15 (ragged last tile: clamped, result discarded)
    const int64_t rlane = row_base + l15;
    d4 acc[RT];
#pragma unroll
    for (int t = 0; t < RT; ++t) acc[t] = (d4){0.0, 0.0, 0.0, 0.0};

    for (int64_t c = cbeg + wave * kMC; c < cend; c += 4 * kMC) {
        const int64_t cc = c + 2 * l4;
        const bool full = c + kMC <= cols;
        double2 b[4];
#pragma unroll
        for (int u = 0; u < 4; ++u)
            b[u] = gok ? (full ? ld2(vr + cc + 8 * u) : ld2_guard(vr, cc + 8 * u, cols)) : make_double2(0.0, 0.0);
#pragma unroll
        for (int h = 0; h < RH; ++h) {
            if (h * 4 < ntile) {  // wave-uniform
                double2 a[4][4];
#pragma unroll
                for (int tt = 0; tt < 4; ++tt) {
                    const int t = h * 4 + tt;
                    if (t < RT) {
                        const double *__restrict__ ar = P.A + min(rlane + 16 * t, rows - 1) * ld;
#pragma unroll
                        for (int u = 0; u < 4; ++u)
                            a[tt][u] = (t < ntile) ? (full ? ld2(ar + cc + 8 * u) : ld2_guard(ar, cc + 8 * u, cols))
                                                   : make_double2(0.0, 0.0);
                    }
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) {
#pragma unroll
                    for (int tt = 0; tt < 4; ++tt)
                        if (h * 4 + tt < RT) acc[h * 4 + tt] = mfma_f64(a[tt][u].x, b[u].x, acc[h * 4 + tt]);
#pragma unroll
                    for (int tt = 0; tt < 4; ++tt)
                        if (h * 4 + tt < RT) acc[h * 4 + tt] = mfma_f64(a[tt][u].y, b[u].y, acc[h * 4 + tt]);
                }
            }
        }
    }
    // cross-wave sum, four tiles at a time
#pragma unroll
    for (int h = 0; h < RH; ++h) {
        __syncthreads();
#pragma unroll
        for (int tt = 0; tt < 4; ++tt)
            if (h * 4 + tt < RT)
#pragma unroll
                for (int r = 0; r < 4; ++r) red[wave][tt][r][lane] = acc[h * 4 + tt][r];
        __syncthreads();
        for (int idx = tid; idx < 4 * 4 * 64; idx += 256) {
            const int ln = idx & 63, r = (idx >> 6) & 3, tt = idx >> 8;
            const int t = h * 4 + tt;
            const int64_t row = row_base + t * 16 + (ln >> 4) + 4 * r;
            const int g = ln & 15;
            if (t < ntile && row < rows && g < G) {
                const double s = (red[0][tt][r][ln] + red[1][tt][r][ln]) + (red[2][tt][r][ln] + red[3][tt][r][ln]);
                P.partial[(int64_t)(g0 + g) * P.pstride + (int64_t)span * rows + row] = s;
            }
        }
    }
}

template <int RT>
static void rows_mfma_launch(GemvRowsLaunch L, int g0, int G, hipStream_t st) {
    for (int k = 0; k < 2; ++k)
        L.p[k].nblocks = L.p[k].nblocks ? (int)(ceil_div(L.p[k].rows, 16 * RT) * L.p[k].nspans) : 0;
    L.nblk0 = L.p[0].nblocks;
    hipLaunchKernelGGL(gemv_rows_mfma_kernel<RT>, dim3(L.p[0].nblocks + L.p[1].nblocks), dim3(256), 0, st, L, g0, G);
}

int launch_gemv_rows_mfma(const GemvRowsLaunch &L, int g0, int G, int tiles, hipStream_t st) {
    if (tiles == 4) rows_mfma_launch<4>(L, g0, G, st);
    else if (tiles == 8) rows_mfma_launch<8>(L, g0, G, st);
    else rows_mfma_launch<16>(L, g0, G, st);
    EVC_LAUNCH_CHECK("gemv_rows_mfma");
    return 0;
}

// ------------------------------------------------------------------ K8: cols GEMM
// Wave = 32*CT columns (CT even/odd tile pairs); block = 4 waves = 128*CT columns; the wave walks
// down the rows 4 at a time (one K step), 4 K steps per iteration so that 4*CT 16-byte loads are in
// flight per lane.  The weights of a row tile (<= 512 rows) are staged in LDS as wl[row][16].
constexpr int kCT = 4;
constexpr int kRowTile = 512;

__global__ __launch_bounds__(256, 3) void gemv_cols_mfma_kernel(GemvColsLaunch L, int g0, int G) {
    extern __shared__ __align__(16) double wl[];  // min(rows, kRowTile) (rounded up to 4) x 16
    int bid = gridDim.x - 1 - blockIdx.x;  // the few blocks of the small second problem are dispatched first
    const int which = bid >= L.nblk0 ? 1 : 0;
    if (which) bid -= L.nblk0;
    const ColProblem &P = L.p[which];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l15 = lane & 15, l4 = lane >> 4;
    const int64_t rows = P.rows, cols = P.cols, ld = P.ld;
    const int64_t c0 = ((int64_t)bid * 4 + wave) * (32 * kCT);   // first column of this wave
    const double *__restrict__ w = P.w + (int64_t)g0 * P.wstride;

    d4 ae[kCT], ao[kCT];
#pragma unroll
    for (int t = 0; t < kCT; ++t) {
        ae[t] = (d4){0.0, 0.0, 0.0, 0.0};
        ao[t] = (d4){0.0, 0.0, 0.0, 0.0};
    }
    // columns of this lane in tile pair t: c0 + 32 t + 2 l15, +1
    const int64_t cl = c0 + 2 * l15;

    for (int64_t r0 = 0; r0 < rows; r0 += kRowTile) {
        const int nr = (int)min((int64_t)kRowTile, rows - r0);
        const int nr4 = (nr + 3) & ~3;
        __syncthreads();
        for (int idx = tid; idx < nr4 * 16; idx += 256) {
            const int r = idx >> 4, g = idx & 15;
            wl[idx] = (r < nr && g < G) ? w[(int64_t)g * P.wstride + r0 + r] : 0.0;
        }
        __syncthreads();
        if (c0 < cols) {
            for (int rb = 0; rb < nr4; rb += 16) {
                double2 x[4][kCT];
                double wf[4];
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) {
                    const int r = rb + 4 * ks + l4;
                    // rows beyond the tile carry zero weights; clamp the address to a valid row
                    const int64_t rr = r0 + (r < nr ? r : nr - 1);
                    const double *row = P.A + rr * ld;
                    const bool live = rb + 4 * ks < nr4;
                    wf[ks] = live ? wl[(rb + 4 * ks + l4) * 16 + l15] : 0.0;
#pragma unroll
                    for (int t = 0; t < kCT; ++t) {
                        const int64_t c = cl + 32 * t;
                        x[ks][t] = live ? ((c + 1 < cols) ? ld2(row + c) : make_double2(c < cols ? row[c] : 0.0, 0.0))
                                        : make_double2(0.0, 0.0);
                    }
                }
#pragma unroll
                for (int ks = 0; ks < 4; ++ks)
#pragma unroll
                    for (int t = 0; t < kCT; ++t) {
                        ae[t] = mfma_f64(wf[ks], x[ks][t].x, ae[t]);
                        ao[t] = mfma_f64(wf[ks], x[ks][t].y, ao[t]);
                    }
            }
        }
    }
    if (c0 < cols) {
#pragma unroll
        for (int t = 0; t < kCT; ++t) {
            const int64_t c = cl + 32 * t;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int g = l4 + 4 * r;
                if (g < G && c < cols) {
                    double *o = P.out + (int64_t)(g0 + g) * P.ostride + c;
                    if (c + 1 < cols) *reinterpret_cast<double2 *>(o) = make_double2(ae[t][r], ao[t][r]);
                    else *o = ae[t][r];
                }
            }
        }
    }
}

int launch_gemv_cols_mfma(GemvColsLaunch L, int g0, int G, hipStream_t st) {
    const int64_t per = 4 * 32 * kCT;
    L.nblk0 = (int)ceil_div(L.p[0].cols, per);
    const int total = L.nblk0 + (int)ceil_div(L.p[1].cols, per);
    if (total == 0) return 0;
    int64_t rmax = L.p[0].rows > L.p[1].rows ? L.p[0].rows : L.p[1].rows;
    if (rmax > kRowTile) rmax = kRowTile;
    const size_t lds = sizeof(double) * 16 * (size_t)((rmax + 3) & ~3);
    static bool attr = false;
    if (!attr) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(gemv_cols_mfma_kernel),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, kRowTile * 16 * (int)sizeof(double));
        attr = true;
    }
    hipLaunchKernelGGL(gemv_cols_mfma_kernel, dim3(total), dim3(256), lds, st, L, g0, G);
    EVC_LAUNCH_CHECK("gemv_cols_mfma");
    return 0;
}

}  // namespace evc
