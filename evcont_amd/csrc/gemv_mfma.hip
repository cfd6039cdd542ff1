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

// Software-pipelined variant (the default).  Block = (column span, row group of <= kRG 16-row tiles);
// the tiles of the matrix are dealt to BALANCED row groups (a group with one or two tiles costs a
// block almost as much as a full one: its chunk steps are latency bound either way).  The four waves
// interleave the 32-column chunks of the span; a wave keeps the V fragments of a chunk in registers
// for all tiles of the group and issues the loads of its NEXT chunk before the MFMAs of the current
// one (two register buffers).  The steady-state loop body is branch-free so that the compiler's vmcnt
// bookkeeping really leaves the younger loads in flight across the MFMAs (a conditional prefetch
// makes it wait for vmcnt(0)).  The MFMAs of a pass add up to ~60 us of pipe time per SIMD at
// H30/T=20, G=16.  Block -> (span, row group) is XCD-aware: workgroups are dealt round-robin to the
// 8 XCDs, each with its own L2, so the row groups of one span (which re-read the same V tile) get
// block ids that are congruent mod 8.
// Measured (tools/micro/rows_insitu.hip, 210 x 405450, G=16): 146 us with groups of <= 3 tiles,
// 165 us with <= 4, 160 us with <= 2; wider groups (5..7 tiles, alternating half-chunk buffers) ran
// out of registers at two waves per SIMD and were slower (200 us).
#define EVC_LD_B(B_, C_)                                                                             \
    {                                                                                                \
        const int64_t cc_ = (C_) + 2 * l4;                                                           \
        _Pragma("unroll") for (int u = 0; u < 4; ++u) B_[u] = ld2(vr + cc_ + 8 * u);                 \
    }
#define EVC_LD_A(A_, T0_, N_, C_)                                                                    \
    {                                                                                                \
        const int64_t cc_ = (C_) + 2 * l4;                                                           \
        _Pragma("unroll") for (int tt = 0; tt < (N_); ++tt)                                          \
            _Pragma("unroll") for (int u = 0; u < 4; ++u) A_[tt][u] = ld2(ar[(T0_) + tt] + cc_ + 8 * u); \
    }
#define EVC_LD_A_GUARD(A_, T0_, N_, C_)                                                              \
    {                                                                                                \
        const int64_t cc_ = (C_) + 2 * l4;                                                           \
        _Pragma("unroll") for (int tt = 0; tt < (N_); ++tt)                                          \
            _Pragma("unroll") for (int u = 0; u < 4; ++u)                                            \
                A_[tt][u] = ld2_guard(ar[(T0_) + tt], cc_ + 8 * u, cols);                            \
    }
#define EVC_MMA(A_, T0_, N_, B_)                                                                     \
    {                                                                                                \
        _Pragma("unroll") for (int u = 0; u < 4; ++u) {                                              \
            _Pragma("unroll") for (int tt = 0; tt < (N_); ++tt)                                      \
                acc[(T0_) + tt] = mfma_f64(A_[tt][u].x, B_[u].x, acc[(T0_) + tt]);                   \
            _Pragma("unroll") for (int tt = 0; tt < (N_); ++tt)                                      \
                acc[(T0_) + tt] = mfma_f64(A_[tt][u].y, B_[u].y, acc[(T0_) + tt]);                   \
        }                                                                                            \
    }

constexpr int kRG = 4;  // most 16-row tiles per row group (register budget of two waves per SIMD)

// One wave's share of a (span, row group) block with NT live 16-row tiles.  Lanes whose geometry slot
// l15 is >= G read geometry g0's vector (a valid address) and their output columns are discarded.
template <int NT, int MAXT>
__device__ __forceinline__ void rows_pipe_body(const RowProblem &P, int64_t row_base, int64_t cbeg, int64_t cend,
                                               const double *__restrict__ vr, int l15, int l4, int wave,
                                               d4 (&acc)[MAXT]) {
    const int64_t rows = P.rows, cols = P.cols, ld = P.ld;
    const int64_t cfull = min(cend, cols & ~(int64_t)31);  // chunks starting below this are complete
    constexpr int64_t kStep = 4 * kMC;
    const double *__restrict__ ar[NT];
#pragma unroll
    for (int tt = 0; tt < NT; ++tt) ar[tt] = P.A + min(row_base + 16 * tt + l15, rows - 1) * ld;
    int64_t c = cbeg + wave * kMC;
    const int64_t nfull = c < cfull ? (cfull - c + kStep - 1) / kStep : 0;  // complete chunks of this wave
    double2 b0[4], b1[4];
    {
        double2 a0[NT][4], a1[NT][4];
        if (nfull > 0) {
            EVC_LD_B(b0, c);
            EVC_LD_A(a0, 0, NT, c);
            int64_t i = 1;
            for (; i + 1 < nfull; i += 2) {
                EVC_LD_B(b1, c + i * kStep);
                EVC_LD_A(a1, 0, NT, c + i * kStep);
                EVC_MMA(a0, 0, NT, b0);
                EVC_LD_B(b0, c + (i + 1) * kStep);
                EVC_LD_A(a0, 0, NT, c + (i + 1) * kStep);
                EVC_MMA(a1, 0, NT, b1);
            }
            if (i < nfull) {
                EVC_LD_B(b1, c + i * kStep);
                EVC_LD_A(a1, 0, NT, c + i * kStep);
                EVC_MMA(a0, 0, NT, b0);
                EVC_MMA(a1, 0, NT, b1);
            } else {
                EVC_MMA(a0, 0, NT, b0);
            }
            c += nfull * kStep;
        }
        if (c < cend) {  // the ragged last chunk of the matrix (at most one wave of one span per row group)
            const int64_t cc = c + 2 * l4;
#pragma unroll
            for (int u = 0; u < 4; ++u) b0[u] = ld2_guard(vr, cc + 8 * u, cols);
            EVC_LD_A_GUARD(a0, 0, NT, c);
            EVC_MMA(a0, 0, NT, b0);
        }
    }
}

__global__ __launch_bounds__(256, 2) void gemv_rows_mfma_pipe_kernel(GemvRowsLaunch L, int g0, int G) {
    constexpr int MAXT = kRG;
    __shared__ double red[4][4][4][64];  // [wave][tile][reg][lane]
    int b = blockIdx.x;
    int which, span, rg;
    if (b < L.nblk1) {  // the few blocks of the small second problem are dispatched first
        which = 1;
        const int nrg1 = L.nrg[1];
        span = b / nrg1;
        rg = b - span * nrg1;
        if (span >= L.p[1].nspans) return;
    } else {
        which = 0;
        b -= L.nblk1;  // nblk1 is a multiple of 8: b & 7 is still the XCD this block was dealt to
        const int nrg0 = L.nrg[0];
        const int xcd = b & 7, idx = b >> 3;
        const int j = idx / nrg0;
        rg = idx - j * nrg0;
        span = j * 8 + xcd;
        if (span >= L.p[0].nspans) return;
    }
    const RowProblem &P = L.p[which];
    const int64_t rows = P.rows;
    const int tpg = L.tpg[which], trem = L.trem[which];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l15 = lane & 15, l4 = lane >> 4;
    const int64_t row_base = (int64_t)(rg * tpg + min(rg, trem)) * 16;
    const int ntile = tpg + (rg < trem ? 1 : 0);  // tiles of this row group (the last one may be ragged)
    const int64_t cbeg = (int64_t)span * P.cps * 512;
    const int64_t cend = min(P.cols, (int64_t)(span + 1) * P.cps * 512);
    const double *__restrict__ vr = P.v + (int64_t)(g0 + (l15 < G ? l15 : 0)) * P.vstride;
    d4 acc[MAXT];
#pragma unroll
    for (int t = 0; t < MAXT; ++t) acc[t] = (d4){0.0, 0.0, 0.0, 0.0};
    if (ntile == 4) rows_pipe_body<4, MAXT>(P, row_base, cbeg, cend, vr, l15, l4, wave, acc);  // wave-uniform
    else if (ntile == 3) rows_pipe_body<3, MAXT>(P, row_base, cbeg, cend, vr, l15, l4, wave, acc);
    else if (ntile == 2) rows_pipe_body<2, MAXT>(P, row_base, cbeg, cend, vr, l15, l4, wave, acc);
    else if (ntile == 1) rows_pipe_body<1, MAXT>(P, row_base, cbeg, cend, vr, l15, l4, wave, acc);
    // cross-wave sum, four tiles per pass
#pragma unroll
    for (int h = 0; h < (MAXT + 3) / 4; ++h) {
        if (h * 4 < ntile) {
            if (h) __syncthreads();
#pragma unroll
            for (int tt = 0; tt < 4; ++tt)
                if (h * 4 + tt < MAXT)
#pragma unroll
                    for (int r = 0; r < 4; ++r) red[wave][tt][r][lane] = acc[h * 4 + tt][r];
            __syncthreads();
            for (int idx = tid; idx < 4 * 4 * 64; idx += 256) {
                const int ln = idx & 63, r = (idx >> 6) & 3, tt = idx >> 8;
                const int t = h * 4 + tt;
                const int64_t row = row_base + t * 16 + (ln >> 4) + 4 * r;
                const int g = ln & 15;
                if (t < ntile && row < rows && g < G) {
                    const double s =
                        (red[0][tt][r][ln] + red[1][tt][r][ln]) + (red[2][tt][r][ln] + red[3][tt][r][ln]);
                    P.partial[(int64_t)(g0 + g) * P.pstride + (int64_t)span * rows + row] = s;
                }
            }
        }
    }
}

static void rows_mfma_pipe_launch(GemvRowsLaunch L, int g0, int G, int max_tiles, hipStream_t st) {
    int *nrg = L.nrg;
    for (int k = 0; k < 2; ++k) {
        // balanced row groups of at most max_tiles tiles
        const int nt = (int)ceil_div(L.p[k].rows > 0 ? L.p[k].rows : 1, 16);
        nrg[k] = (int)ceil_div(nt, max_tiles);
        L.tpg[k] = nt / nrg[k];
        L.trem[k] = nt % nrg[k];
        if (!L.p[k].nblocks) L.p[k].nspans = 0;
    }
    const int nb0 = L.p[0].nblocks ? 8 * nrg[0] * (int)ceil_div(L.p[0].nspans, 8) : 0;
    const int nb1 = L.p[1].nblocks ? (int)ceil_div((int64_t)nrg[1] * L.p[1].nspans, 8) * 8 : 0;
    L.nblk0 = nb0;
    L.nblk1 = nb1;
    hipLaunchKernelGGL(gemv_rows_mfma_pipe_kernel, dim3(nb0 + nb1), dim3(256), 0, st, L, g0, G);
}

template <int RT>
static void rows_mfma_launch(GemvRowsLaunch L, int g0, int G, hipStream_t st) {
    for (int k = 0; k < 2; ++k)
        L.p[k].nblocks = L.p[k].nblocks ? (int)(ceil_div(L.p[k].rows, 16 * RT) * L.p[k].nspans) : 0;
    L.nblk0 = L.p[0].nblocks;
    hipLaunchKernelGGL(gemv_rows_mfma_kernel<RT>, dim3(L.p[0].nblocks + L.p[1].nblocks), dim3(256), 0, st, L, g0, G);
}

int launch_gemv_rows_mfma(const GemvRowsLaunch &L, int g0, int G, int tiles, hipStream_t st) {
    if (tiles <= 0) rows_mfma_pipe_launch(L, g0, G, tiles == 0 ? 3 : (-tiles > kRG ? kRG : -tiles), st);
    else if (tiles == 4) rows_mfma_launch<4>(L, g0, G, st);
    else if (tiles == 8) rows_mfma_launch<8>(L, g0, G, st);
    else rows_mfma_launch<16>(L, g0, G, st);
    EVC_LAUNCH_CHECK("gemv_rows_mfma");
    return 0;
}

// ------------------------------------------------------------------ K8: cols GEMM
// Wave = 32*CT columns (CT even/odd tile pairs); block = 4 waves = 128*CT columns; the wave walks
// down the rows 4 at a time (one K step), 4 K steps per iteration so that 4*CT 16-byte loads are in
// flight per lane.  The weights of a row tile (<= 512 rows) are staged in LDS as wl[row][16].
constexpr int kRowTile = 256;

// CT: 32-column tile pairs per wave; KSN: K steps (4 rows each) per iteration.
template <int CT, int KSN, int MINB>
__global__ __launch_bounds__(256, MINB) void gemv_cols_mfma_kernel(GemvColsLaunch L, int g0, int G) {
    extern __shared__ __align__(16) double wl[];  // min(rows, kRowTile) (rounded up to 4) x 16
    int bid = gridDim.x - 1 - blockIdx.x;  // the few blocks of the small second problem are dispatched first
    const int which = bid >= L.nblk0 ? 1 : 0;
    if (which) bid -= L.nblk0;
    const ColProblem &P = L.p[which];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l15 = lane & 15, l4 = lane >> 4;
    const int64_t rows = P.rows, cols = P.cols, ld = P.ld;
    const int64_t c0 = ((int64_t)bid * 4 + wave) * (32 * CT);   // first column of this wave
    const double *__restrict__ w = P.w + (int64_t)g0 * P.wstride;

    d4 ae[CT], ao[CT];
#pragma unroll
    for (int t = 0; t < CT; ++t) {
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
            for (int rb = 0; rb < nr4; rb += 4 * KSN) {
                double2 x[KSN][CT];
                double wf[KSN];
#pragma unroll
                for (int ks = 0; ks < KSN; ++ks) {
                    const int r = rb + 4 * ks + l4;
                    // rows beyond the tile carry zero weights; clamp the address to a valid row
                    const int64_t rr = r0 + (r < nr ? r : nr - 1);
                    const double *row = P.A + rr * ld;
                    const bool live = rb + 4 * ks < nr4;
                    wf[ks] = live ? wl[(rb + 4 * ks + l4) * 16 + l15] : 0.0;
#pragma unroll
                    for (int t = 0; t < CT; ++t) {
                        const int64_t c = cl + 32 * t;
                        x[ks][t] = live ? ((c + 1 < cols) ? ld2(row + c) : make_double2(c < cols ? row[c] : 0.0, 0.0))
                                        : make_double2(0.0, 0.0);
                    }
                }
#pragma unroll
                for (int ks = 0; ks < KSN; ++ks)
#pragma unroll
                    for (int t = 0; t < CT; ++t) {
                        ae[t] = mfma_f64(wf[ks], x[ks][t].x, ae[t]);
                        ao[t] = mfma_f64(wf[ks], x[ks][t].y, ao[t]);
                    }
            }
        }
    }
    if (c0 < cols) {
#pragma unroll
        for (int t = 0; t < CT; ++t) {
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

template <int CT, int KSN, int MINB>
static void cols_mfma_launch(GemvColsLaunch L, int g0, int G, size_t lds, hipStream_t st) {
    const int64_t per = 4 * 32 * CT;
    L.nblk0 = (int)ceil_div(L.p[0].cols, per);
    const int total = L.nblk0 + (int)ceil_div(L.p[1].cols, per);
    static bool attr = false;
    if (!attr) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(gemv_cols_mfma_kernel<CT, KSN, MINB>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, kRowTile * 16 * (int)sizeof(double));
        attr = true;
    }
    hipLaunchKernelGGL((gemv_cols_mfma_kernel<CT, KSN, MINB>), dim3(total), dim3(256), lds, st, L, g0, G);
}

int launch_gemv_cols_mfma(GemvColsLaunch L, int g0, int G, hipStream_t st) {
    if (L.p[0].cols + L.p[1].cols == 0) return 0;
    int64_t rmax = L.p[0].rows > L.p[1].rows ? L.p[0].rows : L.p[1].rows;
    if (rmax > kRowTile) rmax = kRowTile;
    const size_t lds = sizeof(double) * 16 * (size_t)((rmax + 3) & ~3);
    // measured at H30/T=20, G=16 (in situ): 96-column waves 196 us, 128-column 205 us, 160-column 222 us,
    // 64-column 202 us; fitting all blocks into one resident round did not help
    static const int ct = getenv("EVC_COLS_CT") ? atoi(getenv("EVC_COLS_CT")) : 3;
    if (ct == 4) cols_mfma_launch<4, 4, 3>(L, g0, G, lds, st);
    else cols_mfma_launch<3, 4, 3>(L, g0, G, lds, st);
    EVC_LAUNCH_CHECK("gemv_cols_mfma");
    return 0;
}

}  // namespace evc
