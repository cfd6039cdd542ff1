// Batched streaming contractions on the FP64 matrix cores (v_mfma_f64_16x16x4_f64), up to 32
// geometries (GS = 1 or 2 sets of 16) per pass over the t-RDM:
//   K5  Y[row][g]  = sum_c A[row][c] v[g][c]     M <-> 16 rows,        N <-> 16 geometries, K <-> columns
//   K8  O[g][c]    = sum_r w[g][r] A[r][c]       M <-> 16 geometries,  N <-> 16 columns,    K <-> rows
// These are still HBM-bound streams (A is read once, 681 MB at H30/T=20); the matrix cores are
// used because the MFMA sums over K INSIDE the instruction: a lane keeps 4 accumulator doubles per
// 16x16 output tile instead of one private partial sum per (row, geometry), so 16 geometries cost
// no more registers than one and the per-geometry share of the stream drops to 1/16 (1/32 with two
// sets: the MFMAs of a pass are then ~80 us of pipe time per SIMD, still below the ~130 us the
// stream needs).
//
// Operand maps (cdna_hip_programming.md §3): A[i][k]: lane l holds i = l&15, k = l>>4;
// B[k][j]: k = l>>4, j = l&15; D[i][j]: j = l&15, i = (l>>4) + 4*reg.  The K slot of a lane can
// be ANY column as long as A and B agree, so both operands of K5 are fetched with the same
// coalesced 16-byte pattern: lane (l15,l4) loads columns c+8u+2*l4, +1 of "its" row (a t-RDM row
// for A, a geometry's vector for B) and feeds .x / .y to two MFMAs.
#include <stdlib.h>

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
// Block = (column span, row group of <= kRG 16-row tiles); the tiles of the matrix are dealt to
// BALANCED row groups (a group with one or two tiles costs a block almost as much as a full one: its
// chunk steps are latency bound either way).  The four waves interleave the 32-column chunks of the
// span; a wave keeps the V fragments of a chunk in registers for all tiles of the group and issues
// the loads of its NEXT chunk before the MFMAs of the current one (two register buffers).  The
// steady-state loop body is branch-free so that the compiler's vmcnt bookkeeping really leaves the
// younger loads in flight across the MFMAs (a conditional prefetch makes it wait for vmcnt(0)).
// Block -> (span, row group) is XCD-aware: workgroups are dealt round-robin to the 8 XCDs, each with
// its own L2, so the row groups of one span (which re-read the same V tile) get block ids that are
// congruent mod 8.  Partials: ws[g][span][row].
// Measured (tools/micro/rows_insitu.hip, 210 x 405450, G=16): 146 us with groups of <= 3 tiles,
// 165 us with <= 4, 160 us with <= 2; wider groups (5..7 tiles, alternating half-chunk buffers) ran
// out of registers at two waves per SIMD and were slower (200 us).
// Timing experiments (tools/micro/k5_stamps.py; build with EVC_DEBUG_STAMPS=1): every workgroup of the pipelined rows
// kernel stamps its entry, the end of its main loop and its exit.  Compiled out of the product library.
#ifdef EVC_DEBUG_STAMPS
__device__ long long g_k5_wg[1024 * 4];   // per workgroup: entry, main loop done, exit, hw id
#define EVC_K5_WG(i_)                                                                  \
    do {                                                                               \
        if (threadIdx.x == 0 && blockIdx.x < 1024) {                                   \
            long long t_;                                                              \
            asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory"); \
            g_k5_wg[4 * blockIdx.x + (i_)] = t_;                                       \
            if ((i_) == 0) {                                                           \
                unsigned hw_, xcc_;                                                    \
                asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw_));      \
                asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc_));    \
                g_k5_wg[4 * blockIdx.x + 3] = ((long long)(xcc_ & 0xF) << 32) | hw_;   \
            }                                                                          \
        }                                                                              \
    } while (0)
#else
#define EVC_K5_WG(i_) do { } while (0)
#endif
constexpr int kMC = 32;  // columns per wave chunk
constexpr int kRG = 4;   // most 16-row tiles per row group (register budget of two waves per SIMD)

#define EVC_LD_B(B_, C_)                                                                             \
    {                                                                                                \
        const int64_t cc_ = (C_) + 2 * l4;                                                           \
        _Pragma("unroll") for (int gs = 0; gs < GS; ++gs)                                            \
            _Pragma("unroll") for (int u = 0; u < 4; ++u) B_[gs][u] = ld2(vr[gs] + cc_ + 8 * u);     \
    }
#define EVC_LD_A(A_, N_, C_)                                                                         \
    {                                                                                                \
        const int64_t cc_ = (C_) + 2 * l4;                                                           \
        _Pragma("unroll") for (int tt = 0; tt < (N_); ++tt)                                          \
            _Pragma("unroll") for (int u = 0; u < 4; ++u) A_[tt][u] = ld2(ar[tt] + cc_ + 8 * u);     \
    }
#define EVC_MMA(A_, N_, B_)                                                                          \
    {                                                                                                \
        _Pragma("unroll") for (int u = 0; u < 4; ++u) {                                              \
            _Pragma("unroll") for (int gs = 0; gs < GS; ++gs)                                        \
                _Pragma("unroll") for (int tt = 0; tt < (N_); ++tt)                                  \
                    acc[gs][tt] = mfma_f64(A_[tt][u].x, B_[gs][u].x, acc[gs][tt]);                   \
            _Pragma("unroll") for (int gs = 0; gs < GS; ++gs)                                        \
                _Pragma("unroll") for (int tt = 0; tt < (N_); ++tt)                                  \
                    acc[gs][tt] = mfma_f64(A_[tt][u].y, B_[gs][u].y, acc[gs][tt]);                   \
        }                                                                                            \
    }

// One wave's share of a (span, row group) block with NT live 16-row tiles.  Lanes whose geometry slot
// is >= G read geometry g0's vector (a valid address) and their output columns are discarded.
template <int NT, int MAXT, int GS>
__device__ __forceinline__ void rows_pipe_body(const RowProblem &P, int64_t row_base, int64_t cbeg, int64_t cend,
                                               const double *__restrict__ const (&vr)[GS], int l15, int l4, int wave,
                                               d4 (&acc)[GS][MAXT]) {
    const int64_t rows = P.rows, cols = P.cols, ld = P.ld;
    const int64_t cfull = min(cend, cols & ~(int64_t)31);  // chunks starting below this are complete
    constexpr int64_t kStep = 4 * kMC;
    const double *__restrict__ ar[NT];
#pragma unroll
    for (int tt = 0; tt < NT; ++tt) ar[tt] = P.A + min(row_base + 16 * tt + l15, rows - 1) * ld;
    int64_t c = cbeg + wave * kMC;
    const int64_t nfull = c < cfull ? (cfull - c + kStep - 1) / kStep : 0;  // complete chunks of this wave
    double2 b0[GS][4], b1[GS][4];
    double2 a0[NT][4], a1[NT][4];
    if (nfull > 0) {
        EVC_LD_B(b0, c);
        EVC_LD_A(a0, NT, c);
        int64_t i = 1;
        for (; i + 1 < nfull; i += 2) {
            EVC_LD_B(b1, c + i * kStep);
            EVC_LD_A(a1, NT, c + i * kStep);
            EVC_MMA(a0, NT, b0);
            EVC_LD_B(b0, c + (i + 1) * kStep);
            EVC_LD_A(a0, NT, c + (i + 1) * kStep);
            EVC_MMA(a1, NT, b1);
        }
        if (i < nfull) {
            EVC_LD_B(b1, c + i * kStep);
            EVC_LD_A(a1, NT, c + i * kStep);
            EVC_MMA(a0, NT, b0);
            EVC_MMA(a1, NT, b1);
        } else {
            EVC_MMA(a0, NT, b0);
        }
        c += nfull * kStep;
    }
    if (c < cend) {  // the ragged last chunk of the matrix (at most one wave of one span per row group)
        const int64_t cc = c + 2 * l4;
#pragma unroll
        for (int gs = 0; gs < GS; ++gs)
#pragma unroll
            for (int u = 0; u < 4; ++u) b0[gs][u] = ld2_guard(vr[gs], cc + 8 * u, cols);
#pragma unroll
        for (int tt = 0; tt < NT; ++tt)
#pragma unroll
            for (int u = 0; u < 4; ++u) a0[tt][u] = ld2_guard(ar[tt], cc + 8 * u, cols);
        EVC_MMA(a0, NT, b0);
    }
}

// Lean variant of the same share: one register buffer, no prefetch — few registers, so many waves per
// SIMD hide the load latency instead.
template <int NT, int MAXT, int GS>
__device__ __forceinline__ void rows_lean_body(const RowProblem &P, int64_t row_base, int64_t cbeg, int64_t cend,
                                               const double *__restrict__ const (&vr)[GS], int l15, int l4, int wave,
                                               d4 (&acc)[GS][MAXT]) {
    const int64_t rows = P.rows, cols = P.cols, ld = P.ld;
    const int64_t cfull = min(cend, cols & ~(int64_t)31);
    constexpr int64_t kStep = 4 * kMC;
    const double *__restrict__ ar[NT];
#pragma unroll
    for (int tt = 0; tt < NT; ++tt) ar[tt] = P.A + min(row_base + 16 * tt + l15, rows - 1) * ld;
    double2 b0[GS][4], a0[NT][4];
    int64_t c = cbeg + wave * kMC;
    for (; c < cfull; c += kStep) {
        EVC_LD_B(b0, c);
        EVC_LD_A(a0, NT, c);
        EVC_MMA(a0, NT, b0);
    }
    if (c < cend) {
        const int64_t cc = c + 2 * l4;
#pragma unroll
        for (int gs = 0; gs < GS; ++gs)
#pragma unroll
            for (int u = 0; u < 4; ++u) b0[gs][u] = ld2_guard(vr[gs], cc + 8 * u, cols);
#pragma unroll
        for (int tt = 0; tt < NT; ++tt)
#pragma unroll
            for (int u = 0; u < 4; ++u) a0[tt][u] = ld2_guard(ar[tt], cc + 8 * u, cols);
        EVC_MMA(a0, NT, b0);
    }
}

// GS sets of 16 geometries [g0, g0 + G), G <= 16 GS; MAXT = most tiles per row group; MINW = waves per SIMD
// the register allocation must allow; PIPE selects the software-pipelined body.
template <int GS, int MAXT, int MINW, bool PIPE>
__global__ __launch_bounds__(256, MINW) void gemv_rows_mfma_pipe_kernel(GemvRowsLaunch L, int g0, int G) {
    __shared__ double red[4][MAXT][4][64];  // [wave][tile][reg][lane]
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
    EVC_K5_WG(0);
    const RowProblem &P = L.p[which];
    const int64_t rows = P.rows;
    const int tpg = L.tpg[which], trem = L.trem[which];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l15 = lane & 15, l4 = lane >> 4;
    const int64_t row_base = (int64_t)(rg * tpg + min(rg, trem)) * 16;
    const int ntile = tpg + (rg < trem ? 1 : 0);  // tiles of this row group (the last one may be ragged)
    const int64_t cbeg = (int64_t)span * P.span_cols;
    const int64_t cend = min(P.cols, (int64_t)(span + 1) * P.span_cols);
    const double *__restrict__ vr[GS];
#pragma unroll
    for (int gs = 0; gs < GS; ++gs) {
        const int slot = 16 * gs + l15;
        vr[gs] = P.v + (int64_t)(g0 + (slot < G ? slot : 0)) * P.vstride;
    }
    d4 acc[GS][MAXT];
#pragma unroll
    for (int gs = 0; gs < GS; ++gs)
#pragma unroll
        for (int t = 0; t < MAXT; ++t) acc[gs][t] = (d4){0.0, 0.0, 0.0, 0.0};
#define EVC_ROWS_BODY(NT_)                                                                          \
    {                                                                                               \
        if constexpr (PIPE) rows_pipe_body<NT_, MAXT, GS>(P, row_base, cbeg, cend, vr, l15, l4, wave, acc); \
        else rows_lean_body<NT_, MAXT, GS>(P, row_base, cbeg, cend, vr, l15, l4, wave, acc);        \
    }
    if constexpr (MAXT >= 7) {
        if (ntile == 7) EVC_ROWS_BODY(7)  // wave-uniform
    }
    if constexpr (MAXT >= 6) {
        if (ntile == 6) EVC_ROWS_BODY(6)
    }
    if constexpr (MAXT >= 5) {
        if (ntile == 5) EVC_ROWS_BODY(5)
    }
    if constexpr (MAXT >= 4) {
        if (ntile == 4) EVC_ROWS_BODY(4)  // wave-uniform
    }
    if constexpr (MAXT >= 3) {
        if (ntile == 3) EVC_ROWS_BODY(3)
    }
    if constexpr (MAXT >= 2) {
        if (ntile == 2) EVC_ROWS_BODY(2)
    }
    if (ntile == 1) EVC_ROWS_BODY(1)
#undef EVC_ROWS_BODY
    EVC_K5_WG(1);
    // cross-wave sum, one geometry set per pass
#pragma unroll
    for (int gs = 0; gs < GS; ++gs) {
        if (gs) __syncthreads();
#pragma unroll
        for (int tt = 0; tt < MAXT; ++tt)
#pragma unroll
            for (int r = 0; r < 4; ++r) red[wave][tt][r][lane] = acc[gs][tt][r];
        __syncthreads();
        for (int idx = tid; idx < MAXT * 4 * 64; idx += 256) {
            const int ln = idx & 63, r = (idx >> 6) & 3, tt = idx >> 8;
            const int64_t row = row_base + tt * 16 + (ln >> 4) + 4 * r;
            const int g = 16 * gs + (ln & 15);
            if (tt < ntile && row < rows && g < G) {
                const double s = (red[0][tt][r][ln] + red[1][tt][r][ln]) + (red[2][tt][r][ln] + red[3][tt][r][ln]);
                P.partial[(int64_t)(g0 + g) * P.pstride + (int64_t)span * rows + row] = s;
            }
        }
    }
    EVC_K5_WG(2);
}

int launch_gemv_rows_mfma(const GemvRowsLaunch &Lin, int g0, int G, int tiles, hipStream_t st) {
    // spans planned for the LDS-staged kernel (gemv_lds.hip: whole-line LDS-DMA instead of fragment-shaped loads)
    if (Lin.p[0].nblocks ? Lin.p[0].lds_plan : Lin.p[1].lds_plan) {
        if (Lin.p[0].nblocks && Lin.p[1].nblocks && Lin.p[1].lds_plan <= 0) {
            // a tall second problem (large training sets: T^2 rows) in its own launch: of the LDS-staged kernel as the
            // only problem (plan < 0), else of the fragment-shaped kernel
            GemvRowsLaunch La = Lin, Lb = Lin;
            La.p[1].nblocks = 0;
            if (int rc = launch_gemv_rows_lds(La, g0, G, st)) return rc;
            if (Lin.p[1].lds_plan < 0 && G <= 32) {
                Lb.p[0] = Lin.p[1];
                Lb.p[0].lds_plan = -Lin.p[1].lds_plan;
                Lb.p[1].nblocks = 0;
                return launch_gemv_rows_lds(Lb, g0, G, st);
            }
            Lb.p[0].nblocks = 0;
            Lb.p[0].lds_plan = 0;
            Lb.p[1].lds_plan = 0;
            return launch_gemv_rows_mfma(Lb, g0, G, tiles, st);
        }
        return launch_gemv_rows_lds(Lin, g0, G, st);
    }
    GemvRowsLaunch L = Lin;
    const int gs = G > 16 ? 2 : 1;
    // ONE shape per case, 100*MAXT + 10*MINW + PIPE (the alternatives of rounds 1-3 were measured slower and removed in
    // round 4 with their knobs): one geometry set 421 (pipelined, in situ at G=16: 189 us against 194-221 for the lean
    // shapes); two sets 321 on wide matrices (226 us against 238-300) and 711 on narrow ones (< 200 000 columns, the
    // 8-fold compressed layout: the vectors of the 32 geometries weigh as much as the matrix itself when five row groups
    // re-read them; two groups of seven tiles on ONE wave per SIMD read them twice: 66 against 76 us)
    const int shape = gs == 2 ? (Lin.p[0].cols <= 200000 ? 711 : 321) : 421;
    const int kernel_maxt = shape / 100;
    int max_tiles = tiles <= 0 ? (kernel_maxt >= 5 ? kernel_maxt : kernel_maxt >= 3 ? 3 : kernel_maxt) : tiles;   // see the measurements above
    if (max_tiles > kernel_maxt) max_tiles = kernel_maxt;
    for (int k = 0; k < 2; ++k) {
        // balanced row groups of at most max_tiles tiles
        const int nt = (int)ceil_div(L.p[k].rows > 0 ? L.p[k].rows : 1, 16);
        L.nrg[k] = (int)ceil_div(nt, max_tiles);
        L.tpg[k] = nt / L.nrg[k];
        L.trem[k] = nt % L.nrg[k];
        if (!L.p[k].nblocks) L.p[k].nspans = 0;
    }
    const int nb0 = L.p[0].nblocks ? 8 * L.nrg[0] * (int)ceil_div(L.p[0].nspans, 8) : 0;
    const int nb1 = L.p[1].nblocks ? (int)ceil_div((int64_t)L.nrg[1] * L.p[1].nspans, 8) * 8 : 0;
    L.nblk0 = nb0;
    L.nblk1 = nb1;
#define EVC_ROWS_LAUNCH(GS_, MAXT_, MINW_, PIPE_)                                                      \
    do {                                                                                               \
        hipLaunchKernelGGL((gemv_rows_mfma_pipe_kernel<GS_, MAXT_, MINW_, PIPE_ != 0>), dim3(nb0 + nb1), \
                           dim3(256), 0, st, L, g0, G);                                                \
        note_kernel(EVC_PROF_ROWS, "gemv_rows_mfma_pipe_kernel<%d,%d,%d,%d> G=%d", GS_, MAXT_, MINW_, PIPE_, G); \
    } while (0)
    if (shape == 711) EVC_ROWS_LAUNCH(2, 7, 1, 1);
    else if (shape == 321) EVC_ROWS_LAUNCH(2, 3, 2, 1);
    else EVC_ROWS_LAUNCH(1, 4, 2, 1);
#undef EVC_ROWS_LAUNCH
    EVC_LAUNCH_CHECK("gemv_rows_mfma");
    return 0;
}

// ------------------------------------------------------------------ K8: cols GEMM
// Wave = 32*CT columns (CT even/odd tile pairs); block = 4 waves = 128*CT columns; the wave walks
// down the rows 4 at a time (one K step), KSN K steps per iteration so that KSN*CT 16-byte loads are
// in flight per lane.  The weights of a row tile (<= 256 rows) are staged in LDS as wl[row][16 GS].
// Measured at H30/T=20, G=16 (in situ): 96-column waves 196 us, 128-column 205 us, 160-column 222 us,
// 64-column 202 us; fitting all blocks into one resident round did not help.
constexpr int kRowTile = 128;

template <int CT, int KSN, int MINW, int GS>
__global__ __launch_bounds__(256, MINW) void gemv_cols_mfma_kernel(GemvColsLaunch L, int g0, int G) {
    extern __shared__ __align__(16) double wl[];  // min(rows, kRowTile) (rounded up to 4) x 16 GS
    constexpr int GW = 16 * GS;
    int bid = gridDim.x - 1 - blockIdx.x;  // the few blocks of the small second problem are dispatched first
    const int which = bid >= L.nblk0 ? 1 : 0;
    if (which) bid -= L.nblk0;
    const ColProblem &P = L.p[which];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l15 = lane & 15, l4 = lane >> 4;
    const int64_t rows = P.rows, cols = P.cols, ld = P.ld;
    const int64_t c0 = ((int64_t)bid * 4 + wave) * (32 * CT);   // first column of this wave
    const double *__restrict__ w = P.w + (int64_t)g0 * P.wstride;

    d4 ae[GS][CT], ao[GS][CT];
#pragma unroll
    for (int gs = 0; gs < GS; ++gs)
#pragma unroll
        for (int t = 0; t < CT; ++t) {
            ae[gs][t] = (d4){0.0, 0.0, 0.0, 0.0};
            ao[gs][t] = (d4){0.0, 0.0, 0.0, 0.0};
        }
    // columns of this lane in tile pair t: c0 + 32 t + 2 l15, +1
    const int64_t cl = c0 + 2 * l15;
    const bool inside = c0 + 32 * CT <= cols;   // no column of this wave needs a guard

    for (int64_t r0 = 0; r0 < rows; r0 += kRowTile) {
        const int nr = (int)min((int64_t)kRowTile, rows - r0);
        const int nr4 = (nr + 3) & ~3;
        __syncthreads();
        // consecutive lanes read consecutive rows of ONE geometry's weight vector (the vectors of different
        // geometries lie a workspace apart: the transposed order costs one 64-byte request per element)
        for (int idx = tid; idx < nr4 * GW; idx += 256) {
            const int g = idx / nr4, r = idx - g * nr4;
            wl[r * GW + g] = (r < nr && g < G) ? w[(int64_t)g * P.wstride + r0 + r] : 0.0;
        }
        __syncthreads();
        if (c0 < cols) {
            for (int rb = 0; rb < nr4; rb += 4 * KSN) {
                double2 x[KSN][CT];
                double wf[KSN][GS];
#pragma unroll
                for (int ks = 0; ks < KSN; ++ks) {
                    const int r = rb + 4 * ks + l4;
                    // rows beyond the tile carry zero weights; clamp the address to a valid row
                    const int64_t rr = r0 + (r < nr ? r : nr - 1);
                    const double *row = P.A + rr * ld;
                    const bool live = rb + 4 * ks < nr4;
#pragma unroll
                    for (int gs = 0; gs < GS; ++gs)
                        wf[ks][gs] = live ? wl[(rb + 4 * ks + l4) * GW + 16 * gs + l15] : 0.0;
#pragma unroll
                    for (int t = 0; t < CT; ++t) {
                        const int64_t c = cl + 32 * t;
                        if (inside) x[ks][t] = live ? ld2(row + c) : make_double2(0.0, 0.0);   // wave-uniform
                        else x[ks][t] = live ? ld2_guard(row, c, cols) : make_double2(0.0, 0.0);
                    }
                }
#pragma unroll
                for (int ks = 0; ks < KSN; ++ks)
#pragma unroll
                    for (int gs = 0; gs < GS; ++gs)
#pragma unroll
                        for (int t = 0; t < CT; ++t) {
                            ae[gs][t] = mfma_f64(wf[ks][gs], x[ks][t].x, ae[gs][t]);
                            ao[gs][t] = mfma_f64(wf[ks][gs], x[ks][t].y, ao[gs][t]);
                        }
            }
        }
    }
    if (c0 < cols) {
#pragma unroll
        for (int gs = 0; gs < GS; ++gs)
#pragma unroll
            for (int t = 0; t < CT; ++t) {
                const int64_t c = cl + 32 * t;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int g = 16 * gs + l4 + 4 * r;
                    if (g < G && c < cols) {
                        double *o = P.out + (int64_t)(g0 + g) * P.ostride + c;
                        if (c + 1 < cols) *reinterpret_cast<double2 *>(o) = make_double2(ae[gs][t][r], ao[gs][t][r]);
                        else *o = ae[gs][t][r];
                    }
                }
            }
    }
}

// Row-split variant for matrices with few columns (the 8-fold compressed layout: 108 345 columns at N = 30, 3.7x
// fewer waves than the kernel above gets from the packed layout): a block owns ONE 32-column tile and its four
// waves take the K steps in turn, so a wave's dependent chain is a quarter as long; the four partial tiles are
// summed in fixed order through LDS.  The weights are read straight from memory (a wave needs only those of its
// own K steps: 4 rows x 16 geometries per MFMA operand, L2 resident), so there is no weight tile to fill and no
// barrier inside the stream.
template <int KSN, int MINW, int GS>
__global__ __launch_bounds__(256, MINW) void gemv_cols_mfma_rs_kernel(GemvColsLaunch L, int g0, int G) {
    __shared__ double red[4 * GS * 2 * 4 * 64];  // [wave][gs][eo][reg][lane]
    int bid = gridDim.x - 1 - blockIdx.x;
    const int which = bid >= L.nblk0 ? 1 : 0;
    if (which) bid -= L.nblk0;
    const ColProblem &P = L.p[which];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l15 = lane & 15, l4 = lane >> 4;
    const int64_t rows = P.rows, cols = P.cols, ld = P.ld;
    const int64_t c0 = (int64_t)bid * 32;
    // lanes whose geometry slot is >= G read geometry g0's weights; their output rows are discarded.
    // With a transposed copy (P.wt: [row][kMaxBatchG] per group of geometries) the 16 geometries of a K step are
    // 128 contiguous bytes per row instead of 16 separate 32-byte segments.
    const double *__restrict__ wg[GS];
    const bool wtr = P.wt != nullptr;
    const int64_t wrs = wtr ? kMaxBatchG : 1;   // stride between rows
#pragma unroll
    for (int gs = 0; gs < GS; ++gs) {
        const int slot = 16 * gs + l15, gg = g0 + (slot < G ? slot : 0);
        wg[gs] = wtr ? P.wt + (int64_t)(gg - gg % kMaxBatchG) * P.wstride + gg % kMaxBatchG
                     : P.w + (int64_t)gg * P.wstride;
    }
    d4 ae[GS], ao[GS];
#pragma unroll
    for (int gs = 0; gs < GS; ++gs) {
        ae[gs] = (d4){0.0, 0.0, 0.0, 0.0};
        ao[gs] = (d4){0.0, 0.0, 0.0, 0.0};
    }
    const int64_t cl = c0 + 2 * l15;
    const bool inside = c0 + 32 <= cols;
    for (int64_t rb = (int64_t)wave * 4 * KSN; rb < rows; rb += 16 * KSN) {
        double2 x[KSN];
        double wf[KSN][GS];
#pragma unroll
        for (int ks = 0; ks < KSN; ++ks) {
            const int64_t r = rb + 4 * ks + l4;
            const bool live = r < rows;   // rows beyond the matrix: zero weight, address clamped to a valid row
            const double *row = P.A + (live ? r : rows - 1) * ld;
#pragma unroll
            for (int gs = 0; gs < GS; ++gs) wf[ks][gs] = live ? wg[gs][r * wrs] : 0.0;
            if (inside) x[ks] = ld2(row + cl);
            else x[ks] = ld2_guard(row, cl, cols);
        }
#pragma unroll
        for (int ks = 0; ks < KSN; ++ks)
#pragma unroll
            for (int gs = 0; gs < GS; ++gs) {
                ae[gs] = mfma_f64(wf[ks][gs], x[ks].x, ae[gs]);
                ao[gs] = mfma_f64(wf[ks][gs], x[ks].y, ao[gs]);
            }
    }
#pragma unroll
    for (int gs = 0; gs < GS; ++gs)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            red[(((wave * GS + gs) * 2 + 0) * 4 + r) * 64 + lane] = ae[gs][r];
            red[(((wave * GS + gs) * 2 + 1) * 4 + r) * 64 + lane] = ao[gs][r];
        }
    __syncthreads();
    for (int idx = tid; idx < GS * 4 * 64; idx += 256) {
        const int ln = idx & 63, r = (idx >> 6) & 3, gs = idx >> 8;
        const int g = 16 * gs + (ln >> 4) + 4 * r;
        const int64_t c = c0 + 2 * (ln & 15);
        if (g < G && c < cols) {
            double v[2];
#pragma unroll
            for (int eo = 0; eo < 2; ++eo) {
                const int o = ((gs * 2 + eo) * 4 + r) * 64 + ln;
                constexpr int WS = GS * 2 * 4 * 64;  // one wave's slice
                v[eo] = (red[o] + red[WS + o]) + (red[2 * WS + o] + red[3 * WS + o]);
            }
            double *o = P.out + (int64_t)(g0 + g) * P.ostride + c;
            if (c + 1 < cols) *reinterpret_cast<double2 *>(o) = make_double2(v[0], v[1]);
            else *o = v[0];
        }
    }
}

template <int KSN, int MINW, int GS>
static void cols_mfma_rs_launch(GemvColsLaunch L, int g0, int G, hipStream_t st) {
    L.nblk0 = (int)ceil_div(L.p[0].cols, 32);
    const int total = L.nblk0 + (int)ceil_div(L.p[1].cols, 32);
    hipLaunchKernelGGL((gemv_cols_mfma_rs_kernel<KSN, MINW, GS>), dim3(total), dim3(256), 0, st, L, g0, G);
    note_kernel(EVC_PROF_COLS, "gemv_cols_mfma_rs_kernel<%d,%d,%d>", KSN, MINW, GS);
}

template <int CT, int KSN, int MINW, int GS>
static int cols_mfma_launch(GemvColsLaunch L, int g0, int G, hipStream_t st) {
    const int64_t per = 4 * 32 * CT;
    L.nblk0 = (int)ceil_div(L.p[0].cols, per);
    const int total = L.nblk0 + (int)ceil_div(L.p[1].cols, per);
    int64_t rmax = L.p[0].rows > L.p[1].rows ? L.p[0].rows : L.p[1].rows;
    if (rmax > kRowTile) rmax = kRowTile;
    const size_t lds = sizeof(double) * 16 * GS * (size_t)((rmax + 3) & ~3);
    static LdsAttr attr;
    if (int rc = allow_dynamic_lds(gemv_cols_mfma_kernel<CT, KSN, MINW, GS>, attr,
                                   kRowTile * 16 * GS * (int)sizeof(double), "gemv_cols_mfma"))
        return rc;
    hipLaunchKernelGGL((gemv_cols_mfma_kernel<CT, KSN, MINW, GS>), dim3(total), dim3(256), lds, st, L, g0, G);
    note_kernel(EVC_PROF_COLS, "gemv_cols_mfma_kernel<%d,%d,%d,%d>", CT, KSN, MINW, GS);
    return 0;
}

int launch_gemv_cols_mfma(GemvColsLaunch L, int g0, int G, hipStream_t st) {
    if (L.p[0].cols + L.p[1].cols == 0) return 0;
    // whole-line LDS-DMA pieces, a wave sums over all rows of its column tiles (gemv_lds.hip)
    const int lds_mode = cols_lds_mode(L.p[0], L.p[1], G);
    if (lds_mode == 1) return launch_gemv_cols_lds(L, g0, G, st);
    if (lds_mode == 2 && g0 % kMaxBatchG == 0) return launch_gemv_cols_lds_slab(L, g0, G, st);
    // ONE shape per case (in situ at H30 / T = 20, us per launch: lean shapes with many waves per SIMD won by a wide
    // margin over wide register-blocked ones -- G=16: 126 -> 138, 224 -> 147, 343 (the first design) -> 193; G=32:
    // 224 -> 182, 214 -> 197, 342 -> 312; the alternatives were removed in round 4 with their knobs): the row-split
    // kernel below 200 000 columns (K steps 2 / 8, waves per SIMD 6 / 3 for one / two geometry sets), the column-tiled
    // kernel beyond
    if (L.p[0].cols <= 200000) {
        if (G > 16) cols_mfma_rs_launch<8, 3, 2>(L, g0, G, st);
        else cols_mfma_rs_launch<2, 6, 1>(L, g0, G, st);
        EVC_LAUNCH_CHECK("gemv_cols_mfma_rs");
        return 0;
    }
    if (G > 16) {
        if (int rc_ = cols_mfma_launch<2, 2, 4, 2>(L, g0, G, st)) return rc_;
    } else {
        if (int rc_ = cols_mfma_launch<1, 2, 6, 1>(L, g0, G, st)) return rc_;
    }
    EVC_LAUNCH_CHECK("gemv_cols_mfma");
    return 0;
}

}  // namespace evc

#ifdef EVC_DEBUG_STAMPS
extern "C" int evc_debug_read_k5(long long *stamps) {
    return (int)hipMemcpyFromSymbol(stamps, HIP_SYMBOL(evc::g_k5_wg), sizeof(long long) * 1024 * 4);
}
#endif
