// Four-index basis rotations on the FP64 matrix cores.
//   K3/K14  quarter transform (v_mfma_f64_16x16x4_f64)          electron_integral_utils.py:136,
//                                                              gradients_loewdin.py:224-232,339
//   pair transform of the compressed pipeline (pt_kernel and its pipelined forms: every N <= 32 the LDS-DMA kernel of
//   pair_dma.hip does not take)
// The N^4-sized helpers around them: pack.hip (pack / unpack), y2.hip (K13), ip1.hip (K15).
// blockIdx.y = geometry of the batch (kernels.hpp).
#include <stdlib.h>

#include "common.hpp"
#include "kernels.hpp"
#include <type_traits>

namespace evc {

typedef double d4 __attribute__((ext_vector_type(4)));
typedef double d2 __attribute__((ext_vector_type(2)));

// v_mfma_f64_16x16x4_f64 operand maps (cdna_hip_programming.md §3):
//   A[i][k]: lane l holds i = l&15, k = l>>4        B[k][j]: k = l>>4, j = l&15
//   D[i][j]: j = l&15, i = (l>>4) + 4*reg
__device__ __forceinline__ d4 mfma_f64(double a, double b, d4 c) {
    return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
}

// ------------------------------------------------------------------ quarter transform
// out[q][row] = sum_d in[row][d] * C[d][q],  row = (a,b,c) flattened, rows = n^3.
// MFMA roles: M <-> q (rotated index, C from LDS), N <-> row (16 tensor rows per wave step),
// K <-> d.  Rows of `in` are n contiguous doubles, so the 16 rows a wave consumes form one
// contiguous 16*n*8-byte block; the output is written as 128-byte row segments.
// The first tile's operand loads are issued before the LDS fill of C so both latencies overlap.
template <int NPAD>
__global__ __launch_bounds__(256) void qt_kernel(const double *__restrict__ in, int64_t sin,
                                                 const double *__restrict__ C, int64_t sC, int ct, int n,
                                                 int64_t rows, double *__restrict__ out, int64_t sout) {
    constexpr int LDX = (NPAD % 32 == 0) ? NPAD + 16 : NPAD;  // keeps the two 16-lane halves on disjoint banks
    constexpr int KSTEPS = NPAD / 4;
    constexpr int NT = NPAD / 16;
    __shared__ double Xs[NPAD * LDX];
    in += (int64_t)blockIdx.y * sin;
    C += (int64_t)blockIdx.y * sC;
    out += (int64_t)blockIdx.y * sout;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int l15 = lane & 15, l4 = lane >> 4;
    const int64_t ntiles = (rows + 15) / 16;
    int64_t tile = (int64_t)blockIdx.x * 4 + wave;
    double b[KSTEPS];
    {
        const int64_t row = tile * 16 + l15;
        const bool rok = tile < ntiles && row < rows;
#pragma unroll
        for (int kk = 0; kk < KSTEPS; ++kk) {
            const int d = 4 * kk + l4;
            b[kk] = (rok && d < n) ? in[row * n + d] : 0.0;
        }
    }
    for (int idx = threadIdx.x; idx < NPAD * NPAD; idx += 256) {
        const int d = idx / NPAD, q = idx % NPAD;
        double v = 0.0;
        if (d < n && q < n) v = ct ? C[q * n + d] : C[d * n + q];
        Xs[d * LDX + q] = v;
    }
    __syncthreads();
    while (tile < ntiles) {
        const int64_t row = tile * 16 + l15;
        const bool rok = row < rows;
        const int64_t next = tile + (int64_t)gridDim.x * 4;
        double bn[KSTEPS];
        {
            const int64_t nrow = next * 16 + l15;
            const bool nok = next < ntiles && nrow < rows;
#pragma unroll
            for (int kk = 0; kk < KSTEPS; ++kk) {
                const int d = 4 * kk + l4;
                bn[kk] = (nok && d < n) ? in[nrow * n + d] : 0.0;
            }
        }
        // the NT output tiles are independent accumulation chains: interleave them (a dependent
        // f64 MFMA costs ~3x the issue interval)
        d4 acc[NT];
#pragma unroll
        for (int t = 0; t < NT; ++t) acc[t] = (d4){0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int kk = 0; kk < KSTEPS; ++kk)
#pragma unroll
            for (int t = 0; t < NT; ++t)
                acc[t] = mfma_f64(Xs[(4 * kk + l4) * LDX + t * 16 + l15], b[kk], acc[t]);
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int q = t * 16 + l4 + 4 * r;
                if (rok && q < n) out[(int64_t)q * rows + row] = acc[t][r];
            }
#pragma unroll
        for (int kk = 0; kk < KSTEPS; ++kk) b[kk] = bn[kk];
        tile = next;
    }
}

// Grid: every workgroup stages X (up to 41 KB of LDS) before its first tile, so the launch wants PERSISTENT workgroups --
// one resident round of the chip (the register budget of these kernels allows one wave per SIMD from 64 padded columns
// on, two at 48, four below), each walking many tiles with the next tile's rows in flight -- not one workgroup per four
// tiles (measured per quarter step, one geometry: n = 58 93 -> 48 us, n = 64 133 -> 66 us; a two-tile variant with
// 16-byte operand loads added nothing on top and was dropped).
template <int NPAD>
static int qt_launch(const double *in, int64_t sin, const double *C, int64_t sC, int ct, int n, double *out,
                     int64_t sout, int count, hipStream_t st) {
    const int64_t rows = (int64_t)n * n * n;
    constexpr int kResident = 256 * (NPAD >= 64 ? 1 : NPAD >= 48 ? 2 : 4);   // workgroups of one round
    const int64_t share = (kResident + count - 1) / count;                   // per geometry of the batch
    const int64_t ntiles = (rows + 15) / 16;
    int64_t blocks = (ntiles + 3) / 4;
    if (blocks > share) blocks = share;
    hipLaunchKernelGGL(qt_kernel<NPAD>, dim3((unsigned)blocks, (unsigned)count), dim3(256), 0, st, in, sin, C, sC, ct,
                       n, rows, out, sout);
    EVC_LAUNCH_CHECK("quarter_transform");
    return 0;
}

int launch_quarter_transform(const double *in, int64_t sin, const double *C, int64_t sC, int ct, int n, double *out,
                             int64_t sout, int count, hipStream_t st) {
    const int npad = (n + 15) / 16 * 16;
    switch (npad) {
        case 16: return qt_launch<16>(in, sin, C, sC, ct, n, out, sout, count, st);
        case 32: return qt_launch<32>(in, sin, C, sC, ct, n, out, sout, count, st);
        case 48: return qt_launch<48>(in, sin, C, sC, ct, n, out, sout, count, st);
        case 64: return qt_launch<64>(in, sin, C, sC, ct, n, out, sout, count, st);
        case 80: return qt_launch<80>(in, sin, C, sC, ct, n, out, sout, count, st);
        case 96: return qt_launch<96>(in, sin, C, sC, ct, n, out, sout, count, st);
        default: break;
    }
    set_error("quarter_transform: n=%d not supported (1..96)", n);
    return -1;
}

// ------------------------------------------------------------------ fused pair transform (two quarter steps)
// For every leading index pair (p,q) the n x n block M = in[p][q][:,:] is rotated on both sides,
//     H = M X            (half result, optionally stored as K3[s'][p][q][r] = H[r][s'])
//     N = X^T H          stored as out[r'][s'][p][q]  (= two quarter steps of the rotation scheme)
// or, for the final step of the AO->OAO integral rotation, directly into the packed lower triangle
// (row (r',s') >= column (p,q), diagonal x diag_mult; electron_integral_utils.py:38-66).
// One wave per matrix: the D tiles of H = M X sit in exactly the lanes/registers the B operand of
// X^T H needs (row 4*kk + (l>>4) of H lives in register kk%4 of row-tile kk/4), so the second product
// consumes the accumulators of the first without any data movement; the X fragments serve as B
// operand of the first and A operand of the second product.  A workgroup works through tiles of 8
// leading pairs (2 matrices per wave) and stages N in LDS so that the output leaves as 64-byte runs.
// Symmetric operands (the compressed layout's pipeline) are exploited through three
// independent flags: lead_sym (in[p][q] = in[q][p]: only the n(n+1)/2 pairs q <= p are computed), in_lower
// (M = M^T: only its lower triangle is read) and rs_lower (only N[r'][s'], s' <= r', is wanted: 3 of 4
// result tiles, half the stage, half the output rows / the 8-fold compressed packed vector).
// ROWBUF (operand = dense (pair, pair) matrix, in_pairs): a leading pair's matrix is ONE contiguous row of n(n+1)/2
// doubles.  The wave fetches it with coalesced 16-byte loads (4-5 instructions of 1 KiB instead of 16 eight-byte
// gathers that touch 64 cache lines each), parks it in a wave-private LDS row and reads its MFMA fragments from there
// at lane-constant offsets tri(max(r,s), min(r,s)): triangular numbers of 16 consecutive r fall on 16 different banks.

// Timing experiments (tools/micro/pt_stamps.py; build with EVC_DEBUG_STAMPS=1): the four waves of one workgroup in the
// middle of the grid stamp their phases with the 100 MHz wall clock.  Compiled out of the product library.
#ifdef EVC_DEBUG_STAMPS
__device__ long long g_pt_stamp[4 * 64];
__device__ long long g_pt_wg[4096 * 3];   // per workgroup: entry, exit, (XCC_ID << 8 | CU/SE id register)
#define EVC_PT_WG(i_)                                                                                          \
    do {                                                                                                       \
        const int L_ = blockIdx.y * gridDim.x + blockIdx.x;                                                    \
        if (threadIdx.x == 0 && L_ < 4096) {                                                                   \
            g_pt_wg[3 * L_ + (i_)] = wall_clock64();                                                           \
            if ((i_) == 0) {                                                                                   \
                unsigned hw_, xcc_;                                                                            \
                asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw_));                              \
                asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc_));                            \
                g_pt_wg[3 * L_ + 2] = ((long long)(xcc_ & 0xF) << 32) | hw_;                                   \
            }                                                                                                  \
        }                                                                                                      \
    } while (0)
#define EVC_PT_STAMP(i_)                                                                                       \
    do {                                                                                                       \
        __builtin_amdgcn_sched_barrier(0);                                                                     \
        long long t_;                                                                                          \
        asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");                         \
        if (blockIdx.x == gridDim.x / 2 && blockIdx.y == gridDim.y / 2 && (threadIdx.x & 63) == 0 && (i_) < 64) \
            g_pt_stamp[(threadIdx.x >> 6) * 64 + (i_)] = t_;                                                   \
        __builtin_amdgcn_sched_barrier(0);                                                                     \
    } while (0)
#else
#define EVC_PT_STAMP(i_) do { } while (0)
#define EVC_PT_WG(i_) do { } while (0)
#endif

template <int NPAD, bool ROWBUF>
__global__ __launch_bounds__(256) void pt_kernel(PairTransformArgs a) {
    constexpr int LDX = (NPAD % 32 == 0) ? NPAD + 16 : NPAD;
    constexpr int KS = NPAD / 4;
    constexpr int NT = NPAD / 16;
    constexpr int QT = 8, QP = QT + 1;
    extern __shared__ __align__(16) double sm[];
    double *Xs = sm;                  // NPAD * LDX
    double *stage = Xs + NPAD * LDX;  // n*n*QP, or n(n+1)/2*QP with rs_lower
    const int n = a.n;
    const int64_t n2 = (int64_t)n * n;
    const int64_t g = blockIdx.y;
    const double *__restrict__ in = a.in + g * a.sin;
    const double *__restrict__ C = a.C + g * a.sC;
    // The leading pairs (p,q) are handled in tiles of 8 (one 64-byte run of the output per (r',s')); a workgroup
    // owns `tpw` consecutive tiles: the stores of a tile drain while the next one is loaded and multiplied, and
    // X is staged once.
    //   plain:    tile t = (p, 8-wide q tile), pairs with q >= n are idle;
    //   lead_sym: in[p][q] = in[q][p] and the consumer reads out[..][p][q] only for q <= p (in_lower of the next
    //             step): the tiles run over the n(n+1)/2 pairs q <= p in row-major triangle order.
    const bool sym = a.lead_sym != 0;
    const int ntq = (n + QT - 1) / QT;
    const int npairs = n * (n + 1) / 2;
    const int ild = a.in_ld ? a.in_ld : npairs, old_ = a.out_ld ? a.out_ld : npairs;   // pitch of the dense (pair, pair) forms
    const int ntiles = sym ? (npairs + QT - 1) / QT : n * ntq;
    const int t_begin = blockIdx.x * a.tiles_per_wg, t_end = min(ntiles, t_begin + a.tiles_per_wg);
    if (t_begin >= t_end) return;
    EVC_PT_STAMP(0);
    EVC_PT_WG(0);
    const bool lower = a.in_lower != 0;  // the n x n matrices are symmetric and valid for r >= s only
    // rs_lower: the consumer needs the result N[r'][s'] for s' <= r' only: upper tiles of X^T H are not computed,
    // the stage holds the lower triangle (row index tri(r',s')) and only those rows of `out` are written
    const bool rsl = a.rs_lower != 0;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int l15 = lane & 15, l4 = lane >> 4;
    // pair `ql` of tile t -> (p, q); false if the slot is idle
    auto pair_of = [&](int t, int ql, int &p, int &q) -> bool {
        if (sym) {
            const int e = t * QT + ql;
            if (e >= npairs) return false;
            p = (int)tri_row(e);
            q = e - p * (p + 1) / 2;
            return true;
        }
        p = t / ntq;
        q = (t - p * ntq) * QT + ql;
        return q < n;
    };
    // in_pairs: the operand is the dense (pair, pair) matrix in[tri(p,q)][tri(r,s)] (out_pairs of the previous step)
    const bool inp = a.in_pairs != 0;
    auto load_matrix = [&](double (&m)[NT][KS], bool ok, int p, int q) {
        const double *Mb = inp ? in + (int64_t)(ok ? p * (p + 1) / 2 + q : 0) * ild
                               : in + ((int64_t)(ok ? p : 0) * n + (ok ? q : 0)) * n2;
#pragma unroll
        for (int rt = 0; rt < NT; ++rt)
#pragma unroll
            for (int kk = 0; kk < KS; ++kk) {
                const int r = rt * 16 + l15, s = 4 * kk + l4;
                const int hi = s > r ? s : r, lo = s > r ? r : s;
                const int off = inp ? hi * (hi + 1) / 2 + lo : ((lower && s > r) ? s * n + r : r * n + s);
                m[rt][kk] = (ok && r < n && s < n) ? Mb[off] : 0.0;
            }
    };

    // ROWBUF: wave-private LDS row behind the stage; raw[] = the row in flight (lane i holds doubles 128 u + 2 i, +1 of
    // the 16-byte aligned window that starts `dlt` doubles before the row)
    const int stage_rows = rsl ? npairs : n * n;
    double *mrow = stage + (((int64_t)stage_rows * QP + 1) & ~(int64_t)1) + wave * kPtRowLen;
    const int nraw = ROWBUF ? (npairs + 1 + 127) / 128 : 0;   // wave-uniform
    double2 raw[ROWBUF ? kPtRawMax : 1];
    int dlt = 0, dlt_next = 0;
    int foff[ROWBUF ? NT : 1][ROWBUF ? KS : 1];
    if constexpr (ROWBUF) {
#pragma unroll
        for (int rt = 0; rt < NT; ++rt)
#pragma unroll
            for (int kk = 0; kk < KS; ++kk) {
                const int r = rt * 16 + l15, s = 4 * kk + l4;
                const int hi = s > r ? s : r, lo = s > r ? r : s;
                foff[rt][kk] = (r < n && s < n) ? hi * (hi + 1) / 2 + lo : kPtRawMax * 128;   // else: a zero slot
            }
        if (lane < 4) mrow[kPtRawMax * 128 + lane] = 0.0;
    }
    auto fetch_row = [&](bool ok, int p_, int q_, int &d_) {
        if constexpr (ROWBUF) {
            const double *row = in + (int64_t)(ok ? p_ * (p_ + 1) / 2 + q_ : 0) * ild;
            d_ = (int)((reinterpret_cast<uintptr_t>(row) >> 3) & 1);
            const double *w0 = row - d_;          // 16-byte aligned window
            const int lim = npairs + d_;          // valid doubles of the window
#pragma unroll
            for (int u = 0; u < kPtRawMax; ++u) {
                if (u < nraw) {
                    const int j = 128 * u + 2 * lane;
                    double2 v = make_double2(0.0, 0.0);
                    if (ok) {
                        if (j + 1 < lim) v = *reinterpret_cast<const double2 *>(w0 + j);
                        else if (j < lim) v.x = w0[j];
                    }
                    raw[u] = v;
                }
            }
        }
    };
    auto row_to_fragments = [&](double (&m)[NT][KS], int d_) {
        if constexpr (ROWBUF) {
#pragma unroll
            for (int u = 0; u < kPtRawMax; ++u)
                if (u < nraw) *reinterpret_cast<double2 *>(mrow + 128 * u + 2 * lane) = raw[u];
#pragma unroll
            for (int rt = 0; rt < NT; ++rt)
#pragma unroll
                for (int kk = 0; kk < KS; ++kk) m[rt][kk] = mrow[foff[rt][kk] + d_];
        }
    };

    // first matrix of this wave: operand loads issued before the LDS fill of X
    double mf[NT][KS];
    int p = 0, q = 0;
    bool have = pair_of(t_begin, wave, p, q);
    if constexpr (ROWBUF) fetch_row(have, p, q, dlt);
    else load_matrix(mf, have, p, q);
    for (int idx = threadIdx.x; idx < NPAD * NPAD; idx += 256) {
        const int d = idx / NPAD, c = idx % NPAD;
        double v = 0.0;
        if (d < n && c < n) v = a.ct ? C[c * n + d] : C[d * n + c];
        Xs[d * LDX + c] = v;
    }
    lds_barrier();
    EVC_PT_STAMP(1);
    double xf[KS][NT];
#pragma unroll
    for (int kk = 0; kk < KS; ++kk)
#pragma unroll
        for (int t = 0; t < NT; ++t) xf[kk][t] = Xs[(4 * kk + l4) * LDX + t * 16 + l15];
    if constexpr (ROWBUF) row_to_fragments(mf, dlt);
    EVC_PT_STAMP(2);

    for (int t = t_begin; t < t_end; ++t) {
        [[maybe_unused]] const int sb = 3 + 8 * (t - t_begin);
        for (int ql = wave; ql < QT; ql += 4) {
            // prefetch the wave's next matrix: slot ql + 4 of this tile, else its first slot of the next tile
            double mn[ROWBUF ? 1 : NT][ROWBUF ? 1 : KS];
            int pn = 0, qn = 0;
            const bool have_next = (ql + 4 < QT) ? pair_of(t, ql + 4, pn, qn) : (t + 1 < t_end && pair_of(t + 1, wave, pn, qn));
            [[maybe_unused]] const int fb = (t == t_begin + 1) ? 44 + (ql >= 4 ? 5 : 0) : 64;
            EVC_PT_STAMP(fb);
            if constexpr (ROWBUF) fetch_row(have_next, pn, qn, dlt_next);
            else load_matrix(mn, have_next, pn, qn);
            EVC_PT_STAMP(fb + 1);
            if (have) {  // wave-uniform
                // H = M X
                // (a dependent f64 MFMA costs ~3x the issue interval: the NT*NT tile chains are interleaved)
                d4 h[NT][NT];
#pragma unroll
                for (int rt = 0; rt < NT; ++rt)
#pragma unroll
                    for (int st = 0; st < NT; ++st) h[rt][st] = (d4){0.0, 0.0, 0.0, 0.0};
#pragma unroll
                for (int kk = 0; kk < KS; ++kk)
#pragma unroll
                    for (int rt = 0; rt < NT; ++rt)
#pragma unroll
                        for (int st = 0; st < NT; ++st) h[rt][st] = mfma_f64(mf[rt][kk], xf[kk][st], h[rt][st]);
                EVC_PT_STAMP(fb + 2);
                if (a.k3) {
                    // lead_sym: only K3[s'][p][q][:] with q <= p is written (its consumer folds the p <-> q symmetry),
                    // as K3[s'][tri(p,q)][:] times the multiplicity of (p,q)
                    double *K3 = a.k3 + g * a.sk3;
                    const double km = (sym && p != q) ? 2.0 : 1.0;
                    const int64_t kcol = sym ? (int64_t)(p * (p + 1) / 2 + q) * n : (int64_t)p * n2 + (int64_t)q * n;
                    const int64_t krow = sym ? (int64_t)npairs * n : (int64_t)n * n2;
#pragma unroll
                    for (int rt = 0; rt < NT; ++rt)
#pragma unroll
                        for (int st = 0; st < NT; ++st)
#pragma unroll
                            for (int reg = 0; reg < 4; ++reg) {
                                const int r = rt * 16 + l4 + 4 * reg, s2 = st * 16 + l15;
                                if (r < n && s2 < n) K3[s2 * krow + kcol + r] = h[rt][st][reg] * km;
                            }
                }
                // N = X^T H : B operand of k-step kk is register kk%4 of H's row tile kk/4
                d4 nn[NT][NT];
#pragma unroll
                for (int it = 0; it < NT; ++it)
#pragma unroll
                    for (int st = 0; st < NT; ++st) nn[it][st] = (d4){0.0, 0.0, 0.0, 0.0};
#pragma unroll
                for (int kk = 0; kk < KS; ++kk)
#pragma unroll
                    for (int it = 0; it < NT; ++it)
#pragma unroll
                        for (int st = 0; st < NT; ++st)
                            if (!(rsl && st > it)) nn[it][st] = mfma_f64(xf[kk][it], h[kk / 4][st][kk % 4], nn[it][st]);
                EVC_PT_STAMP(fb + 3);
#pragma unroll
                for (int it = 0; it < NT; ++it)
#pragma unroll
                    for (int st = 0; st < NT; ++st)
#pragma unroll
                        for (int reg = 0; reg < 4; ++reg) {
                            const int r2 = it * 16 + l4 + 4 * reg, s2 = st * 16 + l15;
                            if (r2 < n && s2 < n) {
                                if (!rsl) stage[(r2 * n + s2) * QP + ql] = nn[it][st][reg];
                                else if (s2 <= r2) stage[(r2 * (r2 + 1) / 2 + s2) * QP + ql] = nn[it][st][reg];
                            }
                        }
                EVC_PT_STAMP(fb + 4);
            }
            EVC_PT_STAMP(sb + (ql >= 4 ? 2 : 0));
            if constexpr (ROWBUF) {
                row_to_fragments(mf, dlt_next);   // (the fragments of the current matrix are consumed: same wave, in order)
            } else {
#pragma unroll
                for (int rt = 0; rt < NT; ++rt)
#pragma unroll
                    for (int kk = 0; kk < KS; ++kk) mf[rt][kk] = mn[rt][kk];
            }
            have = have_next;
            p = pn;
            q = qn;
            EVC_PT_STAMP(sb + (ql >= 4 ? 3 : 1));
        }
        lds_barrier();
        EVC_PT_STAMP(sb + 4);
        // write-out: 8 lanes cover the pair run of one (r',s')
        const int wl = threadIdx.x & 7;
        int wp = 0, wq = 0;
        const bool wok = pair_of(t, wl, wp, wq);
        const int64_t Cc = (int64_t)wp * n + wq;
        if (a.out && wok) {
            double *out = a.out + g * a.sout;
            if (!rsl) {
                for (int rs = threadIdx.x >> 3; rs < n * n; rs += 32) out[(int64_t)rs * n2 + Cc] = stage[rs * QP + wl];
            } else if (a.out_pairs) {
                // dense (pair, pair) result: row tri(r',s'), column = the leading pair's own triangle index
                const int e = t * QT + wl;
                for (int u = threadIdx.x >> 3; u < npairs; u += 32) out[(int64_t)u * old_ + e] = stage[u * QP + wl];
            } else {
                // (r', s') of stage row u, advanced incrementally: u += 32
                int r2 = (int)tri_row(threadIdx.x >> 3), s2 = (threadIdx.x >> 3) - r2 * (r2 + 1) / 2;
                for (int u = threadIdx.x >> 3; u < npairs; u += 32) {
                    out[((int64_t)r2 * n + s2) * n2 + Cc] = stage[u * QP + wl];
                    s2 += 32;
                    while (s2 > r2) {
                        s2 -= r2 + 1;
                        ++r2;
                    }
                }
            }
        }
        if (a.packed && !a.sym8) {
            double *pk = a.packed + g * a.spacked;
            if (wok)
                for (int rs = threadIdx.x >> 3; rs < n * n; rs += 32) {
                    const int64_t R = rs;
                    if (R >= Cc) pk[tri_index(R, Cc)] = stage[rs * QP + wl] * (R == Cc ? a.diag_mult : 1.0);
                }
            // zero the padding [M, packed_len) once per geometry
            if (blockIdx.x == 0 && t == t_begin) {
                const int64_t M = n2 * (n2 + 1) / 2;
                for (int64_t m = M + threadIdx.x; m < a.packed_len; m += 256) pk[m] = 0.0;
            }
        }
        if (a.packed && a.sym8) {
            // 8-fold compressed vector: only r' >= s', p >= q, (r's') >= (pq) is kept, with its multiplicity
            double *pk = a.packed + g * a.spacked;
            if (wok && wp >= wq) {
                const int64_t v = tri_index(wp, wq);
                const double mq = (wp != wq) ? 2.0 : 1.0;
                if (rsl) {
                    // the stage row index IS u = tri(r',s'); r' == s' <=> u + 1 is a triangular number's end
                    const int u0 = (int)v + (threadIdx.x >> 3);
                    int r2 = (int)tri_row(u0), s2 = u0 - r2 * (r2 + 1) / 2;
                    for (int u = u0; u < npairs; u += 32) {
                        pk[tri_index(u, v)] =
                            stage[u * QP + wl] * ((u == v ? a.diag_mult : 1.0) * mq * (s2 == r2 ? 1.0 : 2.0));
                        s2 += 32;
                        while (s2 > r2) {
                            s2 -= r2 + 1;
                            ++r2;
                        }
                    }
                } else {
                    for (int rs = threadIdx.x >> 3; rs < n * n; rs += 32) {
                        const int r2 = rs / n, s2 = rs - r2 * n;
                        const int64_t u = tri_index(r2, s2);
                        if (r2 >= s2 && u >= v)
                            pk[tri_index(u, v)] =
                                stage[rs * QP + wl] * ((u == v ? a.diag_mult : 1.0) * mq * (r2 != s2 ? 2.0 : 1.0));
                    }
                }
            }
            if (blockIdx.x == 0 && t == t_begin) {
                const int64_t mm = (int64_t)n * (n + 1) / 2, M = mm * (mm + 1) / 2;
                for (int64_t m = M + threadIdx.x; m < a.packed_len; m += 256) pk[m] = 0.0;
            }
        }
        EVC_PT_STAMP(sb + 5);
        lds_barrier();  // the stage is free again; the stores above drain while the next tile is computed
        EVC_PT_STAMP(sb + 6);
    }
    EVC_PT_WG(1);
}

// ------------------------------------------------------------------ software-pipelined pair transform
// pt_kernel above alternates, per tile of eight leading pairs, a matrix-core phase (two matrices per wave) with a
// write-out phase, and every workgroup of the launch does so at the same moments: phase stamps (tools/micro/
// pt_stamps.py) show the 56 MFMAs of a matrix issue in 1.8 us of the 3.5 us a wave spends per matrix (the rest: index
// arithmetic, the operand row's round trip through LDS, the stage writes), and the write-out of a tile takes 2.5-2.9 us
// because the whole chip stores at once (15 MB per phase) and then not at all.  This kernel is the same arithmetic for
// the fully symmetric case (dense (pair, pair) operand, q <= p, lower-triangle results) as ONE software pipeline per
// wave: everything that is not an MFMA is cut into small pieces that are issued between the MFMA groups of the
// CURRENT matrix --
//   * H phase of matrix i (KS groups of NT*NT MFMAs): the stage writes of matrix i-1's result (kept in registers);
//   * N phase: the operand row of matrix i+1 goes from registers to the wave's LDS row and comes back as MFMA
//     fragments (the registers of matrix i's fragments are free after its H phase), the global fetch of matrix i+2's
//     row is issued, half of the write-out passes of an EARLIER tile --
// with a double-buffered stage (8 doubles per result row and buffer, the slot XOR-swizzled by the row against bank
// conflicts of the accumulator layout) and ONE workgroup barrier per tile: barrier(j) sits after the H phase of
// the first matrix of tile j+1 (which wrote the last results of tile j); tile j is then written out during the
// following two N phases, before barrier(j+1), after which its buffer is written again.  The stores of the launch
// are spread evenly over its duration.  (vmcnt counts loads and stores in order on gfx9: the wait for an operand row
// also waits for every store issued before it, so the write-out sits in the N phases only and the row is awaited at
// the start of the next one, an H phase later.)
// MODE 0: dense (pair, pair) result out[tri(r',s')][tri(p,q)];
// MODE 1: the 8-fold compressed packed vector.
template <int NPAD, int MODE>
__global__ __launch_bounds__(256) void pt_pipe_kernel(PairTransformArgs a) {
    constexpr int KS = NPAD / 4;
    constexpr int NT = NPAD / 16;
    constexpr int NPASS_MAX = (NPAD * (NPAD + 1) / 2 + 63) / 64;   // write-out passes of a tile (64 result rows each)
    constexpr int PH2 = (NPASS_MAX + 1) / 2;                       // ... per N phase
    constexpr int NRES = NT * (NT + 1) / 2 * 4;                    // result registers (doubles) of a matrix per lane
    constexpr int SPG = (NRES + KS - 1) / KS;                      // stage writes per MFMA group
    constexpr int FPG = (NT * KS + (KS - 2) - 1) / (KS - 2);       // fragment reads per MFMA group (groups 2..KS-1)
    constexpr int PPG = (PH2 + KS - 1) / KS;                       // write-out passes per MFMA group (N phase)
    constexpr int RAWN = (NPAD * (NPAD + 1) / 2 + 1 + 127) / 128;  // 16-byte loads per lane that cover a row
    extern __shared__ __align__(16) double sm[];
    const int n = a.n;
    const int npairs = n * (n + 1) / 2;
    const int64_t g = blockIdx.y;
    const double *__restrict__ in = a.in + g * a.sin;
    const double *__restrict__ C = a.C + g * a.sC;
    const int ild = a.in_ld ? a.in_ld : npairs, old_ = a.out_ld ? a.out_ld : npairs;   // pitch of the dense (pair, pair) forms
    const int ntiles = (npairs + 7) / 8;
    const int t_begin = blockIdx.x * a.tiles_per_wg, t_end = min(ntiles, t_begin + a.tiles_per_wg);
    if (t_begin >= t_end) return;
    EVC_PT_WG(0);
    EVC_PT_STAMP(0);
    const int niter = 2 * (t_end - t_begin);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int l15 = lane & 15, l4 = lane >> 4;
    double *mrow = sm + wave * kPtRowLen;   // the wave's operand row
    // the stage: result row u = tri(r',s') of buffer b (tile parity), slot s (pair of the tile) at
    //   u * 16 + b * 8 + (s ^ f(u)),  f(u) = 2 ((u >> 1) & 3)
    // (f is even: the slots 2t, 2t+1 of a row stay an aligned 16-byte pair, which the write-out reads with one
    //  ds_read_b128; 16 consecutive rows of one slot fall on 8 bank pairs, a 2-way conflict on the 12 stage writes of
    //  a matrix; f(u + 64) = f(u))
    double *stage = sm + 4 * kPtRowLen;     // (behind the npairs rows: one dump row of 16 doubles)
    char *__restrict__ outb = nullptr;
    if constexpr (MODE == 0) outb = reinterpret_cast<char *>(a.out + g * a.sout);
    else outb = reinterpret_cast<char *>(a.packed + g * a.spacked);

    int foff[NT][KS];   // fragment (rt, kk) of the symmetric n x n matrix in its packed row
#pragma unroll
    for (int rt = 0; rt < NT; ++rt)
#pragma unroll
        for (int kk = 0; kk < KS; ++kk) {
            const int r = rt * 16 + l15, s = 4 * kk + l4;
            const int hi = s > r ? s : r, lo = s > r ? r : s;
            foff[rt][kk] = (r < n && s < n) ? hi * (hi + 1) / 2 + lo : kPtRawMax * 128;   // else: a zero slot
        }
    // stage index of result register (it, st <= it, reg) for slot `wave` of buffer 0 (registers that hold no result
    // go to a dump row behind the last one); the slot of the second matrix of a tile and the buffer are XORed in:
    // (slot + 4) ^ f = (slot ^ f) ^ 4 for slot < 4
    int sa[NT][NT][4];
#pragma unroll
    for (int it = 0; it < NT; ++it)
#pragma unroll
        for (int st = 0; st <= it; ++st)
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) {
                const int r2 = it * 16 + l4 + 4 * reg, s2 = st * 16 + l15;
                const int u = r2 * (r2 + 1) / 2 + s2;
                sa[it][st][reg] = (r2 < n && s2 <= r2) ? u * 16 + (wave ^ (((u >> 1) & 3) << 1)) : npairs * 16 + wave;
            }
    [[maybe_unused]] const int dq = l15 - l4;   // result register reg of a diagonal tile is r' == s'  <=>  dq == 4 reg
    if (lane < 4) mrow[kPtRawMax * 128 + lane] = 0.0;

    d2 raw[RAWN];
    // row e of the operand (e clamped: idle slots fetch row 0 and their result is never written out)
    auto fetch = [&](int e) -> int {
        const double *row = in + (int64_t)(e < npairs ? e : 0) * ild;
        const int d_ = (int)((reinterpret_cast<uintptr_t>(row) >> 3) & 1);
        const double *w0 = row - d_;          // 16-byte aligned window
        const int lim = npairs + d_;          // valid doubles of the window
        // every lane loads an aligned 16-byte granule that holds at least one double of the row (lanes past the end
        // re-read the first granule): no predicates, and such a load cannot leave the pages of the operand
#pragma unroll
        for (int u = 0; u < RAWN; ++u) {
            const int j = 128 * u + 2 * lane;
            raw[u] = *reinterpret_cast<const d2 *>(w0 + (j < lim ? j : 0));
        }
        return d_;
    };
    auto park = [&]() {
#pragma unroll
        for (int u = 0; u < RAWN; ++u) *reinterpret_cast<d2 *>(mrow + 128 * u + 2 * lane) = raw[u];
    };
    auto is_diag = [&](int x) -> bool {
        const int r = tri_row_small(x);
        return x == r * (r + 3) / 2;
    };

    const int e0 = 8 * t_begin + wave;   // this wave's leading pair of iteration i: e0 + 4 i
    int d_rd = fetch(e0);
    // the X fragments straight from global memory (every wave its own copy: 16 rows of 128 bytes per load instruction,
    // one latency together with the first operand row; no LDS staging, no barrier)
    double xf[KS][NT];
#pragma unroll
    for (int kk = 0; kk < KS; ++kk)
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            const int d = 4 * kk + l4, c = t * 16 + l15;
            const bool ok = d < n && c < n;
            const double *src = C + (ok ? (a.ct ? c * n + d : d * n + c) : 0);
            const double v = *src;
            xf[kk][t] = ok ? v : 0.0;
        }
    if constexpr (MODE == 1) {
        if (blockIdx.x == 0) {   // zero the padding [M, packed_len) once per geometry
            double *pk = a.packed + g * a.spacked;
            const int64_t M = (int64_t)npairs * (npairs + 1) / 2;
            for (int64_t m = M + threadIdx.x; m < a.packed_len; m += 256) pk[m] = 0.0;
        }
    }
    EVC_PT_STAMP(1);
    double mf[NT][KS];
    park();
#pragma unroll
    for (int rt = 0; rt < NT; ++rt)
#pragma unroll
        for (int kk = 0; kk < KS; ++kk) mf[rt][kk] = mrow[foff[rt][kk] + d_rd];
    int d_next = fetch(e0 + 4);
    EVC_PT_STAMP(2);

    // write-out: 4 lanes cover the pair run of one result row, two pairs (16 bytes) each; thread (wl, ur) takes the
    // columns wl, wl + 1 (wl even) of rows u0 + 64 k
    const int wl = 2 * (threadIdx.x & 3), ur = threadIdx.x >> 2;
    [[maybe_unused]] const unsigned stride_b = 64u * (unsigned)old_ * 8u;     // MODE 0: bytes between two passes

    // the result of a matrix stays in registers until it is staged during the next matrix's H phase: two register
    // sets, alternating (the loop body is instantiated for even and odd i)
    d4 nnA[NT][NT], nnB[NT][NT];
#pragma unroll
    for (int it = 0; it < NT; ++it)
#pragma unroll
        for (int st = 0; st < NT; ++st) nnA[it][st] = nnB[it][st] = (d4){0.0, 0.0, 0.0, 0.0};

    // One iteration = one matrix.  COMPUTE: the main loop (idle slots of the last tile compute on row 0, their result is
    // never written out); without: the two drain-only iterations behind it.  Everything that is not an MFMA sits
    // BETWEEN two MFMAs in program order (sched_barrier pins it there), a few instructions at a time: the LDS and
    // memory operations are spread evenly and their latencies end behind MFMAs, not behind each other.
    auto iteration = [&](auto compute_tag, auto odd_tag, const int i, d4 (&nnp)[NT][NT], d4 (&nn)[NT][NT]) {
        constexpr bool COMPUTE = decltype(compute_tag)::value, ODD = decltype(odd_tag)::value;
        const int ei = e0 + 4 * i;
        EVC_PT_STAMP(3 + 3 * i);
        // ---------------------------------------------------------------- H = M X  (+ stage writes of matrix i-1)
        d4 h[NT][NT];
#pragma unroll
        for (int rt = 0; rt < NT; ++rt)
#pragma unroll
            for (int st = 0; st < NT; ++st) h[rt][st] = (d4){0.0, 0.0, 0.0, 0.0};
        if constexpr (COMPUTE || !ODD) {
            // results of matrix i-1: slot wave + 4 ((i-1) & 1) of the buffer of ITS tile (iteration 0 and idle slots
            // write values nobody reads)
            const int xm = ((((i - 1) >> 1) & 1) << 3) | (ODD ? 0 : 4);
#pragma unroll
            for (int kk = 0; kk < KS; ++kk) {
#pragma unroll
                for (int m = 0; m < NT * NT; ++m) {
                    if constexpr (COMPUTE) {
                        const int rt = m / NT, st = m % NT;
                        // (equal priorities let a wave that issues MFMAs back to back keep the issue port: the SIMD mate's
                        // loads, stores and LDS operations go first, tools/micro/mfma_coissue.hip)
                        __builtin_amdgcn_s_setprio(0);
                        h[rt][st] = mfma_f64(mf[rt][kk], xf[kk][st], h[rt][st]);
                        __builtin_amdgcn_s_setprio(1);
                    }
                    const int f = kk * SPG + m;   // the stage write that follows this MFMA
                    if (m < SPG && f < NRES) {
                        const int tile = f / 4, reg = f % 4;
                        const int it = tile == 0 ? 0 : 1, st2 = tile == 2 ? 1 : 0;   // tiles (0,0), (1,0), (1,1)
                        double v = nnp[it][st2][reg];
                        // MODE 1: the write-out doubles every element, r' == s' has multiplicity 1
                        if (MODE == 1 && it == st2) v *= (dq == 4 * reg) ? 0.5 : 1.0;
                        stage[sa[it][st2][reg] ^ xm] = v;
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        }
        EVC_PT_STAMP(4 + 3 * i);
        if (!ODD && i >= 2 && i <= niter) lds_barrier();
        EVC_PT_STAMP(5 + 3 * i);
        // ---------------------------------------------------------------- N = X^T H
#pragma unroll
        for (int it = 0; it < NT; ++it)
#pragma unroll
            for (int st = 0; st < NT; ++st) nn[it][st] = (d4){0.0, 0.0, 0.0, 0.0};
        {
            // write-out: tile j's result is complete after barrier(j) (H phase of iteration 2j+2); it is written out in
            // the N phases of iterations 2j+2 (passes 0 .. PH2 - 1) and 2j+3 (the rest), PPG passes per MFMA group
            const int jn = ODD ? (i - 3) / 2 : (i - 2) / 2;
            constexpr int kn = ODD ? PH2 : 0;
            const bool dn = (ODD ? i >= 3 : i >= 2) && 2 * jn < niter;
            // per tile and thread: LDS index of pass 0, byte offset of its store, rows left, the factors of its two columns
            const int ew = 8 * (t_begin + jn) + wl;            // the first pair (column) this thread writes
            const int u0 = (MODE == 0 ? 0 : ew) + ur;          // MODE 1: rows u >= v = ew only
            const int rem = (dn && ew < npairs) ? npairs - u0 : 0;   // pass k is valid  <=>  64 k < rem
            const bool two = ew + 1 < npairs;                  // (the last pair of an odd count stands alone)
            const d2 *dsrc = reinterpret_cast<const d2 *>(stage + (u0 * 16 + ((jn & 1) << 3) + (wl ^ (((u0 >> 1) & 3) << 1))));
            d2 fac = {1.0, 1.0};
            [[maybe_unused]] d2 fac0 = {1.0, 1.0};
            unsigned off0 = 0;
            [[maybe_unused]] unsigned offA = 0;
            if (dn) {
                if constexpr (MODE == 0) {
                    off0 = (unsigned)(u0 * old_ + ew) * 8u;
                } else {
                    fac[0] = is_diag(ew) ? 2.0 : 4.0;              // multiplicity of (p,q) x 2 (see the stage writes)
                    fac[1] = is_diag(ew + 1) ? 2.0 : 4.0;
                    // pass 0: the thread with u == v takes diag_mult; row u = ew has no column ew + 1 (u < v)
                    fac0[0] = ur == 0 ? fac[0] * a.diag_mult : fac[0];
                    fac0[1] = ur == 1 ? fac[1] * a.diag_mult : fac[1];
                    off0 = (unsigned)(u0 * (u0 + 1) / 2 + ew) * 8u;
                    offA = (unsigned)(64 * u0) * 8u;               // tri(u0 + 64 k) = tri(u0) + k (64 u0) + 2048 k^2 + 32 k
                }
            }
            constexpr int NM = NT * (NT + 1) / 2;   // MFMAs of a group: tiles (0,0), (1,0), (1,1)
#pragma unroll
            for (int kk = 0; kk < KS; ++kk) {
                d2 dv[PPG];
#pragma unroll
                for (int c = 0; c < PPG; ++c) {
                    const int k = kk * PPG + c;   // pass kn + k
                    dv[c] = (k < PH2 && 64 * (kn + k) < rem) ? dsrc[512 * (kn + k)] : (d2){0.0, 0.0};
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int m = 0; m < NM; ++m) {
                    if constexpr (COMPUTE) {
                        const int it = m == 0 ? 0 : 1, st = m == 2 ? 1 : 0;
                        __builtin_amdgcn_s_setprio(0);
                        nn[it][st] = mfma_f64(xf[kk][it], h[kk / 4][st][kk % 4], nn[it][st]);
                        __builtin_amdgcn_s_setprio(1);
                    }
                    if (COMPUTE && m == 0) {
                        // the operand row of matrix i+1 (fetched one iteration ago) -> the wave's LDS row -> fragments;
                        // the row of matrix i+2 is requested (behind the last matrix: row 0, unused)
                        if (kk == 0) {
                            park();
                            d_rd = d_next;
                        }
                        if (kk == 1) d_next = fetch(ei + 8);
                        if (kk >= 2) {
#pragma unroll
                            for (int rt = 0; rt < NT; ++rt)
#pragma unroll
                                for (int k2 = 0; k2 < KS; ++k2)
                                    if ((rt * KS + k2) / FPG == kk - 2) mf[rt][k2] = mrow[foff[rt][k2] + d_rd];
                        }
                    }
                    if (m == NM - 1) {
#pragma unroll
                        for (int c = 0; c < PPG; ++c) {
                            const int k = kk * PPG + c;
                            if (k < PH2 && 64 * (kn + k) < rem) {
                                const unsigned kq = (unsigned)(kn + k);
                                unsigned off;
                                d2 v;
                                bool second = two;
                                if constexpr (MODE == 0) {
                                    off = off0 + kq * stride_b;
                                    v = dv[c];
                                } else {
                                    off = off0 + kq * offA + (2048u * kq * kq + 32u * kq) * 8u;
                                    v = dv[c] * (kq == 0 ? fac0 : fac);
                                    if (kq == 0) second = two && ur >= 1;   // row u = ew: column ew + 1 is above the diagonal
                                }
                                if (second) *reinterpret_cast<d2 *>(outb + off) = v;
                                else *reinterpret_cast<double *>(outb + off) = v[0];
                            }
                        }
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
                if (i == 3) EVC_PT_STAMP(40 + kk);
            }
        }
    };
    for (int i = 0; i < niter; i += 2) {   // (niter is even)
        iteration(std::true_type{}, std::false_type{}, i, nnB, nnA);
        iteration(std::true_type{}, std::true_type{}, i + 1, nnA, nnB);
    }
    iteration(std::false_type{}, std::false_type{}, niter, nnB, nnA);
    iteration(std::false_type{}, std::true_type{}, niter + 1, nnA, nnB);
    EVC_PT_STAMP(3 + 3 * (niter + 2));
    EVC_PT_WG(1);
}

// ------------------------------------------------------------------ the same for a few geometries: tiles of 4 pairs
// With one or two geometries per launch pt_pipe_kernel has 59 workgroups of two matrices per wave and spends more
// time in its prologue and drain-only tail than in between.  Here a tile is 4 leading pairs -- ONE matrix per wave --
// so there are twice as many workgroups of half the length (the write-out runs are 32 bytes, which does not matter at
// this size).  Same pipeline with period one: matrix i's result is staged during the H phase of matrix i+1, barrier,
// and the tile is written out during that matrix's N phase; stage = 4 doubles per result row and buffer.  No K3.
template <int NPAD, int MODE>
__global__ __launch_bounds__(256) void pt_pipe4_kernel(PairTransformArgs a) {
    constexpr int KS = NPAD / 4;
    constexpr int NT = NPAD / 16;
    constexpr int NPASS = (NPAD * (NPAD + 1) / 2 + 127) / 128;     // write-out passes of a tile (128 result rows each)
    constexpr int NRES = NT * (NT + 1) / 2 * 4;
    constexpr int SPG = (NRES + KS - 1) / KS;
    constexpr int FPG = (NT * KS + (KS - 2) - 1) / (KS - 2);
    constexpr int PPG = (NPASS + KS - 1) / KS;
    constexpr int RAWN = (NPAD * (NPAD + 1) / 2 + 1 + 127) / 128;
    extern __shared__ __align__(16) double sm[];
    const int n = a.n;
    const int npairs = n * (n + 1) / 2;
    const int64_t g = blockIdx.y;
    const double *__restrict__ in = a.in + g * a.sin;
    const double *__restrict__ C = a.C + g * a.sC;
    const int ild = a.in_ld ? a.in_ld : npairs, old_ = a.out_ld ? a.out_ld : npairs;   // pitch of the dense (pair, pair) forms
    const int ntiles = (npairs + 3) / 4;
    const int t_begin = blockIdx.x * a.tiles_per_wg, t_end = min(ntiles, t_begin + a.tiles_per_wg);
    if (t_begin >= t_end) return;
    const int niter = t_end - t_begin;   // one matrix per wave and tile
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int l15 = lane & 15, l4 = lane >> 4;
    double *mrow = sm + wave * kPtRowLen;
    // stage: row u, buffer b, slot s at u * 8 + b * 4 + (s ^ f(u)), f(u) = 2 ((u >> 1) & 1); one dump row behind
    double *stage = sm + 4 * kPtRowLen;
    char *__restrict__ outb = nullptr;
    if constexpr (MODE == 0) outb = reinterpret_cast<char *>(a.out + g * a.sout);
    else outb = reinterpret_cast<char *>(a.packed + g * a.spacked);

    int foff[NT][KS];
#pragma unroll
    for (int rt = 0; rt < NT; ++rt)
#pragma unroll
        for (int kk = 0; kk < KS; ++kk) {
            const int r = rt * 16 + l15, s = 4 * kk + l4;
            const int hi = s > r ? s : r, lo = s > r ? r : s;
            foff[rt][kk] = (r < n && s < n) ? hi * (hi + 1) / 2 + lo : kPtRawMax * 128;
        }
    int sa[NT][NT][4];
#pragma unroll
    for (int it = 0; it < NT; ++it)
#pragma unroll
        for (int st = 0; st <= it; ++st)
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) {
                const int r2 = it * 16 + l4 + 4 * reg, s2 = st * 16 + l15;
                const int u = r2 * (r2 + 1) / 2 + s2;
                sa[it][st][reg] = (r2 < n && s2 <= r2) ? u * 8 + (wave ^ (((u >> 1) & 1) << 1)) : npairs * 8 + wave;
            }
    [[maybe_unused]] const int dq = l15 - l4;
    if (lane < 4) mrow[kPtRawMax * 128 + lane] = 0.0;

    d2 raw[RAWN];
    auto fetch = [&](int e) -> int {
        const double *row = in + (int64_t)(e < npairs ? e : 0) * ild;
        const int d_ = (int)((reinterpret_cast<uintptr_t>(row) >> 3) & 1);
        const double *w0 = row - d_;
        const int lim = npairs + d_;
#pragma unroll
        for (int u = 0; u < RAWN; ++u) {
            const int j = 128 * u + 2 * lane;
            raw[u] = *reinterpret_cast<const d2 *>(w0 + (j < lim ? j : 0));
        }
        return d_;
    };
    auto park = [&]() {
#pragma unroll
        for (int u = 0; u < RAWN; ++u) *reinterpret_cast<d2 *>(mrow + 128 * u + 2 * lane) = raw[u];
    };
    auto is_diag = [&](int x) -> bool {
        const int r = tri_row_small(x);
        return x == r * (r + 3) / 2;
    };

    const int e0 = 4 * t_begin + wave;   // this wave's leading pair of iteration i: e0 + 4 i
    int d_rd = fetch(e0);
    double xf[KS][NT];
#pragma unroll
    for (int kk = 0; kk < KS; ++kk)
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            const int d = 4 * kk + l4, c = t * 16 + l15;
            const bool ok = d < n && c < n;
            const double v = C[ok ? (a.ct ? c * n + d : d * n + c) : 0];
            xf[kk][t] = ok ? v : 0.0;
        }
    if constexpr (MODE == 1) {
        if (blockIdx.x == 0) {
            double *pk = a.packed + g * a.spacked;
            const int64_t M = (int64_t)npairs * (npairs + 1) / 2;
            for (int64_t m = M + threadIdx.x; m < a.packed_len; m += 256) pk[m] = 0.0;
        }
    }
    double mf[NT][KS];
    park();
#pragma unroll
    for (int rt = 0; rt < NT; ++rt)
#pragma unroll
        for (int kk = 0; kk < KS; ++kk) mf[rt][kk] = mrow[foff[rt][kk] + d_rd];
    int d_next = fetch(e0 + 4);

    // write-out: 2 lanes cover the 4 pairs of one result row (16 bytes each); thread (wl, ur): columns wl, wl + 1 of
    // rows u0 + 128 k
    const int wl = 2 * (threadIdx.x & 1), ur = threadIdx.x >> 1;
    [[maybe_unused]] const unsigned stride_b = 128u * (unsigned)old_ * 8u;

    d4 nnp[NT][NT];
#pragma unroll
    for (int it = 0; it < NT; ++it)
#pragma unroll
        for (int st = 0; st < NT; ++st) nnp[it][st] = (d4){0.0, 0.0, 0.0, 0.0};

    auto iteration = [&](auto compute_tag, const int i) {
        constexpr bool COMPUTE = decltype(compute_tag)::value;
        const int ei = e0 + 4 * i;
        // ---------------------------------------------------------------- H = M X  (+ stage writes of matrix i-1)
        d4 h[NT][NT];
#pragma unroll
        for (int rt = 0; rt < NT; ++rt)
#pragma unroll
            for (int st = 0; st < NT; ++st) h[rt][st] = (d4){0.0, 0.0, 0.0, 0.0};
        {
            const int xm = ((i - 1) & 1) << 2;   // buffer of tile i-1
#pragma unroll
            for (int kk = 0; kk < KS; ++kk) {
#pragma unroll
                for (int m = 0; m < NT * NT; ++m) {
                    if constexpr (COMPUTE) {
                        const int rt = m / NT, st = m % NT;
                        __builtin_amdgcn_s_setprio(0);
                        h[rt][st] = mfma_f64(mf[rt][kk], xf[kk][st], h[rt][st]);
                        __builtin_amdgcn_s_setprio(1);
                    }
                    const int f = kk * SPG + m;
                    if (m < SPG && f < NRES) {
                        const int tile = f / 4, reg = f % 4;
                        const int it = tile == 0 ? 0 : 1, st2 = tile == 2 ? 1 : 0;
                        double v = nnp[it][st2][reg];
                        if (MODE == 1 && it == st2) v *= (dq == 4 * reg) ? 0.5 : 1.0;
                        stage[sa[it][st2][reg] ^ xm] = v;
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        }
        if (i >= 1) lds_barrier();   // tile i-1 is complete (and tile i-2 written out by everybody)
        // ---------------------------------------------------------------- N = X^T H
        d4 nn[NT][NT];
#pragma unroll
        for (int it = 0; it < NT; ++it)
#pragma unroll
            for (int st = 0; st < NT; ++st) nn[it][st] = (d4){0.0, 0.0, 0.0, 0.0};
        {
            const int jn = i - 1;                              // the tile written out now
            const bool dn = jn >= 0;
            const int ew = 4 * (t_begin + jn) + wl;
            const int u0 = (MODE == 0 ? 0 : ew) + ur;
            const int rem = (dn && ew < npairs) ? npairs - u0 : 0;   // pass k is valid  <=>  128 k < rem
            const bool two = ew + 1 < npairs;
            const d2 *dsrc = reinterpret_cast<const d2 *>(stage + (u0 * 8 + ((jn & 1) << 2) + (wl ^ (((u0 >> 1) & 1) << 1))));
            d2 fac = {1.0, 1.0};
            [[maybe_unused]] d2 fac0 = {1.0, 1.0};
            unsigned off0 = 0;
            [[maybe_unused]] unsigned offA = 0;
            if (dn) {
                if constexpr (MODE == 0) {
                    off0 = (unsigned)(u0 * old_ + ew) * 8u;
                } else {
                    fac[0] = is_diag(ew) ? 2.0 : 4.0;
                    fac[1] = is_diag(ew + 1) ? 2.0 : 4.0;
                    fac0[0] = ur == 0 ? fac[0] * a.diag_mult : fac[0];
                    fac0[1] = ur == 1 ? fac[1] * a.diag_mult : fac[1];
                    off0 = (unsigned)(u0 * (u0 + 1) / 2 + ew) * 8u;
                    offA = (unsigned)(128 * u0) * 8u;   // tri(u0 + 128 k) = tri(u0) + k (128 u0) + 8192 k^2 + 64 k
                }
            }
            constexpr int NM = NT * (NT + 1) / 2;
#pragma unroll
            for (int kk = 0; kk < KS; ++kk) {
                d2 dv[PPG];
#pragma unroll
                for (int c = 0; c < PPG; ++c) {
                    const int k = kk * PPG + c;
                    dv[c] = (k < NPASS && 128 * k < rem) ? dsrc[512 * k] : (d2){0.0, 0.0};
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int m = 0; m < NM; ++m) {
                    if constexpr (COMPUTE) {
                        const int it = m == 0 ? 0 : 1, st = m == 2 ? 1 : 0;
                        __builtin_amdgcn_s_setprio(0);
                        nn[it][st] = mfma_f64(xf[kk][it], h[kk / 4][st][kk % 4], nn[it][st]);
                        __builtin_amdgcn_s_setprio(1);
                    }
                    if (COMPUTE && m == 0) {
                        if (kk == 0) {
                            park();
                            d_rd = d_next;
                        }
                        if (kk == 1) d_next = fetch(ei + 8);
                        if (kk >= 2) {
#pragma unroll
                            for (int rt = 0; rt < NT; ++rt)
#pragma unroll
                                for (int k2 = 0; k2 < KS; ++k2)
                                    if ((rt * KS + k2) / FPG == kk - 2) mf[rt][k2] = mrow[foff[rt][k2] + d_rd];
                        }
                    }
                    if (m == NM - 1) {
#pragma unroll
                        for (int c = 0; c < PPG; ++c) {
                            const int k = kk * PPG + c;
                            if (k < NPASS && 128 * k < rem) {
                                const unsigned kq = (unsigned)k;
                                unsigned off;
                                d2 v;
                                bool second = two;
                                if constexpr (MODE == 0) {
                                    off = off0 + kq * stride_b;
                                    v = dv[c];
                                } else {
                                    off = off0 + kq * offA + (8192u * kq * kq + 64u * kq) * 8u;
                                    v = dv[c] * (kq == 0 ? fac0 : fac);
                                    if (kq == 0) second = two && ur >= 1;
                                }
                                if (second) *reinterpret_cast<d2 *>(outb + off) = v;
                                else *reinterpret_cast<double *>(outb + off) = v[0];
                            }
                        }
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        }
#pragma unroll
        for (int it = 0; it < NT; ++it)
#pragma unroll
            for (int st = 0; st <= it; ++st) nnp[it][st] = nn[it][st];
    };
    for (int i = 0; i < niter; ++i) iteration(std::true_type{}, i);
    iteration(std::false_type{}, niter);
}

int launch_pair_transform(const PairTransformArgs &a_in, int count, hipStream_t st) {
    if (a_in.n > kPairTransformMaxN) {   // 32 < n <= 64: the symmetric step alone exists (pair64.hip)
        if (pair64_applicable(a_in)) return launch_pair_transform64(a_in, count, st);
        set_error("pair transform: n=%d beyond %d needs the symmetric dense-pair form", a_in.n, kPairTransformMaxN);
        return -1;
    }
    PairTransformArgs a = a_in;
    const int n = a.n;
    const int npad = (n + 15) / 16 * 16;
    const int ntq = (n + 7) / 8;
    constexpr int tpw_env = 4;   // (2/3/5/8 tiles per workgroup measured slower, round 2 and again with ptd_kernel in round 4)
    // few geometries: keep one tile per workgroup so that there are enough workgroups for the chip
    a.tiles_per_wg = (count < 4 || tpw_env < 1) ? 1 : (tpw_env > ntq ? ntq : tpw_env);
    const int ntiles = a.lead_sym ? (n * (n + 1) / 2 + 7) / 8 : n * ntq;
    const dim3 grid((unsigned)((ntiles + a.tiles_per_wg - 1) / a.tiles_per_wg), (unsigned)count);
    const size_t stage_rows = a.rs_lower ? (size_t)n * (n + 1) / 2 : (size_t)n * n;
    // fully symmetric step of the compressed layout's pipeline: the software-pipelined kernel, if two of its workgroups
    // fit a CU's LDS (n <= 30; EVC_PT_PIPE=0: the phase-alternating kernel below)
    static const bool pipe_on = !(getenv("EVC_PT_PIPE") && atoi(getenv("EVC_PT_PIPE")) == 0);
    if (pair_transform_dma_applicable(a, count)) return launch_pair_transform_dma(a, count, st);
    if (pipe_on && a.lead_sym && a.in_lower && a.rs_lower && a.in_pairs && (npad == 16 || npad == 32)) {
        const int mode = a.k3 ? -1 : (a.out && a.out_pairs && !a.packed) ? 0 : (a.packed && a.sym8 && !a.out) ? 1 : -1;
        const size_t npairs = (size_t)n * (n + 1) / 2;
        const size_t lds = sizeof(double) * ((size_t)4 * kPtRowLen + npairs * 16 + 16);
        constexpr bool pipe4_on = true;
        if (pipe4_on && mode >= 0 && count < 4 && npairs >= 8) {
            // a few geometries: tiles of 4 pairs, one per workgroup (twice the workgroups, half the length)
            a.tiles_per_wg = 1;
            const size_t lds4 = sizeof(double) * ((size_t)4 * kPtRowLen + npairs * 8 + 8);
            const dim3 grid4((unsigned)((npairs + 3) / 4), (unsigned)count);
#define EVC_PT_PIPE4_CASE(NP_, MODE_) hipLaunchKernelGGL((pt_pipe4_kernel<NP_, MODE_>), grid4, dim3(256), lds4, st, a)
            if (npad == 16 && mode == 0) EVC_PT_PIPE4_CASE(16, 0);
            else if (npad == 16) EVC_PT_PIPE4_CASE(16, 1);
            else if (mode == 0) EVC_PT_PIPE4_CASE(32, 0);
            else EVC_PT_PIPE4_CASE(32, 1);
#undef EVC_PT_PIPE4_CASE
            note_kernel(EVC_PROF_PAIR_TRANSFORM, "pt_pipe4_kernel<%d,%d>", npad, mode);
            EVC_LAUNCH_CHECK("pair_transform_pipe4");
            return 0;
        }
        if (mode >= 0 && lds <= 80 * 1024) {
            const dim3 gridp((unsigned)(((npairs + 7) / 8 + a.tiles_per_wg - 1) / a.tiles_per_wg), (unsigned)count);
#define EVC_PT_PIPE_CASE(NP_, MODE_)                                                                            \
    do {                                                                                                        \
        static LdsAttr attr;                                                                                    \
        if (int rc = allow_dynamic_lds(pt_pipe_kernel<NP_, MODE_>, attr, 160 * 1024, "pair_transform")) return rc; \
        hipLaunchKernelGGL((pt_pipe_kernel<NP_, MODE_>), gridp, dim3(256), lds, st, a);                         \
    } while (0)
            if (npad == 16 && mode == 0) EVC_PT_PIPE_CASE(16, 0);
            else if (npad == 16) EVC_PT_PIPE_CASE(16, 1);
            else if (mode == 0) EVC_PT_PIPE_CASE(32, 0);
            else EVC_PT_PIPE_CASE(32, 1);
#undef EVC_PT_PIPE_CASE
            note_kernel(EVC_PROF_PAIR_TRANSFORM, "pt_pipe_kernel<%d,%d>", npad, mode);
            EVC_LAUNCH_CHECK("pair_transform_pipe");
            return 0;
        }
    }
    // dense (pair, pair) operand: coalesced row fetch through a wave-private LDS row (EVC_PT_ROWBUF=0: gather)
    constexpr bool rowbuf_on = true;
    const bool rowbuf = rowbuf_on && a.in_pairs;
    const size_t rowbuf_doubles = rowbuf ? (size_t)4 * kPtRowLen + 2 : 0;
    if (npad == 16) {
        const size_t lds = sizeof(double) * ((size_t)16 * 16 + stage_rows * 9 + rowbuf_doubles);
        if (rowbuf) hipLaunchKernelGGL((pt_kernel<16, true>), grid, dim3(256), lds, st, a);
        else hipLaunchKernelGGL((pt_kernel<16, false>), grid, dim3(256), lds, st, a);
    } else if (npad == 32) {
        const size_t lds = sizeof(double) * ((size_t)32 * 48 + stage_rows * 9 + rowbuf_doubles);
        if (rowbuf) {
            static LdsAttr attr;
            if (int rc = allow_dynamic_lds(pt_kernel<32, true>, attr, 160 * 1024, "pair_transform")) return rc;
            hipLaunchKernelGGL((pt_kernel<32, true>), grid, dim3(256), lds, st, a);
        } else {
            static LdsAttr attr;
            if (int rc = allow_dynamic_lds(pt_kernel<32, false>, attr, 160 * 1024, "pair_transform")) return rc;
            hipLaunchKernelGGL((pt_kernel<32, false>), grid, dim3(256), lds, st, a);
        }
    } else {
        set_error("pair_transform: n=%d not supported (1..32)", n);
        return -1;
    }
    note_kernel(EVC_PROF_PAIR_TRANSFORM, "pt_kernel<%d,%d>", npad, rowbuf ? 1 : 0);
    EVC_LAUNCH_CHECK("pair_transform");
    return 0;
}

}  // namespace evc

// ------------------------------------------------------------------ C ABI
using namespace evc;

extern "C" int evc_quarter_transform(const double *in, const double *C, int c_transposed, int n, double *out,
                                     void *stream) {
    EVC_REQUIRE(in && C && out, "evc_quarter_transform: null pointer");
    EVC_REQUIRE(n >= 1 && n <= 96, "evc_quarter_transform: n=%d out of range 1..96", n);
    EVC_REQUIRE(in != out, "evc_quarter_transform: in and out must not alias");
    return launch_quarter_transform(in, 0, C, 0, c_transposed, n, out, 0, 1, as_stream(stream));
}

extern "C" int evc_four_index_transform(const double *in, const double *C, int c_transposed, int n, double *out,
                                        double *tmp, double *three_quarter, void *stream) {
    EVC_REQUIRE(in && C && out && tmp, "evc_four_index_transform: null pointer");
    EVC_REQUIRE(n >= 1 && n <= 96, "evc_four_index_transform: n=%d out of range 1..96", n);
    EVC_REQUIRE(in != out && in != tmp && out != tmp && three_quarter != out && three_quarter != tmp &&
                    three_quarter != in,
                "evc_four_index_transform: buffers must not alias");
    hipStream_t st = as_stream(stream);
    auto qt = [&](const double *src, double *dst) {
        return launch_quarter_transform(src, 0, C, 0, c_transposed, n, dst, 0, 1, st);
    };
    // in -> out -> tmp -> (three_quarter | out) -> out ; the third result must survive in
    // `three_quarter` when requested, otherwise ping-pong between out and tmp.
    int rc;
    if ((rc = qt(in, out))) return rc;
    if ((rc = qt(out, tmp))) return rc;
    double *third = three_quarter ? three_quarter : out;
    if ((rc = qt(tmp, third))) return rc;
    if (three_quarter) return qt(third, out);
    if ((rc = qt(third, tmp))) return rc;
    // result sits in tmp: copy back (device-to-device, same stream)
    hipError_t e = hipMemcpyAsync(out, tmp, sizeof(double) * (size_t)n * n * n * n, hipMemcpyDeviceToDevice, st);
    if (e != hipSuccess) {
        set_error("evc_four_index_transform: copy failed: %s", hipGetErrorString(e));
        return (int)e;
    }
    return 0;
}

extern "C" int evc_pack_pair_sym(const double *h2, int n, double diag_mult, double *out, int64_t out_len,
                                 void *stream) {
    EVC_REQUIRE(h2 && out, "evc_pack_pair_sym: null pointer");
    EVC_REQUIRE(n >= 1 && n <= 215, "evc_pack_pair_sym: n=%d out of range", n);
    const int64_t n2 = (int64_t)n * n, M = n2 * (n2 + 1) / 2;
    EVC_REQUIRE(out_len >= M, "evc_pack_pair_sym: out_len=%lld < M=%lld", (long long)out_len, (long long)M);
    return launch_pack(h2, 0, n, diag_mult, out, 0, out_len, 1, as_stream(stream));
}

extern "C" int evc_unpack_pair_sym(const double *packed, int n, double *out, void *stream) {
    EVC_REQUIRE(packed && out, "evc_unpack_pair_sym: null pointer");
    EVC_REQUIRE(n >= 1 && n <= 215, "evc_unpack_pair_sym: n=%d out of range", n);
    return launch_unpack(packed, 0, n, out, 0, 1, as_stream(stream));
}

#ifdef EVC_DEBUG_STAMPS
// Debug: the phase stamps of the last pt_kernel launch (4 waves x 64 stamps).
extern "C" int evc_debug_read_pt(long long *stamps) {
    return (int)hipMemcpyFromSymbol(stamps, HIP_SYMBOL(evc::g_pt_stamp), sizeof(long long) * 4 * 64);
}
extern "C" int evc_debug_read_pt_wg(long long *stamps) {
    return (int)hipMemcpyFromSymbol(stamps, HIP_SYMBOL(evc::g_pt_wg), sizeof(long long) * 4096 * 3);
}
#endif
