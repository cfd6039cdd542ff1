// K13: Y2 = K3 . Gs contraction of the Loewdin-response gradient (gradients_loewdin.py:210-222), the split-K MFMA
// kernels for the dense layouts and the fused pair-block kernel of the compressed pipeline (N <= 32; the LDS-DMA
// variant for 17 <= N <= 30 lives in pair_dma.hip).  blockIdx.y = geometry of the batch (kernels.hpp).
#include <stdlib.h>

#include "common.hpp"
#include "kernels.hpp"

namespace evc {

typedef double d4 __attribute__((ext_vector_type(4)));
typedef double d2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ d4 mfma_f64(double a, double b, d4 c) {
    return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
}
// Both operands are [k][n] row-major, so each MFMA fragment load is 16 contiguous doubles.
// number of K slabs = partial results per geometry (fixed for the life of the process: it sizes the workspace)
static int y2_slab_count() {
    static const int v = [] {
        return 64;
    }();
    return v;
}
int y2_slabs(int) { return y2_slab_count(); }
// slabs the partial buffer of the pipeline is sized for (the fused kernel below uses up to that many workgroups)
// (beyond 32 orbitals y2_64_kernel deals the pairs of ONE geometry to a full round of the chip)
int y2_slab_capacity(int n) { return n > kPairTransformMaxN ? 256 : (y2_slab_count() > 128 ? y2_slab_count() : 128); }

template <int NT>
__global__ __launch_bounds__(256) void y2_kernel(const double *__restrict__ GsT, const double *__restrict__ K3,
                                                 int n, int64_t ktot, double *__restrict__ partial, int64_t sws) {
    __shared__ double red[4][NT * 16][NT * 16 + 1];
    GsT += (int64_t)blockIdx.y * sws;
    K3 += (int64_t)blockIdx.y * sws;
    partial += (int64_t)blockIdx.y * sws;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int l15 = lane & 15, l4 = lane >> 4;
    const int64_t ksteps = (ktot + 3) / 4;
    const int64_t nw = (int64_t)gridDim.x * 4;
    const int64_t per = (ksteps + nw - 1) / nw;
    const int64_t w = (int64_t)blockIdx.x * 4 + wave;
    const int64_t ks0 = w * per, ks1 = min(ksteps, ks0 + per);
    // n > 64: the (n, n) result is produced in 64 x 64 quadrants, one per blockIdx.z
    const int ioff = (int)(blockIdx.z >> 1) * 64, aoff = (int)(blockIdx.z & 1) * 64;
    d4 acc[NT][NT];
#pragma unroll
    for (int ti = 0; ti < NT; ++ti)
#pragma unroll
        for (int ta = 0; ta < NT; ++ta) acc[ti][ta] = (d4){0.0, 0.0, 0.0, 0.0};
    for (int64_t ks = ks0; ks < ks1; ++ks) {
        const int64_t k = ks * 4 + l4;
        const bool kok = k < ktot;
        double af[NT], bf[NT];
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            const int c = t * 16 + l15;
            af[t] = (kok && ioff + c < n) ? GsT[k * n + ioff + c] : 0.0;
            bf[t] = (kok && aoff + c < n) ? K3[k * n + aoff + c] : 0.0;
        }
#pragma unroll
        for (int ti = 0; ti < NT; ++ti)
#pragma unroll
            for (int ta = 0; ta < NT; ++ta) acc[ti][ta] = mfma_f64(af[ti], bf[ta], acc[ti][ta]);
    }
#pragma unroll
    for (int ti = 0; ti < NT; ++ti)
#pragma unroll
        for (int ta = 0; ta < NT; ++ta)
#pragma unroll
            for (int r = 0; r < 4; ++r) red[wave][ti * 16 + l4 + 4 * r][ta * 16 + l15] = acc[ti][ta][r];
    __syncthreads();
    double *dst = partial + (int64_t)blockIdx.x * n * n;
    for (int idx = threadIdx.x; idx < NT * 16 * NT * 16; idx += 256) {
        const int il = idx / (NT * 16), al = idx % (NT * 16);
        const int i = ioff + il, a = aoff + al;
        if (i < n && a < n) dst[(int64_t)i * n + a] = (red[0][il][al] + red[1][il][al]) + (red[2][il][al] + red[3][il][al]);
    }
}

// Same contraction with the first operand given as SB[i][k] (row i = n^3 contiguous doubles), i.e.
// partial[slab][i][a] = sum_{k in slab} SB[i][k] * K3[k][a]: for a fully symmetric 2-RDM the transposed operand GsT
// of the general path is SB itself read row-wise.  A lane fetches two consecutive k of "its" row i with one 16-byte
// load and feeds them to two MFMAs (the K slot of a lane can be any k as long as both operands agree).
template <int NT>
__global__ __launch_bounds__(256) void y2_sb_kernel(const double *__restrict__ SB, const double *__restrict__ K3,
                                                    int n, int64_t ktot, double *__restrict__ partial, int64_t sws) {
    __shared__ double red[4][NT * 16][NT * 16 + 1];
    SB += (int64_t)blockIdx.y * sws;
    K3 += (int64_t)blockIdx.y * sws;
    partial += (int64_t)blockIdx.y * sws;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int l15 = lane & 15, l4 = lane >> 4;
    const int64_t ksteps = (ktot + 7) / 8;
    const int64_t nw = (int64_t)gridDim.x * 4;
    const int64_t per = (ksteps + nw - 1) / nw;
    const int64_t w = (int64_t)blockIdx.x * 4 + wave;
    const int64_t ks0 = w * per, ks1 = min(ksteps, ks0 + per);
    const bool even = (ktot & 1) == 0;  // rows of SB start 16-byte aligned
    // n > 64: the (n, n) result is produced in 64 x 64 quadrants, one per blockIdx.z
    const int ioff = (int)(blockIdx.z >> 1) * 64, aoff = (int)(blockIdx.z & 1) * 64;
    const double *__restrict__ arow[NT];
    bool cok[NT], bok[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        const int c = t * 16 + l15;
        cok[t] = ioff + c < n;
        bok[t] = aoff + c < n;
        arow[t] = SB + (int64_t)(cok[t] ? ioff + c : 0) * ktot;
    }
    d4 acc[NT][NT];
#pragma unroll
    for (int ti = 0; ti < NT; ++ti)
#pragma unroll
        for (int ta = 0; ta < NT; ++ta) acc[ti][ta] = (d4){0.0, 0.0, 0.0, 0.0};
    for (int64_t ks = ks0; ks < ks1; ++ks) {
        const int64_t k = ks * 8 + 2 * l4;
        const bool k0ok = k < ktot, k1ok = k + 1 < ktot;
        double2 af[NT];
        double b0[NT], b1[NT];
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            if (even)
                af[t] = (cok[t] && k0ok) ? *reinterpret_cast<const double2 *>(arow[t] + k) : make_double2(0.0, 0.0);
            else
                af[t] = make_double2((cok[t] && k0ok) ? arow[t][k] : 0.0, (cok[t] && k1ok) ? arow[t][k + 1] : 0.0);
            const int c = aoff + t * 16 + l15;
            b0[t] = (bok[t] && k0ok) ? K3[k * n + c] : 0.0;
            b1[t] = (bok[t] && k1ok) ? K3[(k + 1) * n + c] : 0.0;
        }
#pragma unroll
        for (int ti = 0; ti < NT; ++ti)
#pragma unroll
            for (int ta = 0; ta < NT; ++ta) acc[ti][ta] = mfma_f64(af[ti].x, b0[ta], acc[ti][ta]);
#pragma unroll
        for (int ti = 0; ti < NT; ++ti)
#pragma unroll
            for (int ta = 0; ta < NT; ++ta) acc[ti][ta] = mfma_f64(af[ti].y, b1[ta], acc[ti][ta]);
    }
#pragma unroll
    for (int ti = 0; ti < NT; ++ti)
#pragma unroll
        for (int ta = 0; ta < NT; ++ta)
#pragma unroll
            for (int r = 0; r < 4; ++r) red[wave][ti * 16 + l4 + 4 * r][ta * 16 + l15] = acc[ti][ta][r];
    __syncthreads();
    double *dst = partial + (int64_t)blockIdx.x * n * n;
    for (int idx = threadIdx.x; idx < NT * 16 * NT * 16; idx += 256) {
        const int il = idx / (NT * 16), al = idx % (NT * 16);
        const int i = ioff + il, a = aoff + al;
        if (i < n && a < n) dst[(int64_t)i * n + a] = (red[0][il][al] + red[1][il][al]) + (red[2][il][al] + red[3][il][al]);
    }
}

// ------------------------------------------------------------------ Y2 with the half-transformed integrals recomputed
// The reference's Y2 = sum K3 . Gamma~ needs K3p[j][v][a] = mult(v) (M1_v X)[a][j]; rounds 1-2 had the second pair step of
// the energy phase store it (107 MB per 32 geometries at N = 30: +18 us there) and a split-K contraction read it back
// (y2_pairs_kernel, removed in round 4).  M1_v -- row v of
// the dense (pair, pair) intermediate of the FIRST pair step, a symmetric N x N matrix -- is 16x smaller, and
// SB[tri(i,j)][v] = SB[v][tri(i,j)] is a contiguous row of the symmetric SB as well, so one wave per pair v does
//   H^T = X^T M1_v            (32 MFMAs at N <= 32; X fragments as A operand, the fragments of the symmetric M1_v as B)
//   Y  += mult(v) T_v H^T     (32 MFMAs; T_v = row v of SB as A operand, the accumulator tiles of H^T as B operand:
//                              row 4 kk + (l >> 4) of H^T lives in register kk % 4 of its row tile kk / 4)
// with both rows fetched like the operand rows of the pair transform (coalesced 16-byte loads, wave-private LDS row,
// lane-constant triangle offsets) one pair ahead.  No stage, no stores but the (N, N) partial of the workgroup.
template <int NPAD>
__global__ __launch_bounds__(256) void y2_fused_kernel(const double *__restrict__ SB, const double *__restrict__ M1,
                                                       const double *__restrict__ X, int64_t sX, int n,
                                                       double *__restrict__ partial, int64_t sws, int tiles_per_wg,
                                                       int ppt) {
    constexpr int KS = NPAD / 4;
    constexpr int NT = NPAD / 16;
    constexpr int RAWN = (NPAD * (NPAD + 1) / 2 + 1 + 127) / 128;
    extern __shared__ __align__(16) double sm[];
    const int npairs = n * (n + 1) / 2, ld = pair_ld(n);   // both operands are dense (pair, pair) forms of the pipeline
    const int64_t g = blockIdx.y;
    SB += g * sws;
    M1 += g * sws;
    X += g * sX;
    partial += g * sws;
    const int ntiles = (npairs + ppt - 1) / ppt;   // ppt = 8 or 4 pairs per tile (two / one matrix per wave)
    const int t_begin = blockIdx.x * tiles_per_wg, t_end = min(ntiles, t_begin + tiles_per_wg);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int l15 = lane & 15, l4 = lane >> 4;
    double *rowM = sm + wave * kPtRowLen;             // the wave's two operand rows
    double *rowT = sm + (4 + wave) * kPtRowLen;
    double *red = sm;                                 // [4][NPAD][NPAD + 1], over the rows once they are done with
    d4 yacc[NT][NT];
#pragma unroll
    for (int ti = 0; ti < NT; ++ti)
#pragma unroll
        for (int ta = 0; ta < NT; ++ta) yacc[ti][ta] = (d4){0.0, 0.0, 0.0, 0.0};
    if (t_begin < t_end) {
        const int niter = (ppt / 4) * (t_end - t_begin);
        int foff[NT][KS];   // fragment (rt, kk) of a symmetric n x n matrix in its packed row
#pragma unroll
        for (int rt = 0; rt < NT; ++rt)
#pragma unroll
            for (int kk = 0; kk < KS; ++kk) {
                const int r = rt * 16 + l15, s = 4 * kk + l4;
                const int hi = s > r ? s : r, lo = s > r ? r : s;
                foff[rt][kk] = (r < n && s < n) ? hi * (hi + 1) / 2 + lo : kPtRawMax * 128;   // else: a zero slot
            }
        if (lane < 4) {
            rowM[kPtRawMax * 128 + lane] = 0.0;
            rowT[kPtRawMax * 128 + lane] = 0.0;
        }
        d2 rawM[RAWN], rawT[RAWN];
        auto fetch = [&](const double *base, int e, d2 (&raw)[RAWN]) -> int {
            const double *row = base + (int64_t)(e < npairs ? e : 0) * ld;
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
        auto park = [&](double *row, const d2 (&raw)[RAWN]) {
#pragma unroll
            for (int u = 0; u < RAWN; ++u) *reinterpret_cast<d2 *>(row + 128 * u + 2 * lane) = raw[u];
        };
        auto is_diag = [&](int x) -> bool {
            const int r = tri_row_small(x);
            return x == r * (r + 3) / 2;
        };
        const int e0 = ppt * t_begin + wave;   // this wave's pair of iteration i: e0 + 4 i
        int dM = fetch(M1, e0, rawM), dT = fetch(SB, e0, rawT);
        double xf[KS][NT];
#pragma unroll
        for (int kk = 0; kk < KS; ++kk)
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                const int d = 4 * kk + l4, c = t * 16 + l15;
                const bool ok = d < n && c < n;
                const double v = X[ok ? d * n + c : 0];
                xf[kk][t] = ok ? v : 0.0;
            }
        double mf[NT][KS], tf[NT][KS];
        park(rowM, rawM);
#pragma unroll
        for (int rt = 0; rt < NT; ++rt)
#pragma unroll
            for (int kk = 0; kk < KS; ++kk) mf[rt][kk] = rowM[foff[rt][kk] + dM];
        dM = fetch(M1, e0 + 4, rawM);
        // Iteration i: the T row (fetched one iteration ago) goes to LDS and comes back as fragments behind the MFMAs of
        // the H^T phase (which does not use them) and the next T row is requested; the next M row goes to LDS and comes
        // back behind the MFMAs of the Y phase (which does not use the M fragments), then the M row after that is
        // requested.  No branches in the body: idle slots of the last tile run on row 0 with multiplicity 0.
        for (int i = 0; i < niter; ++i) {
            const int e = e0 + 4 * i;
            const double km = e < npairs ? (is_diag(e) ? 1.0 : 2.0) : 0.0;   // multiplicity of the pair (p,q)
            park(rowT, rawT);
#pragma unroll
            for (int rt = 0; rt < NT; ++rt)
#pragma unroll
                for (int kk = 0; kk < KS; ++kk) tf[rt][kk] = rowT[foff[rt][kk] + dT];
            dT = fetch(SB, e + 4, rawT);
            d4 hT[NT][NT];   // H^T = X^T M: tile (it, st) = rows s' of tile it, columns r of tile st
#pragma unroll
            for (int it = 0; it < NT; ++it)
#pragma unroll
                for (int st = 0; st < NT; ++st) hT[it][st] = (d4){0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int kk = 0; kk < KS; ++kk)
#pragma unroll
                for (int it = 0; it < NT; ++it)
#pragma unroll
                    for (int st = 0; st < NT; ++st) hT[it][st] = mfma_f64(xf[kk][it], mf[st][kk], hT[it][st]);
            park(rowM, rawM);
#pragma unroll
            for (int rt = 0; rt < NT; ++rt)
#pragma unroll
                for (int kk = 0; kk < KS; ++kk) mf[rt][kk] = rowM[foff[rt][kk] + dM];
            dM = fetch(M1, e + 8, rawM);
#pragma unroll
            for (int kk = 0; kk < KS; ++kk)
#pragma unroll
                for (int ti = 0; ti < NT; ++ti) {
                    const double tv = tf[ti][kk] * km;
#pragma unroll
                    for (int ta = 0; ta < NT; ++ta)
                        yacc[ti][ta] = mfma_f64(tv, hT[kk / 4][ta][kk % 4], yacc[ti][ta]);
                }
        }
    }
    // cross-wave sum (every workgroup writes its slab, workgroups without tiles a zero one)
    __syncthreads();
#pragma unroll
    for (int ti = 0; ti < NT; ++ti)
#pragma unroll
        for (int ta = 0; ta < NT; ++ta)
#pragma unroll
            for (int r = 0; r < 4; ++r)
                red[(wave * NPAD + ti * 16 + l4 + 4 * r) * (NPAD + 1) + ta * 16 + l15] = yacc[ti][ta][r];
    __syncthreads();
    double *dst = partial + (int64_t)blockIdx.x * n * n;
    for (int idx = threadIdx.x; idx < n * n; idx += 256) {
        const int i = idx / n, aa = idx % n;
        const int o = i * (NPAD + 1) + aa;
        constexpr int WS = NPAD * (NPAD + 1);
        dst[idx] = (red[o] + red[WS + o]) + (red[2 * WS + o] + red[3 * WS + o]);
    }
}

// 8-pair tiles per workgroup: 4 for batches (as the pair transform), 1 for a few geometries (enough workgroups for the
// chip), never more workgroups than the partial buffer has slabs
static int y2_fused_ppt(int count) { return count < 4 ? 4 : 8; }   // pairs per tile: 4 (one matrix per wave) for a few geometries
static int y2_fused_tiles(int n, int count) {
    const int ppt = y2_fused_ppt(count);
    const int ntiles = (n * (n + 1) / 2 + ppt - 1) / ppt;
    int t = count < 4 ? 1 : 4;
    while ((ntiles + t - 1) / t > y2_slab_capacity(n)) ++t;
    return t;
}
bool y2_fused_available(int n) { return n >= 1 && (n <= kPairTransformMaxN || y2_64_applicable(n)); }
int y2_fused_slabs(int n, int count) {
    if (y2_64_applicable(n)) return y2_64_slabs(n, count);
    const int ppt = y2_fused_ppt(count);
    const int ntiles = (n * (n + 1) / 2 + ppt - 1) / ppt, t = y2_fused_tiles(n, count);
    return (ntiles + t - 1) / t;
}
int launch_y2_fused(const double *SB, const double *M1, const double *X, int64_t sX, int n, double *partial,
                    int64_t sws, int count, hipStream_t st) {
    if (y2_64_applicable(n)) return launch_y2_64(SB, M1, X, sX, n, partial, sws, count, st);
    if (y2_dma_applicable(n))
        return launch_y2_dma(SB, M1, X, sX, n, partial, sws, count, y2_fused_slabs(n, count), y2_fused_tiles(n, count),
                             y2_fused_ppt(count), st);
    const dim3 grid((unsigned)y2_fused_slabs(n, count), (unsigned)count);
    const int npad = (n + 15) / 16 * 16;
    const size_t rows = sizeof(double) * (size_t)8 * kPtRowLen;
    if (npad == 16) {
        const size_t redb = sizeof(double) * 4 * 16 * 17;
        hipLaunchKernelGGL(y2_fused_kernel<16>, grid, dim3(256), rows > redb ? rows : redb, st, SB, M1, X, sX, n, partial,
                           sws, y2_fused_tiles(n, count), y2_fused_ppt(count));
    } else if (npad == 32) {
        const size_t redb = sizeof(double) * 4 * 32 * 33;
        hipLaunchKernelGGL(y2_fused_kernel<32>, grid, dim3(256), rows > redb ? rows : redb, st, SB, M1, X, sX, n, partial,
                           sws, y2_fused_tiles(n, count), y2_fused_ppt(count));
    } else {
        set_error("y2_fused: n=%d not supported (1..32)", n);
        return -1;
    }
    note_kernel(EVC_PROF_Y2, "y2_fused_kernel<%d>", npad);
    EVC_LAUNCH_CHECK("y2_fused");
    return 0;
}

int launch_y2_sb(const double *SB, const double *K3, int n, double *partial, int64_t sws, int count, hipStream_t st) {
    const int64_t ktot = (int64_t)n * n * n;
    const int nt = (n + 15) / 16;
    const dim3 grid((unsigned)y2_slab_count(), (unsigned)count, nt > 4 ? 4u : 1u);   // n > 64: four 64 x 64 quadrants
    switch (nt) {
        case 1: hipLaunchKernelGGL(y2_sb_kernel<1>, grid, dim3(256), 0, st, SB, K3, n, ktot, partial, sws); break;
        case 2: hipLaunchKernelGGL(y2_sb_kernel<2>, grid, dim3(256), 0, st, SB, K3, n, ktot, partial, sws); break;
        case 3: hipLaunchKernelGGL(y2_sb_kernel<3>, grid, dim3(256), 0, st, SB, K3, n, ktot, partial, sws); break;
        case 4: case 5: case 6: case 7: case 8:
            hipLaunchKernelGGL(y2_sb_kernel<4>, grid, dim3(256), 0, st, SB, K3, n, ktot, partial, sws); break;
        default: set_error("y2: n=%d not supported by the gradient path (1..128)", n); return -1;
    }
    EVC_LAUNCH_CHECK("y2_sb");
    return 0;
}

int launch_y2(const double *GsT, const double *K3, int n, double *partial, int64_t sws, int count, hipStream_t st) {
    const int64_t ktot = (int64_t)n * n * n;
    const int nt = (n + 15) / 16;
    const dim3 grid((unsigned)y2_slab_count(), (unsigned)count, nt > 4 ? 4u : 1u);   // n > 64: four 64 x 64 quadrants
    switch (nt) {
        case 1: hipLaunchKernelGGL(y2_kernel<1>, grid, dim3(256), 0, st, GsT, K3, n, ktot, partial, sws); break;
        case 2: hipLaunchKernelGGL(y2_kernel<2>, grid, dim3(256), 0, st, GsT, K3, n, ktot, partial, sws); break;
        case 3: hipLaunchKernelGGL(y2_kernel<3>, grid, dim3(256), 0, st, GsT, K3, n, ktot, partial, sws); break;
        case 4: case 5: case 6: case 7: case 8:
            hipLaunchKernelGGL(y2_kernel<4>, grid, dim3(256), 0, st, GsT, K3, n, ktot, partial, sws); break;
        default: set_error("y2: n=%d not supported by the gradient path (1..128)", n); return -1;
    }
    EVC_LAUNCH_CHECK("y2");
    return 0;
}

// ------------------------------------------------------------------ ip1 contraction + dhcore dots + slab sums
// t2part[(m*3+x)*nchunk + ch] = sum_{e in chunk ch} ip1[x][m][e] * GsAO[m][e],  e = (b,c,d)
// GsAO[m,b,c,d] = G[m,b,c,d] + G[b,m,d,c] + G[c,d,m,b] + G[d,c,b,m]   (G = 2-RDM in the AO basis)
// Blocks [nb1, nb1 + 3A): term3[A*3+x] = sum_ab dhcore[A,x,a,b] * Pao[a,b].
}  // namespace evc
