// The symmetric AO-side pipeline for 32 < n <= 64 orbitals (the reference's cc-pVTZ water runs: n = 58,
// scripts/MD/H2O/md_H2O_vtz_CAS_continuation.py:25-33): the fused pair step of the four-index rotations
// (electron_integral_utils.py:136, ab_initio_gradients_loewdin.py:224-232) and the Y2 contraction with the
// half-transformed integrals recomputed (ab_initio_gradients_loewdin.py:210-222), on 64 x 64 operand matrices.
//
// Same mathematics and the same dense (pair, pair) data forms as the n <= 32 kernels (transform.hip, pair_dma.hip, y2.hip):
// per leading AO pair v = (p >= q) the symmetric n x n matrix M_v is one row of a dense (pair, pair) matrix,
//     pt64:  R_v = C^T M_v C   (H = M_v C: 256 MFMAs; R = C^T H, lower-triangle tiles only: 160 MFMAs; the accumulator
//                               tiles of H are the B operand of the second product)
//     y2_64: Y += mult(v) T_v (C^T M1_v)   (2 x 256 MFMAs)
// What differs is the budget: sixteen 16 x 16 accumulator tiles are 128 registers, so
//   * pt64 keeps H in registers and nothing else: C (32 KB, the 16-double halves of odd rows swapped: the fragment rows of
//     lane groups l4 and l4 + 1 fall on disjoint bank halves) and the operand row live in LDS, the fragment offsets
//     tri(max(r,s), min(r,s)) of a lane -- the same for every matrix -- in an LDS table (one add per fragment instead of
//     the index arithmetic), and the result tiles of R are filed, tile row by tile row, over the wave's own operand row,
//     which is dead after the first product: four waves = four consecutive pairs per workgroup = 32-byte runs of the
//     output columns at write-out;
//   * y2_64 lets the four waves of a workgroup work on ONE pair: wave w owns the column block 16 w ... 16 w + 15 of
//     C^T M1_v (four tiles) and of Y (four accumulator tiles for the whole launch), so no wave holds more than eight
//     tiles and nothing is summed across waves.
// One workgroup per CU (115 KB of LDS); a workgroup of pt64 takes `tiles_per_wg` consecutive tiles of four pairs.
#include <stdlib.h>

#include "common.hpp"
#include "kernels.hpp"

namespace evc {

typedef double d4 __attribute__((ext_vector_type(4)));
typedef double d2 __attribute__((ext_vector_type(2)));

namespace {

constexpr int kP64MaxPairs = 64 * 65 / 2;
constexpr int kP64Raw = (kP64MaxPairs + 1 + 127) / 128;   // double2 per lane of one operand row: 17

__device__ __forceinline__ d4 mfma64(double a, double b, d4 c) {
    return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ int c64(int row, int col) { return row * 64 + (col ^ ((row & 1) << 4)); }

// doubles of one wave's row / stage buffer: the row, two zero slots behind it (fragments outside the matrix; two because
// of the 16-byte window below), a length = 8 mod 32 (the four buffers then sit on disjoint quarter-banks for the
// write-out, which reads slot-fastest)
__host__ __device__ inline int p64_rowlen(int npairs) {
    int r = npairs + 4;
    r += (8 - (r & 31) + 32) & 31;
    return r;
}

// C (or its transpose) into LDS, zero padded to 64 x 64; the lane-constant fragment table; zero slots
__device__ __forceinline__ void p64_setup(const double *__restrict__ C, int n, bool ct, double *Cs, int *ftab, int npairs) {
    for (int idx = threadIdx.x; idx < 64 * 64; idx += 256) {
        const int d = idx >> 6, c = idx & 63;
        double v = 0.0;
        if (d < n && c < n) v = ct ? C[c * n + d] : C[d * n + c];
        Cs[c64(d, c)] = v;
    }
    // fragment (rt, kk) of lane (l15, l4): element (r = 16 rt + l15, s = 4 kk + l4) of the packed symmetric row
    for (int idx = threadIdx.x; idx < 64 * 64; idx += 256) {
        const int f = idx >> 6, lane = idx & 63, rt = f >> 4, kk = f & 15;
        const int r = 16 * rt + (lane & 15), s = 4 * kk + (lane >> 4);
        const int hi = s > r ? s : r, lo = s > r ? r : s;
        ftab[idx] = 8 * ((r < n && s < n) ? hi * (hi + 1) / 2 + lo : npairs + 1);   // (npairs + 1 [+ 1]: the zero slots)
    }
}

// one packed row (npairs doubles at `row`, 8-byte aligned) -> the wave's LDS buffer through a 16-byte aligned window;
// returns the window shift d (0 / 1): element k of the row sits at buf[k + d]
__device__ __forceinline__ int p64_fetch(const double *__restrict__ row, int npairs, int lane, d2 (&raw)[kP64Raw]) {
    const int d = (int)((reinterpret_cast<uintptr_t>(row) >> 3) & 1);
    const double *w0 = row - d;
    const int lim = npairs + d;
#pragma unroll
    for (int u = 0; u < kP64Raw; ++u) {
        const int j = 128 * u + 2 * lane;
        d2 v = {0.0, 0.0};
        if (j + 1 < lim) v = *reinterpret_cast<const d2 *>(w0 + j);
        else if (j < lim) v.x = w0[j];
        raw[u] = v;
    }
    return d;
}
__device__ __forceinline__ void p64_park(double *buf, int npairs, int lane, const d2 (&raw)[kP64Raw]) {
#pragma unroll
    for (int u = 0; u < kP64Raw; ++u) {
        const int j = 128 * u + 2 * lane;
        if (j < npairs + 3) *reinterpret_cast<d2 *>(buf + j) = raw[u];   // (beyond the row: zeros, the zero slots included)
    }
}

__device__ __forceinline__ bool p64_is_diag(int x) {
    const int r = tri_row_small(x);
    return x == r * (r + 3) / 2;
}

}  // namespace

// ------------------------------------------------------------------ pair step
// PairTransformArgs as for the n <= 32 kernels with lead_sym = in_lower = rs_lower = in_pairs = 1: the operand is the
// dense (pair, pair) matrix in[tri(p,q)][tri(r,s)] at pitch in_ld (0: n(n+1)/2); the result goes either to the dense
// out[tri(r',s')][tri(p,q)] at pitch out_ld (out_pairs) or, with its multiplicities, to the 8-fold compressed vector.
__global__ __launch_bounds__(256, 1) void pt64_kernel(PairTransformArgs a) {
    extern __shared__ __align__(16) double sm[];
    const int n = a.n, npairs = n * (n + 1) / 2, rl = p64_rowlen(npairs);
    const int ild = a.in_ld ? a.in_ld : npairs, old_ = a.out_ld ? a.out_ld : npairs;
    double *Cs = sm;                                  // 64 x 64
    int *ftab = reinterpret_cast<int *>(Cs + 4096);   // 64 fragments x 64 lanes
    double *bufs = Cs + 4096 + 2048;                  // 4 x rl
    const int64_t g = blockIdx.y;
    const double *__restrict__ in = a.in + g * a.sin;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, l15 = lane & 15, l4 = lane >> 4;
    const int ntiles = (npairs + 3) / 4;
    const int t_begin = blockIdx.x * a.tiles_per_wg, t_end = min(ntiles, t_begin + a.tiles_per_wg);
    if (t_begin >= t_end) return;
    double *buf = bufs + wave * rl;
    p64_setup(a.C + g * a.sC, n, a.ct != 0, Cs, ftab, npairs);
    d2 raw[kP64Raw];
    int e = 4 * t_begin + wave;
    int d = e < npairs ? p64_fetch(in + (int64_t)e * ild, npairs, lane, raw) : 0;
    __syncthreads();
    for (int t = t_begin; t < t_end; ++t) {
        e = 4 * t + wave;
        const bool have = e < npairs;   // wave-uniform
        if (have) {
            p64_park(buf, npairs, lane, raw);
            const char *rb = reinterpret_cast<const char *>(buf + d);
            // H = M C
            d4 h[4][4];
#pragma unroll
            for (int rt = 0; rt < 4; ++rt)
#pragma unroll
                for (int st = 0; st < 4; ++st) h[rt][st] = (d4){0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int kk = 0; kk < 16; ++kk) {
                double mf[4], xf[4];
#pragma unroll
                for (int rt = 0; rt < 4; ++rt)
                    mf[rt] = *reinterpret_cast<const double *>(rb + ftab[(rt * 16 + kk) * 64 + lane]);
#pragma unroll
                for (int st = 0; st < 4; ++st) xf[st] = Cs[c64(4 * kk + l4, 16 * st + l15)];
#pragma unroll
                for (int rt = 0; rt < 4; ++rt)
#pragma unroll
                    for (int st = 0; st < 4; ++st) h[rt][st] = mfma64(mf[rt], xf[st], h[rt][st]);
            }
            // the next row of this wave is requested now: it arrives behind the second product
            const int en = e + 4;
            const bool more = t + 1 < t_end && en < npairs;
            int dn = 0;
            if (more) dn = p64_fetch(in + (int64_t)en * ild, npairs, lane, raw);
            // R = C^T H, tile rows (0, 3) and (1, 2) together (five accumulator chains each); the tiles go over the
            // operand row, which nothing reads any more
#pragma unroll
            for (int pass = 0; pass < 2; ++pass) {
                const int ia = pass == 0 ? 0 : 1, ib = pass == 0 ? 3 : 2;
                d4 na[4], nb[4];
#pragma unroll
                for (int st = 0; st < 4; ++st) na[st] = nb[st] = (d4){0.0, 0.0, 0.0, 0.0};
#pragma unroll
                for (int kk = 0; kk < 16; ++kk) {
                    const double xa = Cs[c64(4 * kk + l4, 16 * ia + l15)], xb = Cs[c64(4 * kk + l4, 16 * ib + l15)];
#pragma unroll
                    for (int st = 0; st < 4; ++st) {
                        if (st <= ia) na[st] = mfma64(xa, h[kk >> 2][st][kk & 3], na[st]);
                        if (st <= ib) nb[st] = mfma64(xb, h[kk >> 2][st][kk & 3], nb[st]);
                    }
                }
#pragma unroll
                for (int st = 0; st < 4; ++st)
#pragma unroll
                    for (int reg = 0; reg < 4; ++reg) {
                        const int s2 = 16 * st + l15;
                        if (st <= ia) {
                            const int r2 = 16 * ia + l4 + 4 * reg;
                            if (s2 <= r2 && r2 < n) buf[r2 * (r2 + 1) / 2 + s2] = na[st][reg];
                        }
                        if (st <= ib) {
                            const int r2 = 16 * ib + l4 + 4 * reg;
                            if (s2 <= r2 && r2 < n) buf[r2 * (r2 + 1) / 2 + s2] = nb[st][reg];
                        }
                    }
            }
            d = dn;
        }
        __syncthreads();
        // write-out: lane (u, slot) -- slot fastest: the four results of a result pair u are a 32-byte run of its row
        const int slot = threadIdx.x & 3, ev = 4 * t + slot;
        if (ev < npairs) {
            const double *src = bufs + slot * rl;
            if (a.packed) {
                // 8-fold compressed vector: u >= v, times the multiplicities of both pairs (and diag_mult on u == v)
                double *pk = a.packed + g * a.spacked;
                const double mq = p64_is_diag(ev) ? 1.0 : 2.0;
                for (int u = ev + (int)(threadIdx.x >> 2); u < npairs; u += 64) {
                    const double f = (u == ev ? a.diag_mult : 1.0) * mq * (p64_is_diag(u) ? 1.0 : 2.0);
                    pk[(int64_t)u * (u + 1) / 2 + ev] = src[u] * f;
                }
            } else {
                double *out = a.out + g * a.sout;
                for (int u = threadIdx.x >> 2; u < npairs; u += 64) out[(int64_t)u * old_ + ev] = src[u];
            }
        }
        if (a.packed && blockIdx.x == 0 && t == t_begin) {   // zero the padding [M, packed_len) once per geometry
            double *pk = a.packed + g * a.spacked;
            const int64_t M = (int64_t)npairs * (npairs + 1) / 2;
            for (int64_t m = M + threadIdx.x; m < a.packed_len; m += 256) pk[m] = 0.0;
        }
        __syncthreads();
    }
}

bool pair64_applicable(const PairTransformArgs &a) {
    return a.n > kPairTransformMaxN && a.n <= 64 && a.lead_sym && a.in_lower && a.rs_lower && a.in_pairs &&
           !a.k3 && ((a.packed && a.sym8 && !a.out) || (a.out && a.out_pairs && !a.packed));
}

size_t pair64_lds_bytes(int n) {
    return sizeof(double) * (4096 + 2048 + (size_t)4 * p64_rowlen(n * (n + 1) / 2));
}

int launch_pair_transform64(const PairTransformArgs &a_in, int count, hipStream_t st) {
    PairTransformArgs a = a_in;
    const int npairs = a.n * (a.n + 1) / 2, ntiles = (npairs + 3) / 4;
    // one resident round of the chip (one workgroup per CU)
    int tpw = (int)ceil_div((int64_t)ntiles * count, (int64_t)250);
    if (tpw < 1) tpw = 1;
    a.tiles_per_wg = tpw;
    static LdsAttr attr;
    if (int rc = allow_dynamic_lds(pt64_kernel, attr, 160 * 1024, "pt64")) return rc;
    hipLaunchKernelGGL(pt64_kernel, dim3((unsigned)ceil_div(ntiles, tpw), (unsigned)count), dim3(256),
                       pair64_lds_bytes(a.n), st, a);
    note_kernel(EVC_PROF_PAIR_TRANSFORM, "pt64_kernel<%d>", a.packed ? 1 : 0);
    EVC_LAUNCH_CHECK("pt64");
    return 0;
}

// ------------------------------------------------------------------ Y2
// partial[slab][i][a] (+ g * sws): slab = workgroup; Y[i][a] = sum_v mult(v) sum_k T_v[i][k] (C^T M1_v)[k][a], the pairs v
// dealt to the workgroups in contiguous ranges.  SB, M1: dense (pair, pair) matrices at pitch pair_ld(n).
__global__ __launch_bounds__(256, 1) void y2_64_kernel(const double *__restrict__ SB, const double *__restrict__ M1,
                                                       const double *__restrict__ X, int64_t sX, int n,
                                                       double *__restrict__ partial, int64_t sws, int pairs_per_wg) {
    extern __shared__ __align__(16) double sm[];
    const int npairs = n * (n + 1) / 2, ld = pair_ld(n), rl = p64_rowlen(npairs);
    double *Cs = sm;
    int *ftab = reinterpret_cast<int *>(Cs + 4096);
    double *rows = Cs + 4096 + 2048;   // [buffer 2][M, T][rl]
    const int64_t g = blockIdx.y;
    SB += g * sws;
    M1 += g * sws;
    partial += g * sws + (int64_t)blockIdx.x * n * n;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, l15 = lane & 15, l4 = lane >> 4;
    const int e_begin = blockIdx.x * pairs_per_wg, e_end = min(npairs, e_begin + pairs_per_wg);
    d4 yacc[4];
#pragma unroll
    for (int ti = 0; ti < 4; ++ti) yacc[ti] = (d4){0.0, 0.0, 0.0, 0.0};
    if (e_begin < e_end) {
        p64_setup(X + g * sX, n, false, Cs, ftab, npairs);
        // the two rows of a pair are fetched by the two halves of the workgroup (waves 0, 1: M1_v; waves 2, 3: T_v), each
        // lane a share of its row; 16-byte loads (rows of the pipeline's dense forms start on multiples of 128 bytes)
        const int half = threadIdx.x >> 7, ht = threadIdx.x & 127;
        const double *src = half ? SB : M1;
        constexpr int kShare = (kP64MaxPairs + 2 + 255) / 256;   // double2 per thread: 9
        d2 raw[kShare];
        auto fetch = [&](int e) {
            const double *row = src + (int64_t)e * ld;
#pragma unroll
            for (int u = 0; u < kShare; ++u) {
                const int j = 2 * (ht + 128 * u);
                d2 v = {0.0, 0.0};
                if (j + 1 < npairs) v = *reinterpret_cast<const d2 *>(row + j);
                else if (j < npairs) v.x = row[j];
                raw[u] = v;
            }
        };
        auto park = [&](int b) {
            double *dst = rows + (size_t)(2 * b + half) * rl;
#pragma unroll
            for (int u = 0; u < kShare; ++u) {
                const int j = 2 * (ht + 128 * u);
                if (j < npairs + 2) *reinterpret_cast<d2 *>(dst + j) = raw[u];
            }
        };
        fetch(e_begin);
        park(0);
        if (e_begin + 1 < e_end) fetch(e_begin + 1);
        __syncthreads();
        for (int e = e_begin; e < e_end; ++e) {
            const int b = (e - e_begin) & 1;
            const char *rm = reinterpret_cast<const char *>(rows + (size_t)(2 * b) * rl);
            const char *rt_ = reinterpret_cast<const char *>(rows + (size_t)(2 * b + 1) * rl);
            const double km = p64_is_diag(e) ? 1.0 : 2.0;
            // the rows of pair e + 1 (in registers since the previous iteration) go to the other buffer: nobody reads it
            // (the barrier at the end of the previous iteration), then the rows of pair e + 2 are requested
            if (e + 1 < e_end) park(b ^ 1);
            if (e + 2 < e_end) fetch(e + 2);
            // H^T[:, block `wave`] = C^T M1_v[:, block]: tile it = rows s' of tile it, columns r = 16 wave + l15
            d4 hT[4];
#pragma unroll
            for (int it = 0; it < 4; ++it) hT[it] = (d4){0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int kk = 0; kk < 16; ++kk) {
                // (wave-dependent fragment: table row 16 wave + kk, not an immediate)
                const double mf = *reinterpret_cast<const double *>(rm + ftab[(wave * 16 + kk) * 64 + lane]);
#pragma unroll
                for (int it = 0; it < 4; ++it) hT[it] = mfma64(Cs[c64(4 * kk + l4, 16 * it + l15)], mf, hT[it]);
            }
            // Y[:, block] += mult T_v H^T[:, block]
#pragma unroll
            for (int kk = 0; kk < 16; ++kk)
#pragma unroll
                for (int ti = 0; ti < 4; ++ti) {
                    const double tv = *reinterpret_cast<const double *>(rt_ + ftab[(ti * 16 + kk) * 64 + lane]) * km;
                    yacc[ti] = mfma64(tv, hT[kk >> 2][kk & 3], yacc[ti]);
                }
            __syncthreads();
        }
    }
    // every workgroup writes its slab (workgroups without pairs a zero one)
#pragma unroll
    for (int ti = 0; ti < 4; ++ti)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int i = 16 * ti + l4 + 4 * r, aa = 16 * wave + l15;
            if (i < n && aa < n) partial[(int64_t)i * n + aa] = yacc[ti][r];
        }
}

bool y2_64_applicable(int n) { return n > kPairTransformMaxN && n <= 64; }

int y2_64_slabs(int n, int count) {
    // one resident round of the chip, never more slabs than the partial buffer holds
    const int npairs = n * (n + 1) / 2;
    int wgs = count >= 250 ? 1 : 250 / count;
    const int cap = y2_slab_capacity(n);
    if (wgs > cap) wgs = cap;
    if (wgs > npairs) wgs = npairs;
    const int ppw = (npairs + wgs - 1) / wgs;
    return (npairs + ppw - 1) / ppw;
}

int launch_y2_64(const double *SB, const double *M1, const double *X, int64_t sX, int n, double *partial, int64_t sws,
                 int count, hipStream_t st) {
    const int npairs = n * (n + 1) / 2, slabs = y2_64_slabs(n, count), ppw = (npairs + slabs - 1) / slabs;
    const size_t lds = sizeof(double) * (4096 + 2048 + (size_t)4 * p64_rowlen(npairs));
    static LdsAttr attr;
    if (int rc = allow_dynamic_lds(y2_64_kernel, attr, 160 * 1024, "y2_64")) return rc;
    hipLaunchKernelGGL(y2_64_kernel, dim3((unsigned)slabs, (unsigned)count), dim3(256), lds, st, SB, M1, X, sX, n,
                       partial, sws, ppw);
    note_kernel(EVC_PROF_Y2, "y2_64_kernel");
    EVC_LAUNCH_CHECK("y2_64");
    return 0;
}

}  // namespace evc
