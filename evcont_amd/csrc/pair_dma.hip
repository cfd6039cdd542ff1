// The fully symmetric pair step of the compressed layout's pipeline (dense (pair, pair) operand in, dense (pair, pair)
// result out; electron_integral_utils.py:136 / ab_initio_gradients_loewdin.py:224-232 as two-sided rotations
// N_e = X^T M_e X of the n(n+1)/2 symmetric matrices M_e), 16 < n <= 30, with NO vector instruction beside the MFMAs
// in its steady state.
//
// Why (round 4): an FP64 MFMA and every other vector instruction share one pipe on gfx950 (profiles/
// mfma_f64_coissue.txt): pt_pipe_kernel (transform.hip) issued ~250 vector instructions per matrix -- address
// arithmetic, predicates, selects, multiplicities -- beside its 56 MFMAs and kept the matrix pipe 49 % busy.  Loads,
// stores, LDS and scalar instructions DO issue while an MFMA executes.  Here everything per matrix is one of those:
//   * the operand row (one packed symmetric matrix, n(n+1)/2 doubles) comes by LDS-DMA (global_load_lds_dwordx4,
//     scalar row base + four lane-constant offsets) into the wave's LDS row; the fragments are read from it at sixteen
//     lane-constant addresses; the wait for the DMA is a counted vmcnt (the stores issued behind it are counted per
//     wave on the host side of the loop, all conditions wave-uniform);
//   * the stage is slot-major, [buffer][slot][row u] with a slot pitch of 32 bytes mod 128: the 12 stage writes of a
//     matrix are conflict free at lane-constant addresses, buffer and slot are IMMEDIATE offsets (the loop is unrolled
//     over the four (matrix parity, tile parity) phases), the write-out reads its two columns with two ds_read_b64,
//     conflict free as well;
//   * the write-out store is `scalar base + lane-constant offset`; rows are written in whole 16-row groups per wave
//     (the result buffer has pair_ld(n) rows), columns in whole tiles of 8 (its pitch is pair_ld(n)): no predicates;
//   * multiplicities are left to the consumers (the HBM-bound int2e_ip1 dot weighs its operand itself).
// The pipeline around these pieces is pt_pipe_kernel's: one matrix per iteration, result i-1 staged during the H phase
// of matrix i, one barrier per tile of eight leading pairs, tile j written out during the N phases of the two
// iterations behind its barrier, two workgroups per CU.
#include <stdlib.h>

#include "common.hpp"
#include "kernels.hpp"

namespace evc {

namespace {

typedef double d4 __attribute__((ext_vector_type(4)));
typedef double d2 __attribute__((ext_vector_type(2)));

constexpr int kPdMaxN = 30;
constexpr unsigned kPdRaw = 4;                    // 1 KiB pieces per operand row (465 + 1 doubles at most)
constexpr unsigned kPdRB = kPdRaw * 1024 + 64;    // bytes per wave: the row and a zero slot behind it
constexpr unsigned kPdStage = 4 * kPdRB;          // byte offset of the stage (a multiple of 256)
constexpr unsigned kPdP = 3872;                   // slot pitch of the stage: >= 8 (465 + 16) bytes, = 32 mod 128
constexpr unsigned kPdTab = kPdStage + 16 * kPdP;   // MODE 1: multiplicity of pair u (1 on the diagonal i == j, else 2), 512 floats
constexpr unsigned kPdLds = kPdTab, kPdLds1 = kPdTab + 512 * 4;

__device__ __forceinline__ d4 mfma_f64(double a, double b, d4 c) {
    return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
}
// LDS-DMA, 16 bytes per lane: LDS address = M0 + 16 * lane.  Invisible to the compiler's wait counting.
__device__ __forceinline__ void glds16(unsigned voff, const void *sbase, unsigned lds_addr) {
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1"
                 :
                 : "v"(voff), "s"(sbase), "s"(lds_addr)
                 : "memory");
}
// 16-byte store, scalar base + 32-bit lane offset (the compiler forms a 64-bit vector address for the same C++)
__device__ __forceinline__ void gstore16(unsigned voff, d2 v, const void *sbase) {
    asm volatile("global_store_dwordx4 %0, %1, %2" : : "v"(voff), "v"(v), "s"(sbase) : "memory");
}
// an address the compiler shall not take apart (it folds constants out of it and leaves `v_add_u32 v, 0, v` behind)
__device__ __forceinline__ unsigned opaque(unsigned x) {
    asm volatile("" : "+v"(x));
    return x;
}
// LDS accesses at integer byte addresses (through the `lds` array the compiler adds the array's base -- the literal 0 --
// with a vector instruction per access in some of the unrolled phases)
typedef __attribute__((address_space(3))) double lds_f64;
typedef __attribute__((address_space(3))) float lds_f32;
typedef float f2 __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(3))) f2 lds_f2;
__device__ __forceinline__ double lds_ld(unsigned addr) { return *(lds_f64 *)(uintptr_t)addr; }
__device__ __forceinline__ void lds_st(unsigned addr, double v) { *(lds_f64 *)(uintptr_t)addr = v; }
// k (0..3) wave-uniform: scalar compares and one wait
__device__ __forceinline__ void wait_vm_dyn(int k) {
    if (k <= 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    else if (k == 1) asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
    else if (k == 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
}

// (memory, LDS and scalar instructions at a higher priority than the MFMAs: a wave that issues MFMAs back to back
//  otherwise keeps the issue port from its SIMD mate's loads and stores, profiles/mfma_f64_coissue.txt)
#ifdef EVC_PTD_NO_SETPRIO
#define EVC_PTD_PRIO(x_) do { } while (0)
#else
#define EVC_PTD_PRIO(x_) __builtin_amdgcn_s_setprio(x_)
#endif

}  // namespace

// MODE 0: dense (pair, pair) result out[tri(r',s')][tri(p,q)], pitch out_ld.
// MODE 1: the 8-fold compressed packed vector packed[tri(u, v)], u = tri(r',s') >= v = tri(p,q), times the
//         multiplicities of both pairs (and diag_mult on u == v): the write-out walks the rows u = ur + 64 k from 0 with
//         lane-constant offsets tri(u) and skips the passes above the diagonal (k < v / 64, wave-uniform); the pass that
//         holds the diagonal takes a slower, predicated body, once per tile.  Its stores vary from tile to tile, so here
//         the wait for an operand row is vmcnt(0) (the stores behind it are a phase old by then).
template <int MODE>
__global__ __launch_bounds__(256, 2) void ptd_kernel(PairTransformArgs a) {
    constexpr int KS = 8, NT = 2;
    extern __shared__ __align__(16) char lds[];   // the only LDS of this kernel: it starts at LDS address 0
    const int n = a.n;
    const int npairs = n * (n + 1) / 2;
    const int in_ld = a.in_ld ? a.in_ld : npairs, out_ld = a.out_ld;   // (0: the caller's s4 matrix, pitch n(n+1)/2)
    const int64_t g = blockIdx.y;
    const double *__restrict__ in = a.in + g * a.sin;
    const double *__restrict__ C = a.C + g * a.sC;
    const int ntiles = (npairs + 7) / 8;
    const int t_begin = blockIdx.x * a.tiles_per_wg, t_end = min(ntiles, t_begin + a.tiles_per_wg);
    if (t_begin >= t_end) return;
    const int niter = 2 * (t_end - t_begin);
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int l15 = lane & 15, l4 = lane >> 4;

    // this wave's leading pair of iteration i: e0 + 4 i.  The DMA reads 16-byte aligned granules: the window of a row
    // starts dw doubles in front of it (dw = 1 when the row starts on an odd multiple of 8 bytes); rows 4 apart -- and
    // the row idle slots fall back to, row `wave` -- have the same dw whatever the pitch
    const int e0 = 8 * t_begin + wave;
    const int dw = (int)(((reinterpret_cast<uintptr_t>(in) >> 3) + (uintptr_t)((int64_t)e0 * in_ld)) & 1);
    const unsigned rowb = (unsigned)wave * kPdRB;
    unsigned fa[NT][KS];   // LDS address of fragment (rt, kk) of the symmetric matrix in the wave's row
#pragma unroll
    for (int rt = 0; rt < NT; ++rt)
#pragma unroll
        for (int kk = 0; kk < KS; ++kk) {
            const int r = rt * 16 + l15, s = 4 * kk + l4;
            const int hi = s > r ? s : r, lo = s > r ? r : s;
            fa[rt][kk] = (r < n && s < n) ? rowb + 8u * (unsigned)(dw + hi * (hi + 1) / 2 + lo) : rowb + kPdRaw * 1024;
        }
    if (lane < 8) *reinterpret_cast<double *>(lds + rowb + kPdRaw * 1024 + 8 * lane) = 0.0;
    const unsigned wbytes = (unsigned)(((npairs + dw) * 8 + 15) & ~15);   // bytes of the window
    unsigned vo[kPdRaw];   // lanes past the end re-read the first granule
#pragma unroll
    for (unsigned u = 0; u < kPdRaw; ++u) {
        const unsigned o = 1024u * u + 16u * (unsigned)lane;
        vo[u] = o < wbytes ? o : 0u;
    }
    const int npieces = __builtin_amdgcn_readfirstlane((int)((wbytes + 1023u) >> 10));
    auto dma_row = [&](int e) {
        const int ec = e < npairs ? e : wave;
        const char *src = reinterpret_cast<const char *>(in + (int64_t)ec * in_ld) - 8 * dw;
#pragma unroll
        for (unsigned u = 0; u < kPdRaw; ++u)
            if ((int)u < npieces) glds16(vo[u], src, rowb + 1024u * u);
    };
    dma_row(e0);

    // the X fragments straight from global memory (16 rows of 128 bytes per load instruction)
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

    // stage address of result register f = tile * 4 + reg (tiles (0,0), (1,0), (1,1)) for slot `wave` of buffer 0;
    // registers that hold no result go to the 16 dump doubles behind the column's rows
    unsigned sa[2][12];   // [buffer]: (an immediate offset beyond 32 KB makes the compiler emit `v_add_u32 v, 0, v` per write)
#pragma unroll
    for (int f = 0; f < 12; ++f) {
        const int tile = f / 4, reg = f % 4;
        const int it = tile == 0 ? 0 : 1, st = tile == 2 ? 1 : 0;
        const int r2 = it * 16 + l4 + 4 * reg, s2 = st * 16 + l15;
        const bool ok = r2 < n && s2 <= r2;
        sa[0][f] = opaque(kPdStage + (unsigned)wave * kPdP + 8u * (unsigned)(ok ? r2 * (r2 + 1) / 2 + s2 : npairs + l15));
        sa[1][f] = opaque(sa[0][f] + 8u * kPdP);
    }
    // write-out: thread (wl, ur) takes the columns wl, wl + 1 (wl even) of the rows ur + 64 k; a wave writes whole
    // groups of 16 rows (the result buffer has pair_ld(n) rows): pass k of this wave is valid <=> 64 k + 16 wave < npairs
    const int wl = 2 * (threadIdx.x & 3), ur = threadIdx.x >> 2;
    const unsigned wo = opaque(kPdStage + (unsigned)wl * kPdP + 8u * (unsigned)ur);
    const unsigned gvo = (unsigned)((ur * out_ld + wl) * 8);
    char *outg = reinterpret_cast<char *>(MODE == 0 ? a.out + g * a.sout : a.packed + g * a.spacked);
    [[maybe_unused]] unsigned tv[8];     // MODE 1: byte offset of (row ur + 64 k, column wl) in the packed vector
    [[maybe_unused]] double frow[8];     // MODE 1: multiplicity of row ur + 64 k
    if constexpr (MODE == 1) {
        // table of pair multiplicities (the diagonal pairs u = tri(i,i) = i (i + 3) / 2 are 1, the others 2)
        for (int u = threadIdx.x; u < 512; u += 256) {
            const int r = tri_row_small(u);
            *(lds_f32 *)(uintptr_t)(kPdTab + 4u * (unsigned)u) = (u == r * (r + 3) / 2) ? 1.0f : 2.0f;
        }
        if (blockIdx.x == 0) {   // zero the padding [M, packed_len) once per geometry
            double *pk = a.packed + g * a.spacked;
            const int64_t M = (int64_t)npairs * (npairs + 1) / 2;
            for (int64_t m = M + threadIdx.x; m < a.packed_len; m += 256) pk[m] = 0.0;
        }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int u = ur + 64 * k;
            tv[k] = (unsigned)((u * (u + 1) / 2 + wl) * 8);
            frow[k] = (double)*(lds_f32 *)(uintptr_t)(kPdTab + 4u * (unsigned)(u & 511));
        }
    }
    const int wrows = npairs - 16 * wave;   // pass k valid <=> 64 k < wrows
    // stores of an N phase that are younger than its DMA (passes c = 1..3 of the phase): what the wait for the next
    // operand row leaves in flight
    int vmask = 0;   // bit k: this wave writes pass k
#pragma unroll
    for (int k = 0; k < 8; ++k) vmask |= (64 * k < wrows) ? 1 << k : 0;
    vmask = __builtin_amdgcn_readfirstlane(vmask);
    int cntA = 0, cntB = 0;
#pragma unroll
    for (int c = 1; c < 4; ++c) {
        cntA += (64 * c < wrows) ? 1 : 0;
        cntB += (64 * (4 + c) < wrows) ? 1 : 0;
    }

    double mf[NT][KS];
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
    for (int rt = 0; rt < NT; ++rt)
#pragma unroll
        for (int kk = 0; kk < KS; ++kk) mf[rt][kk] = lds_ld(fa[rt][kk]);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    dma_row(e0 + 4);

    d4 nnA[NT][NT], nnB[NT][NT];
#pragma unroll
    for (int it = 0; it < NT; ++it)
#pragma unroll
        for (int st = 0; st < NT; ++st) nnA[it][st] = nnB[it][st] = (d4){0.0, 0.0, 0.0, 0.0};
    const d4 zero = {0.0, 0.0, 0.0, 0.0};

    // One iteration = one matrix (phase PH = i & 3: matrix parity ODD, tile parity TP).  COMPUTE: the main loop;
    // without: the two drain-only iterations behind it.
    auto iteration = [&](auto compute_tag, auto ph_tag, const int i, d4 (&nnp)[NT][NT], d4 (&nn)[NT][NT]) {
        constexpr bool COMPUTE = decltype(compute_tag)::value;
        constexpr int PH = decltype(ph_tag)::value;
        constexpr bool ODD = (PH & 1) != 0;
        constexpr unsigned TP = PH >> 1;
        // ---------------------------------------------------------------- H = M X  (+ stage writes of matrix i-1)
        d4 h[NT][NT];
        if constexpr (COMPUTE || !ODD) {
            // matrix i-1: second half of its tile (slot wave + 4) when i is even; its tile's buffer
            constexpr unsigned simm = ODD ? 0u : 4u * kPdP, sbuf = ODD ? TP : (TP ^ 1u);
#pragma unroll
            for (int kk = 0; kk < KS; ++kk) {
#pragma unroll
                for (int m = 0; m < NT * NT; ++m) {
                    if constexpr (COMPUTE) {
                        const int rt = m / NT, st = m % NT;
                        EVC_PTD_PRIO(0);
                        h[rt][st] = mfma_f64(mf[rt][kk], xf[kk][st], kk == 0 ? zero : h[rt][st]);
                        EVC_PTD_PRIO(1);
                    }
                    const int f = kk * 2 + m;
                    if (m < 2 && f < 12) {
                        const int tile = f / 4, reg = f % 4;
                        const int it = tile == 0 ? 0 : 1, st2 = tile == 2 ? 1 : 0;
                        lds_st(sa[sbuf][f] + simm, nnp[it][st2][reg]);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        }
        if (!ODD && i >= 2) lds_barrier();
        // ---------------------------------------------------------------- N = X^T H  (+ everything else)
        {
            // tile jt (relative to t_begin) is complete after this barrier: written out in the N phases of iterations
            // 2 jt + 2 (passes 0..3) and 2 jt + 3 (passes 4..7); its buffer has the other parity than this matrix's tile
            int act = (ODD ? i >= 3 : i >= 2) ? vmask : 0;   // passes this wave writes in this phase (bit k)
            const int jt = (ODD ? i - 3 : i - 2) >> 1;
            constexpr unsigned bimm = (TP ^ 1u) * 8u * kPdP;
            char *ob = outg + (int64_t)(8 * (t_begin + jt)) * 8;
#ifdef EVC_PTD_K0
            const int K = 0;
#else
            const int K = (MODE == 0 && i >= 3) ? (ODD ? cntA : cntB) : 0;
#endif
            d2 dv[4];
            [[maybe_unused]] int kd = 0;       // MODE 1: the pass that holds the diagonal u == v of this tile's columns
            [[maybe_unused]] d2 cf = {1.0, 1.0};   // MODE 1: multiplicities of the thread's two columns
            if constexpr (MODE == 1) {
                const int tg = t_begin + (jt < 0 ? 0 : jt);
                kd = tg >> 3;
                act &= ~((1 << kd) - 1);
                ob = outg + (int64_t)(8 * tg) * 8;
                const f2 c2 = *(lds_f2 *)(uintptr_t)(kPdTab + 4u * (unsigned)(8 * tg + wl));
                cf = (d2){(double)c2[0], (double)c2[1]};
            }
            asm volatile("" : "+s"(act));   // (kept a scalar: bit tests at the uses, no lane-mask copies)
#pragma unroll
            for (int kk = 0; kk < KS; ++kk) {
                if ((kk & 1) == 0) {
                    const int c = kk / 2, k = (ODD ? 4 : 0) + c;
                    if (act >> k & 1) {
                        dv[c][0] = lds_ld(wo + 512u * k + bimm);
                        dv[c][1] = lds_ld(wo + 512u * k + bimm + kPdP);
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int m = 0; m < 3; ++m) {
                    if constexpr (COMPUTE) {
                        const int it = m == 0 ? 0 : 1, st = m == 2 ? 1 : 0;
                        EVC_PTD_PRIO(0);
                        nn[it][st] = mfma_f64(xf[kk][it], h[kk / 4][st][kk % 4], kk == 0 ? zero : nn[it][st]);
                        EVC_PTD_PRIO(1);
                        if (m == 0) {
                            // the operand row of matrix i+1 (requested one iteration ago) has landed -> fragments;
                            // then the row of matrix i+2 is requested into the same LDS row
                            if (kk == 0) wait_vm_dyn(K);
                            if (kk < 2) {
#pragma unroll
                                for (int k2 = 0; k2 < KS; ++k2)
                                    mf[kk][k2] = lds_ld(fa[kk][k2]);
                            }
                            if (kk == 2) {
                                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                                dma_row(e0 + 4 * (i + 2));
                            }
                        }
                    }
                    if (m == 2 && (kk & 1) == 1) {
                        const int c = kk / 2, k = (ODD ? 4 : 0) + c;
                        if constexpr (MODE == 0) {
                            if (act >> k & 1) gstore16(gvo, dv[c], ob + (int64_t)(64 * k) * out_ld * 8);
                        } else if (act >> k & 1) {
                            const int u = ur + 64 * k;
                            d2 f = cf * frow[k];
                            if (k != kd) {          // below the diagonal: both columns, rows beyond the last one masked
                                const d2 v = dv[c] * f;
                                if (u < npairs) gstore16(tv[k], v, ob);
                            } else {                // the diagonal pass: column v of row u only for u >= v, diag_mult on u == v
                                const int d = u - (8 * (t_begin + jt) + wl);
                                if (d == 0) f[0] *= a.diag_mult;
                                if (d == 1) f[1] *= a.diag_mult;
                                const d2 v = dv[c] * f;
                                if (u < npairs) {
                                    if (d >= 1) *reinterpret_cast<d2 *>(ob + tv[k]) = v;
                                    else if (d == 0) *reinterpret_cast<double *>(ob + tv[k]) = v[0];
                                }
                            }
                        }
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        }
    };
    using T = std::true_type;
    using F = std::false_type;
    using P0 = std::integral_constant<int, 0>;
    using P1 = std::integral_constant<int, 1>;
    using P2 = std::integral_constant<int, 2>;
    using P3 = std::integral_constant<int, 3>;
    int i = 0;
    for (; i + 4 <= niter; i += 4) {
        iteration(T{}, P0{}, i, nnB, nnA);
        iteration(T{}, P1{}, i + 1, nnA, nnB);
        iteration(T{}, P2{}, i + 2, nnB, nnA);
        iteration(T{}, P3{}, i + 3, nnA, nnB);
    }
    if (i < niter) {   // (niter = 2 mod 4)
        iteration(T{}, P0{}, i, nnB, nnA);
        iteration(T{}, P1{}, i + 1, nnA, nnB);
        iteration(F{}, P2{}, i + 2, nnB, nnA);
        iteration(F{}, P3{}, i + 3, nnA, nnB);
    } else {
        iteration(F{}, P0{}, i, nnB, nnA);
        iteration(F{}, P1{}, i + 1, nnA, nnB);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // (the rows requested beyond the last matrix)
}

// ---------------------------------------------------------------------------------- Y2 with LDS-DMA operand rows
// y2_fused_kernel (y2.hip) with the machinery above: per pair v one wave computes H^T = X^T M1_v (32 MFMAs) and
// Y += mult(v) T_v H^T (32 MFMAs), M1_v = row v of the first pair step's intermediate, T_v = row v of SB, both dense
// (pair, pair) forms at the pitch pair_ld(n).  The two rows come by LDS-DMA into two wave-private LDS rows; the T
// fragments are read during the H phase and the next T row requested, the next M fragments during the Y phase and
// the M row after that requested -- each DMA has a whole iteration to land and the wait in front of either read is the
// constant "one row younger" vmcnt.  No barrier, no store, no vector instruction in the loop: the multiplicity of the
// pair (1 on the diagonal i == j, else 2) selects one of two accumulator sets (wave-uniform branch), Y = Yd + 2 Yo at
// the end.
__global__ __launch_bounds__(256, 2) void y2d_kernel(const double *__restrict__ SB, const double *__restrict__ M1,
                                                     const double *__restrict__ X, int64_t sX, int n,
                                                     double *__restrict__ partial, int64_t sws, int tiles_per_wg, int ppt) {
    constexpr int KS = 8, NT = 2, NPAD = 32;
    extern __shared__ __align__(16) char lds[];
    const int npairs = n * (n + 1) / 2, ld = pair_ld(n);
    const int64_t g = blockIdx.y;
    SB += g * sws;
    M1 += g * sws;
    X += g * sX;
    partial += g * sws;
    const int ntiles = (npairs + ppt - 1) / ppt;
    const int t_begin = blockIdx.x * tiles_per_wg, t_end = min(ntiles, t_begin + tiles_per_wg);
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int l15 = lane & 15, l4 = lane >> 4;
    d4 yD[NT][NT], yO[NT][NT];
#pragma unroll
    for (int ti = 0; ti < NT; ++ti)
#pragma unroll
        for (int ta = 0; ta < NT; ++ta) yD[ti][ta] = yO[ti][ta] = (d4){0.0, 0.0, 0.0, 0.0};
    if (t_begin < t_end) {
        const int niter = (ppt / 4) * (t_end - t_begin);
        const unsigned rowM = (unsigned)wave * 2u * kPdRB, rowT = rowM + kPdRB;
        unsigned fa[NT][KS];   // LDS address of fragment (rt, kk) in the M row (T row: + kPdRB)
#pragma unroll
        for (int rt = 0; rt < NT; ++rt)
#pragma unroll
            for (int kk = 0; kk < KS; ++kk) {
                const int r = rt * 16 + l15, s = 4 * kk + l4;
                const int hi = s > r ? s : r, lo = s > r ? r : s;
                fa[rt][kk] = opaque((r < n && s < n) ? rowM + 8u * (unsigned)(hi * (hi + 1) / 2 + lo) : rowM + kPdRaw * 1024);
            }
        if (lane < 8) {
            *reinterpret_cast<double *>(lds + rowM + kPdRaw * 1024 + 8 * lane) = 0.0;
            *reinterpret_cast<double *>(lds + rowT + kPdRaw * 1024 + 8 * lane) = 0.0;
        }
        const unsigned wbytes = (unsigned)((npairs * 8 + 15) & ~15);   // (rows start on 128-byte lines)
        unsigned vo[kPdRaw];
#pragma unroll
        for (unsigned u = 0; u < kPdRaw; ++u) {
            const unsigned o = 1024u * u + 16u * (unsigned)lane;
            vo[u] = o < wbytes ? o : 0u;
        }
        const int npieces = __builtin_amdgcn_readfirstlane((int)((wbytes + 1023u) >> 10));
        auto dma_row = [&](const double *base, int e, unsigned dst) {
            const char *src = reinterpret_cast<const char *>(base + (int64_t)(e < npairs ? e : wave) * ld);
#pragma unroll
            for (unsigned u = 0; u < kPdRaw; ++u)
                if ((int)u < npieces) glds16(vo[u], src, dst + 1024u * u);
        };
        // "everything but the youngest row has landed": npieces DMA instructions may stay in flight
        auto wait_older = [&]() {
            if (npieces >= 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
            else if (npieces == 3) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
            else if (npieces == 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
        };
        const int e0 = ppt * t_begin + wave;   // this wave's pair of iteration i: e0 + 4 i
        dma_row(M1, e0, rowM);
        dma_row(SB, e0, rowT);
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
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
        for (int rt = 0; rt < NT; ++rt)
#pragma unroll
            for (int kk = 0; kk < KS; ++kk) mf[rt][kk] = lds_ld(fa[rt][kk]);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        dma_row(M1, e0 + 4, rowM);
        // (p, q) of the pair e, advanced by 4 per iteration with scalar arithmetic: on the diagonal <=> q == p
        int pp = __builtin_amdgcn_readfirstlane(tri_row_small(e0 < npairs ? e0 : 0)), qq = (e0 < npairs ? e0 : 0) - pp * (pp + 1) / 2;
        const d4 zero = {0.0, 0.0, 0.0, 0.0};
        for (int i = 0; i < niter; ++i) {
            const int e = e0 + 4 * i;
            const bool live = e < npairs, diag = qq == pp;
            // ------------------------------------------------------------ H^T = X^T M_e  (+ T fragments of pair e)
            d4 hT[NT][NT];
#pragma unroll
            for (int kk = 0; kk < KS; ++kk) {
#pragma unroll
                for (int m = 0; m < NT * NT; ++m) {
                    const int it = m / NT, st = m % NT;
                    hT[it][st] = mfma_f64(xf[kk][it], mf[st][kk], kk == 0 ? zero : hT[it][st]);
                    if (m == 0) {
                        if (kk == 0) wait_older();   // the T row of this pair (the M row of the next one may be in flight)
                        if (kk < 2) {
#pragma unroll
                            for (int k2 = 0; k2 < KS; ++k2) tf[kk][k2] = lds_ld(fa[kk][k2] + kPdRB);
                        }
                        if (kk == 2) {
                            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                            dma_row(SB, e + 4, rowT);
                        }
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            // ------------------------------------------------------------ Y += mult(e) T_e H^T  (+ M fragments of pair e + 4)
            // (one wave-uniform branch per phase selects the accumulator set; idle slots of the last tile move data only)
            auto yphase = [&](auto live_tag, d4 (&y)[NT][NT]) {
#pragma unroll
                for (int kk = 0; kk < KS; ++kk) {
#pragma unroll
                    for (int m = 0; m < NT * NT; ++m) {
                        const int ti = m / NT, ta = m % NT;
                        if constexpr (decltype(live_tag)::value)
                            y[ti][ta] = mfma_f64(tf[ti][kk], hT[kk / 4][ta][kk % 4], y[ti][ta]);
                        if (m == 0) {
                            if (kk == 0) wait_older();   // the M row of the next pair
                            if (kk < 2) {
#pragma unroll
                                for (int k2 = 0; k2 < KS; ++k2) mf[kk][k2] = lds_ld(fa[kk][k2]);
                            }
                            if (kk == 2) {
                                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                                dma_row(M1, e + 8, rowM);
                            }
                        }
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
            };
            if (!live) yphase(std::false_type{}, yD);
            else if (diag) yphase(std::true_type{}, yD);
            else yphase(std::true_type{}, yO);
            qq += 4;
            while (qq > pp) {
                qq -= pp + 1;
                ++pp;
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    // cross-wave sum (every workgroup writes its slab, workgroups without tiles a zero one)
    __syncthreads();
    double *red = reinterpret_cast<double *>(lds);   // [4][NPAD][NPAD + 1], over the rows once they are done with
#pragma unroll
    for (int ti = 0; ti < NT; ++ti)
#pragma unroll
        for (int ta = 0; ta < NT; ++ta)
#pragma unroll
            for (int r = 0; r < 4; ++r)
                red[(wave * NPAD + ti * 16 + l4 + 4 * r) * (NPAD + 1) + ta * 16 + l15] = yD[ti][ta][r] + 2.0 * yO[ti][ta][r];
    __syncthreads();
    double *dst = partial + (int64_t)blockIdx.x * n * n;
    for (int idx = threadIdx.x; idx < n * n; idx += 256) {
        const int i = idx / n, aa = idx % n;
        const int o = i * (NPAD + 1) + aa;
        constexpr int WS = NPAD * (NPAD + 1);
        dst[idx] = (red[o] + red[WS + o]) + (red[2 * WS + o] + red[3 * WS + o]);
    }
}

bool y2_dma_applicable(int n) {
    static const bool on = !(getenv("EVC_PT_DMA") && atoi(getenv("EVC_PT_DMA")) == 0);
    return on && n > 16 && n <= kPdMaxN;
}

int launch_y2_dma(const double *SB, const double *M1, const double *X, int64_t sX, int n, double *partial, int64_t sws,
                  int count, int slabs, int tiles_per_wg, int ppt, hipStream_t st) {
    constexpr unsigned ldsb = 8 * kPdRB > 4 * 32 * 33 * 8 ? 8 * kPdRB : 4 * 32 * 33 * 8;
    hipLaunchKernelGGL(y2d_kernel, dim3((unsigned)slabs, (unsigned)count), dim3(256), ldsb, st, SB, M1, X, sX, n, partial,
                       sws, tiles_per_wg, ppt);
    note_kernel(EVC_PROF_Y2, "y2d_kernel");
    EVC_LAUNCH_CHECK("y2_dma");
    return 0;
}

bool pair_transform_dma_applicable(const PairTransformArgs &a, int count) {
    static const bool on = !(getenv("EVC_PT_DMA") && atoi(getenv("EVC_PT_DMA")) == 0);
    const int npairs = a.n * (a.n + 1) / 2;
    return on && a.n > 16 && a.n <= kPdMaxN && count >= 4 && a.lead_sym && a.in_lower && a.rs_lower && a.in_pairs &&
           !a.k3 && (a.in_ld == 0 || a.in_ld >= npairs) &&
           ((a.out && a.out_pairs && !a.packed && a.out_ld >= 8 * ((npairs + 7) / 8) && a.out_ld % 2 == 0) ||
            (a.packed && a.sym8 && !a.out));
}

int launch_pair_transform_dma(const PairTransformArgs &a_in, int count, hipStream_t st) {
    PairTransformArgs a = a_in;
    const int npairs = a.n * (a.n + 1) / 2, ntiles = (npairs + 7) / 8;
    if (a.tiles_per_wg < 1) a.tiles_per_wg = 4;
    const dim3 grid((unsigned)((ntiles + a.tiles_per_wg - 1) / a.tiles_per_wg), (unsigned)count);
    if (a.packed) {
        static LdsAttr attr1;
        if (int rc = allow_dynamic_lds(ptd_kernel<1>, attr1, 160 * 1024, "pair_transform_dma")) return rc;
        hipLaunchKernelGGL((ptd_kernel<1>), grid, dim3(256), kPdLds1, st, a);
        note_kernel(EVC_PROF_PAIR_TRANSFORM, "ptd_kernel<1>");
    } else {
        static LdsAttr attr;
        if (int rc = allow_dynamic_lds(ptd_kernel<0>, attr, 160 * 1024, "pair_transform_dma")) return rc;
        hipLaunchKernelGGL((ptd_kernel<0>), grid, dim3(256), kPdLds, st, a);
        note_kernel(EVC_PROF_PAIR_TRANSFORM, "ptd_kernel<0>");
    }
    EVC_LAUNCH_CHECK("pair_transform_dma");
    return 0;
}

}  // namespace evc
