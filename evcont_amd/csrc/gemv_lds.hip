// The batched streaming contractions with the t-RDM streamed through LDS by LDS-DMA (global_load_lds_dwordx4).
// K5 (this part of the file; batches of 12 .. 32 geometries per pass), K8 further down (17 .. 32):
//   Y[row][g] = sum_c A[row][c] v[g][c]      (reference: ab_initio_eigenvector_continuation.py:57-64, the
//   np.einsum / np.sum contractions of the stored transition RDMs with the rotated integrals)
//
// Why a second rows kernel (round 3, tools/micro/rows_pattern.hip with the caches flushed by a READ of 2 GB before
// every launch): the fragment-shaped loads of gemv_rows_mfma_pipe_kernel (a load instruction = 16 rows x 64 bytes)
// read the 210 x 108 345 matrix COLD at 4.0 TB/s, whole-line pieces at 5.0 TB/s -- the earlier comparison (5.5 against
// 5.9 TB/s) had the 182 MB matrix sitting in the 256 MB memory-side cache between the repeats, which the kernel in
// situ never sees (a batch moves 1.4 GB between two passes over the matrix).  Here every load instruction moves
// 8 rows x 128 bytes (whole cache lines) straight into LDS, no VGPRs in between, so one wave per SIMD keeps a whole
// image (30-34 instructions = 30-34 KB) in flight, and the MFMA fragments come out of LDS with ds_read_b128.
//
// Decomposition: block = (column span, row group of NT 16-row tiles), one block per CU, exactly as many blocks as
// units of work (host-built block table).  The four waves interleave the 16-column chunks of the span (128 bytes per
// row: one cache line) and are completely independent until the epilogue: no barrier in the stream.  A wave's LDS
// image holds NCH chunks, each GS tiles of geometry vectors (16 geometries x 16 columns) followed by NT tiles of
// matrix rows, 2 KB per tile.  Slot p of the image is refilled for the NEXT image one position after it was read, so
// the image is a sliding window and (NSI - 1) tiles = 2 (NSI - 1) instructions are in flight all the time; LDS-DMA
// completes in order, so the wait in front of the reads of slot p is the constant vmcnt(2 (NSI - 2)).
// Shapes (NT, NCH) = (14,1) (7,2) (4,3) (2,4): see lds_pick_nt below for what the row groups buy.
//
// LDS image of a tile (rule "linear destination, swizzled SOURCE, same swizzle on the read"): instruction j of a
// tile writes 1 KB = rows 8j..8j+7 x 128 bytes, lane i -> row 8j + (i >> 3), 16-byte position i & 7; the position
// holds piece q = pos ^ ((row >> 1) & 7) of the row.  A fragment read (ds_read_b128: lane (l15, l4) takes piece
// 4u + l4 of row l15) is then conflict free in every 16-lane group of the instruction (SQ_LDS_BANK_CONFLICT = 0,
// profiles/r03_pmc_lds_sym8_batch32.csv).
#include <stdlib.h>

#include "common.hpp"
#include "kernels.hpp"

namespace evc {

typedef double d4 __attribute__((ext_vector_type(4)));
typedef double d2v __attribute__((ext_vector_type(2)));

namespace {

constexpr int kLW = 16;           // columns per wave chunk
constexpr int kTileBytes = 2048;  // 16 rows x 16 columns

__device__ __forceinline__ d4 mfma64(double a, double b, d4 c) {
    return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
}

// LDS-DMA, 16 bytes per lane: LDS address = M0 + 16 * lane.  The compiler neither counts these loads nor knows that
// they write LDS: every wait below is explicit, and the "memory" clobbers keep its own LDS reads on their side.
__device__ __forceinline__ void glds_s(unsigned voff, const void *sbase, unsigned lds_addr) {
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1"
                 :
                 : "v"(voff), "s"(sbase), "s"(lds_addr)
                 : "memory");
}
__device__ __forceinline__ void glds_v(const void *vaddr, unsigned lds_addr) {
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" : : "v"(vaddr), "s"(lds_addr) : "memory");
}
// wave-uniform values the compiler's divergence analysis may not recognise as such (an "s" constraint alone does not
// move a VGPR value into an SGPR)
__device__ __forceinline__ unsigned uniform_u32(unsigned x) { return (unsigned)__builtin_amdgcn_readfirstlane((int)x); }
__device__ __forceinline__ const void *uniform_ptr(const void *p) {
    const unsigned long long b = reinterpret_cast<unsigned long long>(p);
    const unsigned lo = uniform_u32((unsigned)b), hi = uniform_u32((unsigned)(b >> 32));
    return reinterpret_cast<const void *>(((unsigned long long)hi << 32) | lo);
}
template <int N>
__device__ __forceinline__ void wait_vm() {
    asm volatile("s_waitcnt vmcnt(%0)" : : "n"(N) : "memory");
}
// (n is a constant after unrolling: the switch folds to one instruction)
__device__ __forceinline__ void wait_vm_n(int n) {
#define EVC_VM_CASE(k) case k: wait_vm<k>(); break;
    switch (n) {
        EVC_VM_CASE(0) EVC_VM_CASE(2) EVC_VM_CASE(4) EVC_VM_CASE(6) EVC_VM_CASE(8) EVC_VM_CASE(10) EVC_VM_CASE(12)
        EVC_VM_CASE(14) EVC_VM_CASE(16) EVC_VM_CASE(18) EVC_VM_CASE(20) EVC_VM_CASE(22) EVC_VM_CASE(24)
        EVC_VM_CASE(26) EVC_VM_CASE(28) EVC_VM_CASE(30) EVC_VM_CASE(32) EVC_VM_CASE(34) EVC_VM_CASE(36)
        EVC_VM_CASE(38) EVC_VM_CASE(40) EVC_VM_CASE(42) EVC_VM_CASE(44) EVC_VM_CASE(46) EVC_VM_CASE(48)
        EVC_VM_CASE(50) EVC_VM_CASE(52) EVC_VM_CASE(54) EVC_VM_CASE(56) EVC_VM_CASE(58) EVC_VM_CASE(60)
        default: wait_vm<0>(); break;
    }
#undef EVC_VM_CASE
}
__device__ __forceinline__ void wait_lds() { asm volatile("s_waitcnt lgkmcnt(0)" : : : "memory"); }

__device__ __forceinline__ double2 ld2g(const double *row, int64_t c, int64_t cols) {
    if (c + 1 < cols) return *reinterpret_cast<const double2 *>(row + c);
    return make_double2(c < cols ? row[c] : 0.0, 0.0);
}

}  // namespace

// Timing experiments (tools/micro/k5_stamps.py; build with EVC_DEBUG_STAMPS=1), compiled out of the product library.
#ifdef EVC_DEBUG_STAMPS
__device__ long long g_k5l_wg[1024 * 8];   // per workgroup: entry, main loop done (wave 0), exit, hw id, 4 epilogue stamps
#define EVC_K5L_WG(i_)                                                                 \
    do {                                                                               \
        if (threadIdx.x == 0 && blockIdx.x < 1024) {                                   \
            long long t_;                                                              \
            asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory"); \
            g_k5l_wg[8 * blockIdx.x + (i_)] = t_;                                      \
            if ((i_) == 0) {                                                           \
                unsigned hw_, xcc_;                                                    \
                asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw_));      \
                asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc_));    \
                g_k5l_wg[8 * blockIdx.x + 3] = ((long long)(xcc_ & 0xF) << 32) | hw_;  \
            }                                                                          \
        }                                                                              \
    } while (0)
#else
#define EVC_K5L_WG(i_) do { } while (0)
#endif

// block -> which << 15 | row group << 8 | span (lds_block_map below)
struct LdsBlockMap {
    unsigned short e[256];
};

// One wave's state: everything the unrolled body needs.  NCH chunks (consecutive chunks of this wave, 64 columns
// apart) make one image of NSI = NCH (NT + GS) slots.
template <int GS, int NT, int NCH>
struct LdsRowsWave {
    static constexpr int NS = NT + GS;
    static constexpr int NSI = NS * NCH;
    const char *abase;       // first row of the row group, column 0 (wave-uniform)
    int64_t tile_stride;     // 16 rows in bytes
    int rows_left;           // rows of the matrix from the row group's first row on (>= 1)
    int64_t ld;
    int64_t clast;           // first column of this wave's last complete chunk (chunks beyond it re-read it)
    const char *vb[GS][2];   // per lane: its 16 bytes of its geometry's vector at column 0, both instructions of a tile
    unsigned voff[2];        // per lane: its 16 bytes relative to the tile's first row, column 0 (full tiles)
    unsigned lbase;          // LDS byte address of this wave's image
    int prow, ppiece;        // producer: row within the 8-row piece (lane >> 3) and 16-byte position (lane & 7)
    unsigned raddr[2];       // consumer: byte offset of (row l15, k pair u) within a tile

    // refill slot S of the image whose first chunk starts at column c0
    template <int S>
    __device__ __forceinline__ void issue(int64_t c0) const {
        constexpr int j = S / NS, sl = S % NS;
        const unsigned la = lbase + S * kTileBytes;
        int64_t cc = c0 + j * (4 * kLW);
        if (NCH > 1 && cc > clast) cc = clast;   // (a chunk the wave does not have: valid bytes, never consumed)
        if constexpr (sl < GS) {
            glds_v(vb[sl][0] + cc * 8, la);
            glds_v(vb[sl][1] + cc * 8, la + 1024);
        } else {
            constexpr int t = sl - GS;
            const int left = rows_left - 16 * t;   // rows of the matrix in and below this tile (wave-uniform)
            if (left >= 16) {
                const char *sb = abase + t * tile_stride + cc * 8;
                glds_s(voff[0], sb, la);
                glds_s(voff[1], sb, la + 1024);
            } else {
                // ragged or empty tile: rows beyond the matrix re-read its last row (their results are discarded)
                const int first = left >= 1 ? 16 * t : rows_left - 1;   // first row fetched, relative to the group
                const int last = left >= 1 ? left - 1 : 0;              // last valid row relative to `first`
                const char *sb = abase + (int64_t)first * ld * 8 + cc * 8;
#pragma unroll
                for (int jj = 0; jj < 2; ++jj) {
                    const int r = 8 * jj + prow;
                    const int q = ppiece ^ ((r >> 1) & 7);
                    const int rc = r < last ? r : last;
                    glds_s((unsigned)(((int64_t)rc * ld + 2 * q) * 8), sb, la + 1024 * jj);
                }
            }
        }
    }
};

// slot index known after unrolling: the chain folds to the one call
template <int GS, int NT, int NCH, int S = 0>
__device__ __forceinline__ void lds_issue_dyn(const LdsRowsWave<GS, NT, NCH> &w, int s, int64_t c0) {
    if constexpr (S < (NT + GS) * NCH) {
        if (s == S) w.template issue<S>(c0);
        else lds_issue_dyn<GS, NT, NCH, S + 1>(w, s, c0);
    }
}
// slots 0 .. NSI-2 (the prologue: slot NSI-1 is requested by position 0 of the image itself)
template <int GS, int NT, int NCH, int S = 0>
__device__ __forceinline__ void lds_issue_range(const LdsRowsWave<GS, NT, NCH> &w, int64_t c0) {
    if constexpr (S < (NT + GS) * NCH - 1) {
        w.template issue<S>(c0);
        lds_issue_range<GS, NT, NCH, S + 1>(w, c0);
    }
}

// GS sets of 16 geometries [g0, g0 + G), 16 (GS - 1) < G <= 16 GS; NT = tiles per row group (every row group runs NT
// tiles: tiles beyond the group's own are fetched from valid rows and discarded); NCH = chunks per image.
template <int GS, int NT, int NCH>
__global__ __launch_bounds__(256, 1) void gemv_rows_lds_kernel(GemvRowsLaunch L, LdsBlockMap M, int g0, int G) {
    extern __shared__ __align__(16) double lds_img[];
    using W = LdsRowsWave<GS, NT, NCH>;
    constexpr int NS = W::NS, NSI = W::NSI;
    constexpr int IMG = NSI * kTileBytes;   // bytes per wave
    // block -> (problem, span, row group) by the host's table: exactly one block per unit of work -- one workgroup
    // fits a CU, a launch of more blocks than CUs runs its last blocks in a second round (measured: +20 us)
    const unsigned e = M.e[blockIdx.x];
    const int which = e >> 15, rg = (e >> 8) & 127, span = e & 255;
    EVC_K5L_WG(0);
    const RowProblem &P = L.p[which];
    const int64_t rows = P.rows, cols = P.cols, ld = P.ld;
    const int tpg = L.tpg[which], trem = L.trem[which];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l15 = lane & 15, l4 = lane >> 4;
    const int64_t row_base = (int64_t)(rg * tpg + min(rg, trem)) * 16;
    const int ntile = tpg + (rg < trem ? 1 : 0);  // tiles of this row group (the last one may be ragged)
    const int64_t cbeg = (int64_t)span * P.span_cols;
    const int64_t cend = min(cols, (int64_t)(span + 1) * P.span_cols);
    const int64_t cfull = min(cend, cols & ~(int64_t)(kLW - 1));   // chunks starting below this are complete
    int64_t c = cbeg + wave * kLW;
    const int nmy = c < cfull ? (int)((cfull - c + 4 * kLW - 1) / (4 * kLW)) : 0;   // complete chunks of this wave

    W w;
    w.abase = reinterpret_cast<const char *>(P.A + row_base * ld);
    w.tile_stride = 16 * ld * 8;
    w.rows_left = (int)(rows - row_base);
    w.ld = ld;
    w.clast = c + (int64_t)(nmy > 0 ? nmy - 1 : 0) * (4 * kLW);
    w.prow = lane >> 3;
    w.ppiece = lane & 7;
    w.lbase = (unsigned)(wave * IMG);   // the dynamic array is the only LDS of this kernel: it starts at address 0
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int r = 8 * j + w.prow;
        const int q = w.ppiece ^ ((r >> 1) & 7);
        w.voff[j] = (unsigned)(((int64_t)r * ld + 2 * q) * 8);
#pragma unroll
        for (int gs = 0; gs < GS; ++gs) {
            const int slot = 16 * gs + r;
            const int gg = g0 + (slot < G ? slot : 0);   // slots beyond G read geometry g0 (a valid address)
            w.vb[gs][j] = reinterpret_cast<const char *>(P.v + (int64_t)gg * P.vstride + 2 * q);
        }
    }
#pragma unroll
    for (int u = 0; u < 2; ++u)
        w.raddr[u] = (unsigned)((l15 >> 3) * 1024 + (l15 & 7) * 128 + (((4 * u + l4) ^ ((l15 >> 1) & 7)) * 16));

    d4 acc[GS][NT];
#pragma unroll
    for (int gs = 0; gs < GS; ++gs)
#pragma unroll
        for (int t = 0; t < NT; ++t) acc[gs][t] = (d4){0.0, 0.0, 0.0, 0.0};

    const char *img = reinterpret_cast<const char *>(lds_img) + wave * IMG;
    double2 bf[GS][2], bfn[GS][2];   // geometry-vector fragments of the chunk in work / of the chunk being opened
    double2 af[2][2];                // matrix fragments, alternating tiles
#define EVC_LDS_MMA(T_, AF_)                                                                    \
    {                                                                                           \
        _Pragma("unroll") for (int u = 0; u < 2; ++u) {                                         \
            _Pragma("unroll") for (int gs = 0; gs < GS; ++gs)                                   \
                acc[gs][T_] = mfma64(AF_[u].x, bf[gs][u].x, acc[gs][T_]);                       \
            _Pragma("unroll") for (int gs = 0; gs < GS; ++gs)                                   \
                acc[gs][T_] = mfma64(AF_[u].y, bf[gs][u].y, acc[gs][T_]);                       \
        }                                                                                       \
    }
// One image: chunks at C0_, C0_ + 64, ... of which NV_ are the wave's own; NEXT_: another image follows at CN_ (its slots
// are refilled one position behind the reads).  PEND_: the last matrix tile of the chunk before still waits for its MFMAs
// (they run behind the first fragment read of the next chunk, so no LDS latency is exposed between chunks).
// Position p = (chunk j of the image, slot sl): wait for slot p, read its fragments, run the MFMAs of the tile before,
// refill slot p - 1 (position 0: the image's own last slot).  LDS-DMA completes in order and NSI - 2 tiles are
// younger than slot p whenever it is waited for: the constant vmcnt(2 (NSI - 2)) in the stream, counted down in the
// last image.
#define EVC_LDS_IMAGE(NEXT_, C0_, CN_, NV_)                                                                 \
    {                                                                                                       \
        _Pragma("unroll") for (int p = 0; p < NSI; ++p) {                                                   \
            const int j = p / NS, sl = p - j * NS;                                                          \
            wait_vm_n((NEXT_) || p == 0 ? 2 * (NSI - 2) : (p < NSI - 1 ? 2 * (NSI - 1 - p) : 0));            \
            if (j < (NV_)) {                                                                                \
                const char *tp = img + p * kTileBytes;                                                      \
                if (sl < GS) {                                                                              \
                    _Pragma("unroll") for (int u = 0; u < 2; ++u)                                           \
                        bfn[sl < GS ? sl : 0][u] = *reinterpret_cast<const double2 *>(tp + w.raddr[u]);     \
                    if (sl == 0 && pend) EVC_LDS_MMA(NT - 1, af[(NS - 1) & 1])                              \
                } else {                                                                                    \
                    if (sl == GS) {                                                                         \
                        _Pragma("unroll") for (int gs = 0; gs < GS; ++gs)                                   \
                            _Pragma("unroll") for (int u = 0; u < 2; ++u) bf[gs][u] = bfn[gs][u];           \
                    }                                                                                       \
                    _Pragma("unroll") for (int u = 0; u < 2; ++u)                                           \
                        af[sl & 1][u] = *reinterpret_cast<const double2 *>(tp + w.raddr[u]);                \
                    if (sl >= GS + 1) EVC_LDS_MMA(sl - GS - 1 >= 0 ? sl - GS - 1 : 0, af[(sl - 1) & 1])     \
                    if (sl == NS - 1) pend = true;                                                          \
                }                                                                                           \
            }                                                                                               \
            __builtin_amdgcn_sched_barrier(0);                                                              \
            wait_lds();   /* the reads of slot p - 1 (and of slot p) have left LDS: the slot may be refilled */ \
            if (p == 0) w.template issue<NSI - 1>(C0_);                                                     \
            else if (NEXT_) lds_issue_dyn<GS, NT, NCH>(w, p - 1, CN_);                                      \
        }                                                                                                   \
    }
    bool pend = false;
    if (nmy > 0) {
        lds_issue_range<GS, NT, NCH>(w, c);   // slots 0 .. NSI-2 of the first image
        int left = nmy;
        for (; left > NCH; left -= NCH) {
            EVC_LDS_IMAGE(true, c, c + NCH * 4 * kLW, NCH)
            c += NCH * 4 * kLW;
        }
        EVC_LDS_IMAGE(false, c, c, left)
        EVC_LDS_MMA(NT - 1, af[(NS - 1) & 1])   // (pend is set: the image held at least one chunk)
    }
#undef EVC_LDS_IMAGE
    // the ragged last chunk of the matrix (fewer than 16 columns): the wave whose turn it is, with guarded loads
    const int64_t ctail = cols & ~(int64_t)(kLW - 1);
    if (ctail < cols && ctail >= cbeg && ctail < cend && (int)(((ctail - cbeg) / kLW) & 3) == wave) {
        wait_vm<0>();
#pragma unroll
        for (int gs = 0; gs < GS; ++gs) {
            const int slot = 16 * gs + l15;
            const double *vr = P.v + (int64_t)(g0 + (slot < G ? slot : 0)) * P.vstride;
#pragma unroll
            for (int u = 0; u < 2; ++u) bf[gs][u] = ld2g(vr, ctail + 8 * u + 2 * l4, cols);
        }
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            const double *ar = P.A + min(row_base + 16 * t + l15, rows - 1) * ld;
#pragma unroll
            for (int u = 0; u < 2; ++u) af[0][u] = ld2g(ar, ctail + 8 * u + 2 * l4, cols);
            EVC_LDS_MMA(t, af[0])
        }
    }
#undef EVC_LDS_MMA
    EVC_K5L_WG(1);
    // Cross-wave sum through the (now idle) images, one geometry set per pass, as a reduce-scatter: tile tt belongs to
    // wave tt & 3; the other three waves file their accumulators of it in the accumulator's own layout
    // ([tile][copy][half][lane][2 doubles]: 16-byte LDS accesses, lanes 16 bytes apart), the owner adds the four in wave order --
    // (w0 + w1) + (w2 + w3), whoever owns the tile -- and stores the tile straight from its registers.
    wait_vm<0>();
    d2v *red = reinterpret_cast<d2v *>(lds_img);
#pragma unroll
    for (int gs = 0; gs < GS; ++gs) {
        __syncthreads();
        if (gs == 0) EVC_K5L_WG(4);
#pragma unroll
        for (int tt = 0; tt < NT; ++tt) {
            if ((tt & 3) != wave) {
                const int k = (wave - (tt & 3) - 1) & 3;   // 0..2: wave owner+1+k
                // (two 16-byte planes: consecutive lanes 16 bytes apart -- a 32-byte lane pitch is a 2-way conflict)
                red[((tt * 3 + k) * 2 + 0) * 64 + lane] = (d2v){acc[gs][tt][0], acc[gs][tt][1]};
                red[((tt * 3 + k) * 2 + 1) * 64 + lane] = (d2v){acc[gs][tt][2], acc[gs][tt][3]};
            }
        }
        if (gs == 0) EVC_K5L_WG(5);
        __syncthreads();
        if (gs == 0) EVC_K5L_WG(6);
        const int g = 16 * gs + l15;
        double *dst = P.partial + (int64_t)(g0 + g) * P.pstride + (int64_t)span * rows + row_base + l4;
#pragma unroll
        for (int tt = 0; tt < NT; ++tt) {
            if ((tt & 3) == wave && tt < ntile) {
                d4 v[4];
                v[tt & 3] = acc[gs][tt];
#pragma unroll
                for (int k = 0; k < 3; ++k) {
                    const d2v lo = red[((tt * 3 + k) * 2 + 0) * 64 + lane], hi = red[((tt * 3 + k) * 2 + 1) * 64 + lane];
                    v[((tt & 3) + 1 + k) & 3] = (d4){lo[0], lo[1], hi[0], hi[1]};
                }
                const d4 sum = (v[0] + v[1]) + (v[2] + v[3]);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int64_t row = row_base + tt * 16 + l4 + 4 * r;
                    if (row < rows && g < G) dst[tt * 16 + 4 * r] = sum[r];
                }
            }
        }
        if (gs == 0) EVC_K5L_WG(7);
    }
    EVC_K5L_WG(2);
}

// ---------------------------------------------------------------------------------- host side
constexpr int kLdsBlocks = 242;     // workgroups of the large problem: one per CU (the images fill the LDS), one round
constexpr int kLdsBlocksSmall = 8;  // ... of the small (one-body) problem

// These kernels put ONE workgroup on a CU (their LDS images fill it) and size their grids for a whole MI355X: a device
// (or partition) with fewer CUs than blocks would run them in several rounds -- there the kernels of gemv_mfma.hip stay.
static bool lds_device_fits() {
    static std::atomic<uint64_t> known{0}, fits{0};   // bit d: device d examined / large enough
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return false;
    const uint64_t bit = (uint64_t)1 << dev;
    if (!(known.load(std::memory_order_acquire) & bit)) {
        int cus = 0;
        if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) cus = 0;
        if (cus >= kLdsBlocks + kLdsBlocksSmall) fits.fetch_or(bit, std::memory_order_release);
        known.fetch_or(bit, std::memory_order_release);
    }
    return (fits.load(std::memory_order_acquire) & bit) != 0;
}

static bool rows_lds_enabled() {
    static const bool on = !(getenv("EVC_ROWS_LDS") && atoi(getenv("EVC_ROWS_LDS")) == 0);
    return on && lds_device_fits();
}

// Kernel shapes (tiles per row group, chunks per image): every shape keeps 16-18 tiles = 32-36 KB in flight per wave.
// Fewer tiles per row group = more row groups = fewer column spans for the same number of workgroups: the partial sums
// (spans x rows x geometries, written by every workgroup at the same moment and flushed at the end of the launch) shrink
// -- measured at H30 / T = 20 / 32 geometries they cost 3 us in the epilogue and 3 us at the kernel boundary with 242
// spans -- while the geometry vectors are fetched once per row group (the row groups of a span share an XCD: L2 hits).
struct LdsShape {
    int nt, nch;
};
static const LdsShape kLdsShapes[] = {{14, 1}, {7, 2}, {4, 3}, {2, 4}};

// Tiles per row group for a matrix of `rows` rows (EVC_ROWS_LDS_NT forces a shape): the largest shape that makes at
// least two row groups and pads the matrix by less than 1/8 (tiles fetched beyond its rows), else the largest one
// that pads by less than 1/8.  Measured at H30 / T = 20 (14 tiles), 32 geometries, rocprofv3 averages of the launch
// and of rows_reduce_kernel behind it: 14 tiles per group 52.7 + 5.9 us, 7: 50.1 + 4.6, 4: 51.2 + 4.5, 2: 51.5 + 0
// (33 spans: summed inside the eigensolver kernel, +2 us there); the fragment-shaped kernel 57.4 + 4.6.
static int lds_pick_nt(int64_t rows) {
    static const int forced = getenv("EVC_ROWS_LDS_NT") ? atoi(getenv("EVC_ROWS_LDS_NT")) : 0;
    const int nt = (int)ceil_div(rows > 0 ? rows : 1, 16);
    static const int order[] = {7, 4, 2, 14};
    int best = 0;
    for (int want_groups = 2; want_groups >= 1 && !best; --want_groups)
        for (int cand : order) {
            const int nrg = (int)ceil_div(nt, cand);
            if (!best && nrg >= want_groups && nrg * cand * 8 <= nt * 9 && nrg <= kLdsBlocks / 8) best = cand;
        }
    if (!best) {   // few tiles: the shape that fetches the fewest tiles, the larger one on a tie
        int fewest = 1 << 30;
        for (int cand : order) {
            const int tiles = (int)ceil_div(nt, cand) * cand;
            if (tiles < fewest || (tiles == fewest && cand > best)) {
                fewest = tiles;
                best = cand;
            }
        }
    }
    for (const LdsShape &sh : kLdsShapes)
        if (forced == sh.nt) best = forced;
    return best;
}

static int lds_row_groups(int64_t rows, int nt) { return (int)ceil_div(ceil_div(rows > 0 ? rows : 1, 16), nt); }

// Span plan of the LDS-staged kernel: spans of 64 m columns (m 16-column chunks for each of the four waves),
// as many as fit `budget` workgroups in one round; row groups of `ntg` tiles.
static void plan_one_lds(RowProblem &P, int budget, int ntg) {
    const int nrg = lds_row_groups(P.rows, ntg);
    int spans = budget / nrg;
    if (spans < 1) spans = 1;
    const int64_t chunks = ceil_div(P.cols, kLW);
    int64_t m = ceil_div(chunks, 4 * (int64_t)spans);
    if (m < 1) m = 1;
    P.span_cols = 4 * kLW * m;
    P.nspans = (int)ceil_div(P.cols, P.span_cols);
    P.nblocks = nrg * P.nspans;
    P.lds_plan = ntg;
}

bool rows_lds_applicable(const RowProblem &p0, const RowProblem &p1) {
    // (narrow matrices stay with the fragment-shaped kernel: nothing to stream; EVC_ROWS_LDS_MINCOLS=1 sends the small
    //  shapes of the parity tests through this kernel)
    static const int64_t min_cols = getenv("EVC_ROWS_LDS_MINCOLS") ? atoll(getenv("EVC_ROWS_LDS_MINCOLS")) : 4096;
    if (!rows_lds_enabled() || p0.rows <= 0 || p0.cols < min_cols) return false;
    // 32-bit lane offsets inside a tile
    if (16 * p0.ld * 8 >= ((int64_t)1 << 31) || 16 * p1.ld * 8 >= ((int64_t)1 << 31)) return false;
    return lds_row_groups(p0.rows, 14) <= kLdsBlocks / 8;
}

// the small (one-body) problem rides in the launch of the large one if its row groups fit a few workgroups; a tall one
// (T^2 rows: large training sets) keeps its own launch of the fragment-shaped kernel (plan_rows)
static bool lds_small_rides(const RowProblem &p1) { return lds_row_groups(p1.rows, 14) <= kLdsBlocksSmall; }

// Both problems run in ONE launch, i.e. in one kernel shape (the large problem's); together they fill one round.  A tall
// second problem (T^2 rows: large training sets) gets a launch of its own -- of this kernel too if its row groups leave
// room for a few spans, else of the fragment-shaped kernel (plan_rows).
static bool lds_tall_alone(const RowProblem &p1) {
    return p1.cols >= 4 * kLW && lds_row_groups(p1.rows, 14) <= (kLdsBlocks + kLdsBlocksSmall) / 4;
}
void plan_rows_lds(RowProblem &p0, RowProblem &p1) {
    const int ntg = lds_pick_nt(p0.rows);
    if (lds_small_rides(p1)) plan_one_lds(p1, kLdsBlocksSmall, ntg);
    else if (lds_tall_alone(p1)) {
        plan_one_lds(p1, kLdsBlocks + kLdsBlocksSmall, 14);
        p1.lds_plan = -14;   // (< 0: planned for a launch of its own)
    } else {
        plan_rows(p1, true);
        p1.lds_plan = 0;
    }
    plan_one_lds(p0, kLdsBlocks + kLdsBlocksSmall - (p1.lds_plan > 0 ? p1.nblocks : 0), ntg);
}

// most spans any plan makes of this problem (the partial buffers are carved for it)
int rows_max_spans(const RowProblem &P, bool small) {
    RowProblem a = P, b = P;
    plan_rows(a, true);
    // (the 14-tile shape has the fewest row groups, hence the most spans; a tall small problem is planned for a whole
    //  round of its own)
    const int nrg = lds_row_groups(P.rows, 14);
    int spans = (small && nrg <= kLdsBlocksSmall ? kLdsBlocksSmall : kLdsBlocks + kLdsBlocksSmall) / nrg;
    if (spans < 1) spans = 1;
    const int64_t chunks = ceil_div(P.cols > 0 ? P.cols : 1, kLW);
    const int64_t m = ceil_div(chunks, 4 * (int64_t)spans);
    b.nspans = (int)ceil_div(P.cols > 0 ? P.cols : 1, 4 * kLW * (m < 1 ? 1 : m));
    return a.nspans > b.nspans ? a.nspans : b.nspans;
}

// Blocks are dealt to the 8 XCDs round-robin by their index (each XCD has its own L2): the row groups of one span,
// which read the same pieces of the geometry vectors, get indices that are congruent mod 8 as long as the XCD has
// `nrg` indices left; the last spans take what remains.  The small problem's blocks come first.
static void lds_block_map(const GemvRowsLaunch &L, int nb1, int nb0, LdsBlockMap &M) {
    int n = 0;
    for (int b = 0; b < nb1; ++b) M.e[n++] = (unsigned short)(1u << 15 | (unsigned)(b % L.nrg[1]) << 8 | (unsigned)(b / L.nrg[1]));
    int next[8], left[8];   // next free index of each XCD, indices it has left
    for (int x = 0; x < 8; ++x) {
        next[x] = nb1 + ((x - nb1) & 7);
        left[x] = next[x] < nb1 + nb0 ? (nb1 + nb0 - 1 - next[x]) / 8 + 1 : 0;
    }
    const int nrg = L.nrg[0];
    int x = nb1 & 7;
    for (int s = 0; s < L.p[0].nspans; ++s) {
        int pick = -1;
        for (int k = 0; k < 8 && pick < 0; ++k)   // round-robin over the XCDs that can still take a whole span
            if (left[(x + k) & 7] >= nrg) pick = (x + k) & 7;
        for (int rg = 0; rg < nrg; ++rg) {
            int xx = pick;
            if (xx < 0) {   // no XCD has room for the whole span: the one with the most indices left, block by block
                xx = 0;
                for (int k = 1; k < 8; ++k)
                    if (left[k] > left[xx]) xx = k;
            }
            M.e[next[xx]] = (unsigned short)((unsigned)rg << 8 | (unsigned)s);
            next[xx] += 8;
            --left[xx];
        }
        if (pick >= 0) x = (pick + 1) & 7;
    }
}

template <int GS, int NT, int NCH>
static int lds_launch(const GemvRowsLaunch &L, const LdsBlockMap &M, int nblocks, int g0, int G, hipStream_t st) {
    constexpr int lds = 4 * (NT + GS) * NCH * kTileBytes;
    static LdsAttr attr;
    if (int rc = allow_dynamic_lds(gemv_rows_lds_kernel<GS, NT, NCH>, attr, lds, "gemv_rows_lds")) return rc;
    hipLaunchKernelGGL((gemv_rows_lds_kernel<GS, NT, NCH>), dim3(nblocks), dim3(256), lds, st, L, M, g0, G);
    note_kernel(EVC_PROF_ROWS, "gemv_rows_lds_kernel<%d,%d,%d> G=%d", GS, NT, NCH, G);
    return 0;
}

// most geometries one launch takes with the span plan of `p0`: 64 (four sets) with row groups of <= 7 tiles, else 32
int rows_lds_max_g(const RowProblem &p0, const RowProblem &p1) {
    const int ntg = p0.nblocks ? p0.lds_plan : p1.lds_plan;
    return (ntg >= 1 && ntg <= 7) ? 64 : 32;
}

int launch_gemv_rows_lds(const GemvRowsLaunch &Lin, int g0, int G, hipStream_t st) {
    GemvRowsLaunch L = Lin;
    // one kernel shape per launch: the large problem's (the small problem's row groups are dealt for the same shape)
    const int ntg = L.p[0].nblocks ? L.p[0].lds_plan : L.p[1].lds_plan;
    for (int k = 0; k < 2; ++k) {
        const int nt = (int)ceil_div(L.p[k].rows > 0 ? L.p[k].rows : 1, 16);
        L.nrg[k] = (int)ceil_div(nt, ntg);
        L.tpg[k] = nt / L.nrg[k];
        L.trem[k] = nt % L.nrg[k];
        if (!L.p[k].nblocks) L.p[k].nspans = 0;
        EVC_REQUIRE(L.p[k].span_cols % kLW == 0, "gemv_rows_lds: span of %lld columns", (long long)L.p[k].span_cols);
        EVC_REQUIRE(!L.p[k].nblocks || (aligned16(L.p[k].A) && aligned16(L.p[k].v) && L.p[k].ld % 2 == 0 &&
                                        L.p[k].vstride % 2 == 0),
                    "gemv_rows_lds: operands must be 16-byte aligned with even pitches");
        EVC_REQUIRE(L.p[k].nspans <= 255 && L.nrg[k] <= 127, "gemv_rows_lds: %d spans x %d row groups", L.p[k].nspans,
                    L.nrg[k]);
    }
    const int nb0 = L.nrg[0] * L.p[0].nspans, nb1 = L.nrg[1] * L.p[1].nspans;
    EVC_REQUIRE(nb0 + nb1 > 0 && nb0 + nb1 <= 256, "gemv_rows_lds: %d blocks", nb0 + nb1);
    L.nblk0 = nb0;
    L.nblk1 = nb1;
    LdsBlockMap M;
    lds_block_map(L, nb1, nb0, M);
    int rc = -1;
    // (four geometry sets, 33 .. 64 geometries per pass: 8 accumulator registers per (set, tile) -- 7 tiles at most --
    //  and fewer chunks per image, the four vector tiles of a chunk count)
#define EVC_LDS_CASE(NT_, NCH_, NCH4_)                                                 \
    case NT_:                                                                          \
        if (G > 32) {                                                                  \
            if constexpr (NT_ <= 7) rc = lds_launch<4, NT_, NCH4_>(L, M, nb0 + nb1, g0, G, st);                 \
            else { set_error("gemv_rows_lds: %d geometries need <= 7 tiles per row group", G); return -1; }     \
        } else                                                                         \
            rc = G > 16 ? lds_launch<2, NT_, NCH_>(L, M, nb0 + nb1, g0, G, st)         \
                        : lds_launch<1, NT_, NCH_>(L, M, nb0 + nb1, g0, G, st);        \
        break;
    switch (ntg) {
        EVC_LDS_CASE(14, 1, 1) EVC_LDS_CASE(7, 2, 1) EVC_LDS_CASE(4, 3, 2) EVC_LDS_CASE(2, 4, 2)
        default: set_error("gemv_rows_lds: no kernel for %d tiles per row group", ntg); return -1;
    }
#undef EVC_LDS_CASE
    if (rc) return rc;
    EVC_LAUNCH_CHECK("gemv_rows_lds");
    return 0;
}

// ================================================================================== K8 through LDS-DMA
//   O[g][c] = sum_r w[g][r] A[r][c]      (reference: ab_initio_gradients_loewdin.py:343-356, the predicted RDMs as the
//   weighted sum of the stored transition RDMs)
// M <-> 16 geometries, N <-> 16 columns, K <-> 4 rows per MFMA.  A wave owns whole 16-column tiles (128 bytes = one cache
// line per row) and sums over ALL rows itself: no partial sums, no cross-wave reduction, no barrier after the weights
// are staged.  The matrix streams through a per-wave ring of D pieces (a piece = one LDS-DMA instruction = 8 rows x
// 128 bytes, in the order of the wave's tiles), D - 1 of them in flight; the weights of the whole launch sit in LDS as
// wl[row][32 geometries] (the two halves of a row swapped on odd rows: the two rows a 32-lane group reads fall on
// different banks).  Per piece: 2 + 4 ds_read_b64, 4 MFMAs (two K steps x two geometry sets), one refill.
// A tile's pieces are padded to an even number (weights of the padding rows are zero, its loads re-read the last row), so
// the loop body is two pieces with alternating fragment registers: the MFMAs of a piece run behind the fragment reads of
// the next one.
template <int GS, int D, int NW>
__device__ __forceinline__ void cols_lds_body(const ColProblem &P, int g0, int G, int blk, int nblk, double *lds, int wave,
                                              int lane) {
    const int l15 = lane & 15, l4 = lane >> 4;
    const int rows = (int)P.rows;
    const int64_t cols = P.cols, ld = P.ld;
    const int NP = (rows + 7) >> 3;          // pieces per tile
    const int H = (NP + 1) >> 1;             // loop iterations per tile (two pieces each)
    const int NPe = 2 * H;
    const int rows_w = NPe * 8;              // weight rows in LDS (zero beyond the matrix)
    // ---- weights -> LDS (the only barrier of the kernel)
    for (int idx = threadIdx.x; idx < rows_w * 32; idx += 64 * NW) {
        const int r = idx >> 5, sl = idx & 31;
        double v = 0.0;
        if (r < rows && sl < G && sl < 16 * GS) {
            const int gg = g0 + sl;
            v = P.wt ? P.wt[(int64_t)(gg - gg % kMaxBatchG) * P.wstride + (int64_t)r * kMaxBatchG + gg % kMaxBatchG]
                     : P.w[(int64_t)gg * P.wstride + r];
        }
        lds[r * 32 + ((((sl >> 4) ^ (r & 1)) & 1) << 4) + (sl & 15)] = v;
    }
    __syncthreads();
    const int ntiles = (int)((cols + 15) >> 4);
    // tiles of this wave: (k nblk + blk) NW + wave, k = 0, 1, ... (the waves of a workgroup take adjacent tiles)
    const int tstride = nblk * NW;
    int ctile = blk * NW + wave;             // consumer's tile
    if (ctile >= ntiles) return;
    const unsigned ring = (unsigned)(rows_w * 256 + wave * D * 1024);   // LDS byte address of this wave's ring
    // producer lane constants: its 16 bytes of a piece
    const int prow = lane >> 3, ppos = lane & 7;
    const int last_valid = rows - 1 - 8 * (NP - 1);   // last valid row of the last piece, relative to it
    const unsigned voff = (unsigned)(((int64_t)prow * ld + 2 * ppos) * 8);
    const unsigned voff_last = (unsigned)(((int64_t)(prow < last_valid ? prow : last_valid) * ld + 2 * ppos) * 8);
    const unsigned voff_dummy = (unsigned)(2 * ppos * 8);
    const char *Ab = reinterpret_cast<const char *>(P.A);
    // producer cursor
    int ptile = ctile, ppiece = 0, pslot = 0;
    bool pdone = false;
    // one piece into ring slot pslot, cursor advanced: scalar work only (the vector pipe belongs to the FP64 MFMAs --
    // a vector instruction of this wave waits for them) and a short common path: a full piece of a tile that does not
    // hold the matrix's last columns; `psb` runs along the rows of the producer's tile
    const int64_t piece_step = 8 * ld * 8;
    const int tail_tile = (int)(ld >> 4) < ntiles && (((int64_t)(ld >> 4) * 16 + 16) > ld) ? (int)(ld >> 4) : -1;
    const char *psb = Ab + (int64_t)ptile * 128;
    int pfast = ptile == tail_tile ? 0 : NP - 1;   // pieces of the producer's tile that take the common path
#define EVC_COLS_PRODUCE()                                                                                       \
    {                                                                                                            \
        const unsigned la = uniform_u32(ring + (unsigned)pslot * 1024);                                          \
        if (ppiece < pfast) glds_s(voff, uniform_ptr(psb), la);                                                  \
        else {                                                                                                   \
            unsigned back = 0; /* the tile with the matrix's last columns: stay inside the row (ld is even) */   \
            if (ptile == tail_tile) {                                                                            \
                const int over = 2 * ppos - (int)(ld - 2 - (int64_t)ptile * 16);                                 \
                back = over > 0 ? (unsigned)(over * 8) : 0u;                                                     \
            }                                                                                                    \
            if (ppiece < NP - 1) glds_s(voff - back, uniform_ptr(psb), la);                                      \
            else if (ppiece == NP - 1) glds_s(voff_last - back, uniform_ptr(psb), la);                           \
            else glds_s(voff_dummy - back, uniform_ptr(Ab + ((int64_t)(rows - 1) * ld + (int64_t)ptile * 16) * 8), la); \
        }                                                                                                        \
        pslot = pslot + 1 == D ? 0 : pslot + 1;                                                                  \
        psb += piece_step;                                                                                       \
        if (++ppiece == NPe) {                                                                                   \
            ppiece = 0;                                                                                          \
            ptile += tstride;                                                                                    \
            psb = Ab + (int64_t)ptile * 128;                                                                     \
            pfast = ptile == tail_tile ? 0 : NP - 1;                                                             \
            if (ptile >= ntiles) pdone = true;                                                                   \
        }                                                                                                        \
    }
#pragma unroll 1
    for (int i = 0; i < D - 1; ++i) {
        if (!pdone) EVC_COLS_PRODUCE()
        else {   // (fewer pieces than ring slots: keep the in-order count with re-reads of the last row)
            const char *sb = Ab + (int64_t)(rows - 1) * ld * 8;
            glds_s(voff_dummy, uniform_ptr(sb), uniform_u32(ring + (unsigned)pslot * 1024));
            pslot = pslot + 1 == D ? 0 : pslot + 1;
        }
    }
    // consumer
    const unsigned xlane = (unsigned)(l4 * 128 + l15 * 8);                       // B fragment within a piece
    unsigned wl0[GS];                                                           // weight fragment of piece 0, K step 0
#pragma unroll
    for (int gs = 0; gs < GS; ++gs) wl0[gs] = (unsigned)(l4 * 256 + (((gs ^ (l4 & 1)) & 1) << 7) + l15 * 8);
    const char *ldsb = reinterpret_cast<const char *>(lds);
    int cslot = 0;
    d4 acc[GS];
    double xa[2], xb[2], wa[2][GS], wb[2][GS];
#define EVC_COLS_READ(X_, W_, PIECE_)                                                                         \
    {                                                                                                        \
        const char *xp = ldsb + ring + (unsigned)cslot * 1024 + xlane;                                       \
        X_[0] = *reinterpret_cast<const double *>(xp);                                                       \
        X_[1] = *reinterpret_cast<const double *>(xp + 512);                                                 \
        _Pragma("unroll") for (int gs = 0; gs < GS; ++gs) {                                                  \
            const char *wp = ldsb + (unsigned)(PIECE_) * 2048 + wl0[gs];                                     \
            W_[0][gs] = *reinterpret_cast<const double *>(wp);                                               \
            W_[1][gs] = *reinterpret_cast<const double *>(wp + 1024);                                        \
        }                                                                                                    \
        cslot = cslot + 1 == D ? 0 : cslot + 1;                                                              \
    }
#define EVC_COLS_MMA_K(X_, W_, K_)                                                                           \
    {                                                                                                        \
        _Pragma("unroll") for (int gs = 0; gs < GS; ++gs) acc[gs] = mfma64(W_[K_][gs], X_[K_], acc[gs]);     \
    }
#define EVC_COLS_MMA(X_, W_)                                                                                 \
    {                                                                                                        \
        EVC_COLS_MMA_K(X_, W_, 0)                                                                            \
        EVC_COLS_MMA_K(X_, W_, 1)                                                                            \
    }
// Position n of the wave's piece sequence: wait for piece n (D - 2 pieces are younger while the producer runs), read
// its fragments, start the MFMAs of piece n - 1, and refill the slot read at position n - 1 BETWEEN the two halves of
// those MFMAs: the scalar work of the refill and the LDS latency of piece n hide behind them.
#define EVC_COLS_STEP(X_, W_, PIECE_, XP_, WP_, MMA_)                                                        \
    {                                                                                                        \
        if (pdone) wait_vm<0>();                                                                             \
        else wait_vm<D - 2>();                                                                               \
        EVC_COLS_READ(X_, W_, PIECE_)                                                                        \
        __builtin_amdgcn_sched_barrier(0);                                                                   \
        if (MMA_) EVC_COLS_MMA_K(XP_, WP_, 0)                                                                \
        __builtin_amdgcn_sched_barrier(0);                                                                   \
        wait_lds();                                                                                          \
        if (!pdone) EVC_COLS_PRODUCE()                                                                       \
        __builtin_amdgcn_sched_barrier(0);                                                                   \
        if (MMA_) EVC_COLS_MMA_K(XP_, WP_, 1)                                                                \
        __builtin_amdgcn_sched_barrier(0);                                                                   \
    }
    for (; ctile < ntiles; ctile += tstride) {
#pragma unroll
        for (int gs = 0; gs < GS; ++gs) acc[gs] = (d4){0.0, 0.0, 0.0, 0.0};
        for (int it = 0; it < H; ++it) {
            EVC_COLS_STEP(xa, wa, 2 * it, xb, wb, it > 0)
            EVC_COLS_STEP(xb, wb, 2 * it + 1, xa, wa, true)
        }
        EVC_COLS_MMA(xb, wb)
        // D[i][j]: i = geometry l4 + 4 reg, j = column l15
        const int64_t c = (int64_t)ctile * 16 + l15;
        if (c < cols) {
#pragma unroll
            for (int gs = 0; gs < GS; ++gs)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int g = 16 * gs + l4 + 4 * r;
                    if (g < G) P.out[(int64_t)(g0 + g) * P.ostride + c] = acc[gs][r];
                }
        }
    }
#undef EVC_COLS_READ
#undef EVC_COLS_MMA
#undef EVC_COLS_MMA_K
#undef EVC_COLS_STEP
#undef EVC_COLS_PRODUCE
    wait_vm<0>();
}

// Tall matrices (large training sets: more rows than weights fit in LDS): the rows go in SLABS of kSlabRows; a wave owns
// NTW ADJACENT column tiles (one workgroup round), keeps their accumulators and walks slab by slab, piece by piece, tile by
// tile -- so it still sums over all rows itself (no partial sums), a piece's weights are read once for all its tiles, and
// a workgroup's eight waves read 8 NTW x 128 contiguous bytes of every row within a few steps (DRAM pages are reused:
// tile-by-tile order, each 128-byte piece of a row on its own, streamed a cold 3.4 GB matrix at 4.4 TB/s).
// The weights of slab s + 1 are fetched by LDS-DMA from the transposed copy `wt` (rows x 32, contiguous) into the second
// weight buffer while slab s is in work: each wave issues its share at the start of slab s, and the constant vmcnt
// waits of the piece stream retire them long before the slab ends, so the slab boundary is ONE LDS-only barrier and the
// ring never drains.  (The one-body problem has no transposed weights: its few workgroups stage a slab with plain
// loads between two barriers.)  Waves at the end of the matrix run their missing tiles on the last tile (not stored).
constexpr int kSlabRows = 160;          // 20 pieces; two weight buffers of 40 KB + eight rings of 10 KB = 160 KB
constexpr int kSlabRing = 10;
template <int GS, int NTW>
__device__ __forceinline__ void cols_lds_slab_body(const ColProblem &P, int g0, int G, int blk, int nblk, double *lds,
                                                   int wave, int lane) {
    constexpr int NW = 8, D = kSlabRing;
    constexpr unsigned WB = kSlabRows * 256;   // bytes of one weight buffer
    const int l15 = lane & 15, l4 = lane >> 4;
    const int rows = (int)P.rows;
    const int64_t cols = P.cols, ld = P.ld;
    const int nslab = (rows + kSlabRows - 1) / kSlabRows;
    const int ntiles = (int)((cols + 15) >> 4);
    const bool dma_w = P.wt != nullptr;        // workgroup-uniform
    const int tile0 = (blk * NW + wave) * NTW; // this wave's tiles: tile0 .. tile0 + NTW - 1
    const bool active = tile0 < ntiles;
    // pieces of slab s (even: the weights of the padding rows are zero, its loads re-read the matrix's last row)
    auto slab_pieces = [&](int sidx) {
        const int left = rows - sidx * kSlabRows;
        const int np = ((left < kSlabRows ? left : kSlabRows) + 7) >> 3;
        return (np + 1) & ~1;
    };
    const unsigned ring = 2 * WB + (unsigned)(wave * D * 1024);   // LDS byte address of this wave's ring
    const int prow = lane >> 3, ppos = lane & 7;
    const char *Ab = reinterpret_cast<const char *>(P.A);
    const int tail_tile = (int)(ld >> 4) < ntiles && (((int64_t)(ld >> 4) * 16 + 16) > ld) ? (int)(ld >> 4) : -1;
    const unsigned voff_full = (unsigned)(((int64_t)prow * ld + 2 * ppos) * 8);
    // weights of slab `sidx` into buffer `buf` by LDS-DMA: instruction k covers rows 4k .. 4k+3 of the slab (1 KB, lane i ->
    // row 4k + (i >> 4), 16-byte piece i & 15); the piece holds geometries 16 (h ^ (row & 1)) + 2 (i & 7), + 1 with
    // h = (i >> 3) & 1: the halves of odd rows swapped (the image the fragment reads expect).  Rows beyond the matrix
    // fetch its last row and are zeroed once the buffer is complete.
    const double *wtb = dma_w ? P.wt + (int64_t)(g0 - g0 % kMaxBatchG) * P.wstride + g0 % kMaxBatchG : nullptr;
    auto issue_weights = [&](int sidx, int buf) {
        const int r00 = sidx * kSlabRows;
        for (int k = wave; k < kSlabRows / 4; k += NW) {
            const int rl = 4 * k + (lane >> 4), r = r00 + rl;
            const int rc = r < rows ? r : rows - 1;
            const int h = (lane >> 3) & 1;
            const double *src = wtb + (int64_t)rc * kMaxBatchG + 16 * (h ^ (rl & 1)) + 2 * (lane & 7);
            glds_v(src, uniform_u32((unsigned)buf * WB + (unsigned)k * 1024));
        }
    };
    // producer cursor: slab, piece, tile (the order the consumer reads in)
    int pslab = 0, pj = 0, ppiece = 0, pslot = 0, pnp = slab_pieces(0);
    bool pdone = !active;
#define EVC_SLAB_PRODUCE()                                                                                       \
    {                                                                                                            \
        const int pt_ = tile0 + pj;                                                                              \
        const int ptile = pt_ < ntiles ? pt_ : ntiles - 1;          /* (a tile the wave does not have) */        \
        const int r0 = pslab * kSlabRows + 8 * ppiece;              /* first row of the piece */                 \
        const int rf = r0 < rows ? r0 : rows - 1;                   /* (padding piece: the last row) */          \
        const int lastv = rows - 1 - rf;                            /* last valid row relative to rf */          \
        unsigned vo = voff_full;                                    /* (common path: no vector arithmetic) */    \
        if (lastv < 7 || ptile == tail_tile) {                                                                   \
            const int rr = prow < lastv ? prow : lastv;                                                          \
            vo = (unsigned)(((int64_t)rr * ld + 2 * ppos) * 8);                                                  \
            if (ptile == tail_tile) { /* stay inside the row (ld is even) */                                     \
                const int over = 2 * ppos - (int)(ld - 2 - (int64_t)ptile * 16);                                 \
                if (over > 0) vo -= (unsigned)(over * 8);                                                        \
            }                                                                                                    \
        }                                                                                                        \
        glds_s(vo, uniform_ptr(Ab + ((int64_t)rf * ld + (int64_t)ptile * 16) * 8),                               \
               uniform_u32(ring + (unsigned)pslot * 1024));                                                      \
        pslot = pslot + 1 == D ? 0 : pslot + 1;                                                                  \
        if (++pj == NTW) {                                                                                       \
            pj = 0;                                                                                              \
            if (++ppiece == pnp) {                                                                               \
                ppiece = 0;                                                                                      \
                if (++pslab == nslab) pdone = true;                                                              \
                else pnp = slab_pieces(pslab);                                                                   \
            }                                                                                                    \
        }                                                                                                        \
    }
    if (dma_w) issue_weights(0, 0);
#pragma unroll 1
    for (int i = 0; i < D - 1; ++i) {
        if (!pdone) EVC_SLAB_PRODUCE()
        else {   // (fewer pieces than ring slots: keep the in-order count with re-reads of the last row)
            glds_s((unsigned)(2 * ppos * 8), uniform_ptr(Ab + (int64_t)(rows - 1) * ld * 8),
                   uniform_u32(ring + (unsigned)pslot * 1024));
            pslot = pslot + 1 == D ? 0 : pslot + 1;
        }
    }
    const unsigned xlane = (unsigned)(l4 * 128 + l15 * 8);
    unsigned wl0[GS];
#pragma unroll
    for (int gs = 0; gs < GS; ++gs) wl0[gs] = (unsigned)(l4 * 256 + (((gs ^ (l4 & 1)) & 1) << 7) + l15 * 8);
    const char *ldsb = reinterpret_cast<const char *>(lds);
    int cslot = 0;
    d4 acc[NTW][GS];
#pragma unroll
    for (int j = 0; j < NTW; ++j)
#pragma unroll
        for (int gs = 0; gs < GS; ++gs) acc[j][gs] = (d4){0.0, 0.0, 0.0, 0.0};
    double x[2][NTW][2];   // [piece parity][tile][K step]
    double wf[2][2][GS];   // [piece parity][K step][geometry set]
// position (piece parity Q_, tile J_): wait for its piece, read the matrix fragments (and, at the first tile, the piece's
// weights), first half of the MFMAs of the position before (PQ_, PJ_), refill, second half
#define EVC_SLAB_STEP(Q_, J_, PIECE_, PQ_, PJ_, MMA_)                                                        \
    {                                                                                                        \
        if (pdone) wait_vm<0>();                                                                             \
        else wait_vm<D - 2>();                                                                               \
        {                                                                                                    \
            const char *xp = ldsb + ring + (unsigned)cslot * 1024 + xlane;                                   \
            x[Q_][J_][0] = *reinterpret_cast<const double *>(xp);                                            \
            x[Q_][J_][1] = *reinterpret_cast<const double *>(xp + 512);                                      \
            if ((J_) == 0) {                                                                                 \
                _Pragma("unroll") for (int gs = 0; gs < GS; ++gs) {                                          \
                    const char *wp = ldsb + wbuf + (unsigned)(PIECE_) * 2048 + wl0[gs];                      \
                    wf[Q_][0][gs] = *reinterpret_cast<const double *>(wp);                                   \
                    wf[Q_][1][gs] = *reinterpret_cast<const double *>(wp + 1024);                            \
                }                                                                                            \
            }                                                                                                \
            cslot = cslot + 1 == D ? 0 : cslot + 1;                                                          \
        }                                                                                                    \
        __builtin_amdgcn_sched_barrier(0);                                                                   \
        if (MMA_) {                                                                                          \
            _Pragma("unroll") for (int gs = 0; gs < GS; ++gs)                                                \
                acc[PJ_][gs] = mfma64(wf[PQ_][0][gs], x[PQ_][PJ_][0], acc[PJ_][gs]);                         \
        }                                                                                                    \
        __builtin_amdgcn_sched_barrier(0);                                                                   \
        wait_lds();                                                                                          \
        if (!pdone) EVC_SLAB_PRODUCE()                                                                       \
        __builtin_amdgcn_sched_barrier(0);                                                                   \
        if (MMA_) {                                                                                          \
            _Pragma("unroll") for (int gs = 0; gs < GS; ++gs)                                                \
                acc[PJ_][gs] = mfma64(wf[PQ_][1][gs], x[PQ_][PJ_][1], acc[PJ_][gs]);                         \
        }                                                                                                    \
        __builtin_amdgcn_sched_barrier(0);                                                                   \
    }
    for (int sidx = 0; sidx < nslab; ++sidx) {
        const unsigned wbuf = dma_w ? (unsigned)(sidx & 1) * WB : 0u;
        const int r00 = sidx * kSlabRows;
        if (dma_w) {
            // every DMA of this slab's weights (issued a slab ago, or in the prologue) has been retired by the piece
            // waits since; a wave that had no pieces to wait for waits here
            if (sidx == 0 || !active || pdone) wait_vm<0>();
            lds_barrier();   // ... by EVERY wave: the buffer is complete; and nobody reads the other buffer any more
            // rows of the slab beyond the matrix: zero weights (they were fetched from the last row)
            if (r00 + kSlabRows > rows) {
                const int first = rows > r00 ? rows - r00 : 0;
                for (int idx = threadIdx.x; idx < (kSlabRows - first) * 32; idx += 64 * NW)
                    lds[wbuf / 8 + (first + (idx >> 5)) * 32 + (idx & 31)] = 0.0;
                lds_barrier();
            }
            if (sidx + 1 < nslab) issue_weights(sidx + 1, (sidx + 1) & 1);
        } else {
            __syncthreads();   // every wave has read the last weights of the slab before
            for (int idx = threadIdx.x; idx < kSlabRows * 32; idx += 64 * NW) {
                const int rl = idx >> 5, sl = idx & 31, r = r00 + rl;
                double v = 0.0;
                if (r < rows && sl < G && sl < 16 * GS) v = P.w[(int64_t)(g0 + sl) * P.wstride + r];
                lds[rl * 32 + ((((sl >> 4) ^ (rl & 1)) & 1) << 4) + (sl & 15)] = v;
            }
            __syncthreads();
        }
        if (active) {
            const int H = slab_pieces(sidx) >> 1;
            for (int it = 0; it < H; ++it) {
#pragma unroll
                for (int j = 0; j < NTW; ++j) {   // piece 2 it; the position before: (piece 2 it - 1, last tile) or (this, j - 1)
                    if (j == 0) EVC_SLAB_STEP(0, 0, 2 * it, 1, NTW - 1, it > 0)
                    else EVC_SLAB_STEP(0, j, 2 * it, 0, (j > 0 ? j - 1 : 0), true)
                }
#pragma unroll
                for (int j = 0; j < NTW; ++j) {   // piece 2 it + 1
                    if (j == 0) EVC_SLAB_STEP(1, 0, 2 * it + 1, 0, NTW - 1, true)
                    else EVC_SLAB_STEP(1, j, 2 * it + 1, 1, (j > 0 ? j - 1 : 0), true)
                }
            }
            // the last position of the slab (its fragments are in registers: the next slab's weights do not matter)
#pragma unroll
            for (int k = 0; k < 2; ++k)
#pragma unroll
                for (int gs = 0; gs < GS; ++gs)
                    acc[NTW - 1][gs] = mfma64(wf[1][k][gs], x[1][NTW - 1][k], acc[NTW - 1][gs]);
        }
    }
#pragma unroll
    for (int j = 0; j < NTW; ++j) {
        const int64_t c = (int64_t)(tile0 + j) * 16 + l15;
        if (active && tile0 + j < ntiles && c < cols) {
#pragma unroll
            for (int gs = 0; gs < GS; ++gs)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int g = 16 * gs + l4 + 4 * r;
                    if (g < G) P.out[(int64_t)(g0 + g) * P.ostride + c] = acc[j][gs][r];
                }
        }
    }
#undef EVC_SLAB_STEP
#undef EVC_SLAB_PRODUCE
    wait_vm<0>();
}

template <int GS, int NTW>
__global__ __launch_bounds__(512, 1) void gemv_cols_lds_slab_kernel(GemvColsLaunch L, int nblk1, int g0, int G) {
    extern __shared__ __align__(16) double lds_cols_slab[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
    // the few blocks of the small second problem are dispatched first
    if ((int)blockIdx.x < nblk1) cols_lds_slab_body<GS, 1>(L.p[1], g0, G, blockIdx.x, nblk1, lds_cols_slab, wave, lane);
    else cols_lds_slab_body<GS, NTW>(L.p[0], g0, G, blockIdx.x - nblk1, gridDim.x - nblk1, lds_cols_slab, wave, lane);
}

template <int GS, int D0, int D1, int NW>
__global__ __launch_bounds__(64 * NW, 1) void gemv_cols_lds_kernel(GemvColsLaunch L, int nblk1, int g0, int G) {
    extern __shared__ __align__(16) double lds_cols[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
    // the few blocks of the small second problem are dispatched first
    if ((int)blockIdx.x < nblk1) cols_lds_body<GS, D1, NW>(L.p[1], g0, G, blockIdx.x, nblk1, lds_cols, wave, lane);
    else cols_lds_body<GS, D0, NW>(L.p[0], g0, G, blockIdx.x - nblk1, gridDim.x - nblk1, lds_cols, wave, lane);
}

// ring depth of a problem with `rows` rows and nw waves per workgroup: 24, 12 or 6 pieces, 0: the weights
// (rows x 32) and the rings do not fit the LDS
static int cols_lds_depth(int64_t rows, int nw) {
    const int64_t rows_w = ((rows + 15) / 16) * 16;
    const int64_t d = (160 * 1024 - rows_w * 256) / (nw * 1024);
    return d >= 24 ? 24 : (d >= 12 ? 12 : (d >= 6 ? 6 : 0));
}
// eight waves per workgroup (two per SIMD): with one wave per SIMD and rings of 24 the loop was bound by its own
// scalar instructions (54 us against 43 at H30; removed)
static int cols_lds_waves() { return 8; }

// 0: not applicable; 1: both problems in one launch of gemv_cols_lds_kernel (the weights of all rows fit LDS); 2: tall
// matrix, both problems in one launch of gemv_cols_lds_slab_kernel
int cols_lds_mode(const ColProblem &p0, const ColProblem &p1, int G) {
    static const bool on = !(getenv("EVC_COLS_LDS") && atoi(getenv("EVC_COLS_LDS")) == 0);
    constexpr bool slab_on = true;
    static const int64_t min_cols = getenv("EVC_ROWS_LDS_MINCOLS") ? atoll(getenv("EVC_ROWS_LDS_MINCOLS")) : 4096;
    if (!on || !lds_device_fits() || p0.cols < min_cols || p0.rows <= 0 || !aligned16(p0.A) || p0.ld % 2 ||
        p0.rows > (1 << 20))
        return 0;
    if (16 * p0.ld * 8 >= ((int64_t)1 << 31) || 16 * p1.ld * 8 >= ((int64_t)1 << 31)) return 0;
    // one set of geometries (G <= 16): the row-split kernel of gemv_mfma.hip is faster (31.8 against 36.1 us at H30)
    if (G <= 16) return 0;
    const int nw = cols_lds_waves();
    if (cols_lds_depth(p0.rows, nw) >= 12) {
        if (p1.cols > 0 && (cols_lds_depth(p1.rows, nw) < 6 || !aligned16(p1.A) || p1.ld % 2 || p1.part)) return 0;
        return 1;
    }
    // slab kernel: every wave's tiles in one round of <= 4, transposed weights for the LDS-DMA staging, a small second
    // problem (<= 128 tiles: one per wave of <= 16 workgroups)
    if (!slab_on || !p0.wt || !aligned16(p0.wt)) return 0;
    if (ceil_div(ceil_div(p0.cols, 16), (int64_t)8 * kLdsBlocks) > 4) return 0;
    if (p1.cols > 0 && (p1.cols > 2048 || !aligned16(p1.A) || p1.ld % 2)) return 0;
    return 2;
}

template <int GS, int D0, int D1, int NW>
static int cols_lds_launch(const GemvColsLaunch &L, int nblk1, int nblk0, size_t lds, int g0, int G, hipStream_t st) {
    static LdsAttr attr;
    if (int rc = allow_dynamic_lds(gemv_cols_lds_kernel<GS, D0, D1, NW>, attr, 160 * 1024, "gemv_cols_lds")) return rc;
    hipLaunchKernelGGL((gemv_cols_lds_kernel<GS, D0, D1, NW>), dim3(nblk0 + nblk1), dim3(64 * NW), lds, st, L, nblk1, g0, G);
    note_kernel(EVC_PROF_COLS, "gemv_cols_lds_kernel<%d,%d,%d,%d>", GS, D0, D1, NW);
    return 0;
}

int launch_gemv_cols_lds(const GemvColsLaunch &L, int g0, int G, hipStream_t st) {
    const int nw = cols_lds_waves();
    int d0 = cols_lds_depth(L.p[0].rows, nw);
    if (nw == 8 && d0 > 12) d0 = 12;
    const int d1 = L.p[1].cols > 0 ? (cols_lds_depth(L.p[1].rows, nw) >= 12 ? 12 : 6) : 12;
    const int nblk1 = L.p[1].cols > 0 ? kLdsBlocksSmall : 0;
    // workgroups of the large problem: every wave the same number of 16-column tiles (rounds), in one round of blocks
    const int64_t ntiles = ceil_div(L.p[0].cols, 16);
    const int64_t rounds = ceil_div(ntiles, (int64_t)nw * kLdsBlocks);
    const int nblk0 = (int)ceil_div(ntiles, nw * rounds);
    auto need = [&](const ColProblem &P, int d) {
        const int64_t rows_w = ((P.rows + 15) / 16) * 16;
        return (size_t)(rows_w * 256 + (int64_t)nw * d * 1024);
    };
    size_t lds = need(L.p[0], d0);
    if (L.p[1].cols > 0 && need(L.p[1], d1) > lds) lds = need(L.p[1], d1);
    int rc = -1;
#define EVC_CL_CASE(GS_, D0_, D1_, NW_)                                                              \
    if ((G > 16 ? 2 : 1) == GS_ && d0 == D0_ && d1 == D1_ && nw == NW_)                              \
        rc = cols_lds_launch<GS_, D0_, D1_, NW_>(L, nblk1, nblk0, lds, g0, G, st);
    // (two geometry sets only: one set stays with the row-split kernel, cols_lds_applicable)
    EVC_CL_CASE(2, 12, 12, 8) EVC_CL_CASE(2, 12, 6, 8)
#undef EVC_CL_CASE
    if (rc == -1) set_error("gemv_cols_lds: no kernel for ring depths %d / %d, %d waves", d0, d1, nw);
    if (rc) return rc;
    EVC_LAUNCH_CHECK("gemv_cols_lds");
    return 0;
}

int launch_gemv_cols_lds_slab(const GemvColsLaunch &L, int g0, int G, hipStream_t st) {
    const int64_t ntiles = ceil_div(L.p[0].cols, 16);
    const int64_t rounds = ceil_div(ntiles, (int64_t)8 * kLdsBlocks);
    const int nblk0 = (int)ceil_div(ntiles, 8 * rounds);
    const int nblk1 = L.p[1].cols > 0 ? (int)ceil_div(ceil_div(L.p[1].cols, 16), 8) : 0;
    const size_t lds = (size_t)2 * kSlabRows * 256 + (size_t)8 * kSlabRing * 1024;
    int rc = -1;
#define EVC_SLAB_CASE(NTW_)                                                                                   \
    if (rounds == NTW_) {                                                                                    \
        static LdsAttr attr;                                                                                 \
        if ((rc = allow_dynamic_lds(gemv_cols_lds_slab_kernel<2, NTW_>, attr, 160 * 1024, "gemv_cols_lds_slab"))) \
            return rc;                                                                                       \
        hipLaunchKernelGGL((gemv_cols_lds_slab_kernel<2, NTW_>), dim3(nblk0 + nblk1), dim3(512), lds, st, L, nblk1, g0, G); \
        note_kernel(EVC_PROF_COLS, "gemv_cols_lds_slab_kernel<2,%d>", NTW_);                                  \
    }
    EVC_SLAB_CASE(1) EVC_SLAB_CASE(2) EVC_SLAB_CASE(3) EVC_SLAB_CASE(4)
#undef EVC_SLAB_CASE
    if (rc == -1) {
        set_error("gemv_cols_lds_slab: no kernel for %lld rounds", (long long)rounds);
        return -1;
    }
    EVC_LAUNCH_CHECK("gemv_cols_lds_slab");
    return 0;
}

}  // namespace evc

#ifdef EVC_DEBUG_STAMPS
extern "C" int evc_debug_read_k5l(long long *stamps) {
    return (int)hipMemcpyFromSymbol(stamps, HIP_SYMBOL(evc::g_k5l_wg), sizeof(long long) * 1024 * 8);
}
#endif
