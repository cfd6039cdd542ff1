// Pack / unpack of the index symmetries of two-body quantities (N^4-sized copy kernels).
//   a4/a5   pack / unpack of the electron-exchange symmetry     electron_integral_utils.py:38-88
//   sym8    8-fold compressed vectors <-> dense (pair,pair) / (N,N,N,N) arrays (include/evcont_hip.h EVC_LAYOUT_SYM8)
// blockIdx.y = geometry of the batch (kernels.hpp).
#include <stdlib.h>

#include "common.hpp"
#include "kernels.hpp"

namespace evc {

typedef double d4 __attribute__((ext_vector_type(4)));
typedef double d2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ d4 mfma_f64(double a, double b, d4 c) {
    return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
}
// ------------------------------------------------------------------ pack / unpack
__global__ void pack_kernel(const double *__restrict__ h2, int64_t sh2, int n, double mult, double *__restrict__ out,
                            int64_t sout, int64_t M, int64_t out_len) {
    const int64_t n2 = (int64_t)n * n;
    h2 += (int64_t)blockIdx.y * sh2;
    out += (int64_t)blockIdx.y * sout;
    for (int64_t m = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; m < out_len;
         m += (int64_t)gridDim.x * blockDim.x) {
        double v = 0.0;
        if (m < M) {
            const int64_t R = tri_row(m), Cc = m - R * (R + 1) / 2;
            v = h2[R * n2 + Cc];
            if (R == Cc) v *= mult;
        }
        out[m] = v;
    }
}

__global__ void unpack_kernel(const double *__restrict__ p, int64_t sp, int n, double *__restrict__ out,
                              int64_t sout) {
    const int64_t n2 = (int64_t)n * n, n4 = n2 * n2;
    p += (int64_t)blockIdx.y * sp;
    out += (int64_t)blockIdx.y * sout;
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < n4;
         idx += (int64_t)gridDim.x * blockDim.x) {
        const int64_t R = idx / n2, Cc = idx - R * n2;
        out[idx] = R >= Cc ? p[tri_index(R, Cc)] : p[tri_index(Cc, R)];
    }
}

// 8-fold compressed form (EVC_LAYOUT_SYM8) of a tensor with the symmetries of real two-electron integrals
__global__ void pack_sym8_kernel(const double *__restrict__ h2, int64_t sh2, int n, double mult,
                                 double *__restrict__ out, int64_t sout, int64_t M, int64_t out_len) {
    const int64_t n2 = (int64_t)n * n;
    h2 += (int64_t)blockIdx.y * sh2;
    out += (int64_t)blockIdx.y * sout;
    for (int64_t m = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; m < out_len;
         m += (int64_t)gridDim.x * blockDim.x) {
        double v = 0.0;
        if (m < M) {
            const int64_t u = tri_row(m), w = m - u * (u + 1) / 2;
            const int64_t i = tri_row(u), j = u - i * (i + 1) / 2;
            const int64_t k = tri_row(w), l = w - k * (k + 1) / 2;
            v = h2[(i * n + j) * n2 + k * n + l] *
                ((u == w ? mult : 1.0) * (i != j ? 2.0 : 1.0) * (k != l ? 2.0 : 1.0));
        }
        out[m] = v;
    }
}

static unsigned grid_for(int64_t work, int block) {
    int64_t g = (work + block - 1) / block;
    if (g > 8192) g = 8192;
    if (g < 1) g = 1;
    return (unsigned)g;
}

int launch_pack(const double *h2, int64_t sh2, int n, double mult, double *out, int64_t sout, int64_t out_len,
                int count, hipStream_t st) {
    const int64_t n2 = (int64_t)n * n, M = n2 * (n2 + 1) / 2;
    hipLaunchKernelGGL(pack_kernel, dim3(grid_for(out_len, 256), (unsigned)count), dim3(256), 0, st, h2, sh2, n, mult,
                       out, sout, M, out_len);
    EVC_LAUNCH_CHECK("pack_pair_sym");
    return 0;
}

int launch_pack_sym8(const double *h2, int64_t sh2, int n, double mult, double *out, int64_t sout, int64_t out_len,
                     int count, hipStream_t st) {
    const int64_t mm = (int64_t)n * (n + 1) / 2, M = mm * (mm + 1) / 2;
    hipLaunchKernelGGL(pack_sym8_kernel, dim3(grid_for(out_len, 256), (unsigned)count), dim3(256), 0, st, h2, sh2, n,
                       mult, out, sout, M, out_len);
    EVC_LAUNCH_CHECK("pack_sym8");
    return 0;
}

int launch_unpack(const double *p, int64_t sp, int n, double *out, int64_t sout, int count, hipStream_t st) {
    const int64_t n4 = (int64_t)n * n * n * n;
    hipLaunchKernelGGL(unpack_kernel, dim3(grid_for(n4, 256), (unsigned)count), dim3(256), 0, st, p, sp, n, out, sout);
    EVC_LAUNCH_CHECK("unpack_pair_sym");
    return 0;
}

// ------------------------------------------------------------------ OAO symmetrisation (transposed)
// GsT[(j,k,l)][i] = G[i,j,k,l] + G[j,i,k,l] + G[l,k,j,i] + G[k,l,i,j]
__global__ void sym_oao_t_kernel(const double *__restrict__ G, int64_t sG, int n, double *__restrict__ out,
                                 int64_t sout) {
    const int64_t n2 = (int64_t)n * n, n3 = n2 * n, n4 = n2 * n2;
    G += (int64_t)blockIdx.y * sG;
    out += (int64_t)blockIdx.y * sout;
    for (int64_t o = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; o < n4;
         o += (int64_t)gridDim.x * blockDim.x) {
        const int i = (int)(o % n);
        int64_t r = o / n;
        const int l = (int)(r % n);
        r /= n;
        const int k = (int)(r % n);
        const int j = (int)(r / n);
        out[o] = G[i * n3 + j * n2 + k * n + l] + G[j * n3 + i * n2 + k * n + l] +
                 G[l * n3 + k * n2 + j * n + i] + G[k * n3 + l * n2 + i * n + j];
    }
}

int launch_sym_oao_t(const double *G, int64_t sG, int n, double *out, int64_t sout, int count, hipStream_t st) {
    const int64_t n4 = (int64_t)n * n * n * n;
    hipLaunchKernelGGL(sym_oao_t_kernel, dim3(grid_for(n4, 256), (unsigned)count), dim3(256), 0, st, G, sG, n, out,
                       sout);
    EVC_LAUNCH_CHECK("sym_oao_t");
    return 0;
}

// ------------------------------------------------------------------ packed fast path: unpack + both symmetrisations
// One workgroup per (j,k); its n*n elements (i,l) are produced with l fastest (coalesced SB/G rows),
// staged in LDS and written to GsT with i fastest ((j,k) fixes a contiguous n*n block of GsT).
__global__ __launch_bounds__(256) void unpack_sym_kernel(const double *__restrict__ p, int64_t sp, int n,
                                                         double *__restrict__ GsT, double *__restrict__ SB,
                                                         int64_t sws, double *__restrict__ Gout, int64_t sG,
                                                         int count) {
    extern __shared__ __align__(16) double tile[];  // [l][i], row length n+1
    const int64_t n2 = (int64_t)n * n, n3 = n2 * n;
    // XCD-aware order: workgroups are dealt round-robin to the 8 XCDs; the blocks an XCD receives work
    // through ONE geometry at a time, so that geometry's packed vector (3.2 MB at N=30) stays in the
    // XCD's 4 MB L2 for the three gathers per element.
    // The first count - count%8 geometries are laid out that way; the remainder (and any batch of fewer
    // than 8) is spread over all XCDs in plain (geometry, jk) order.
    const int nx = count & ~7;
    const int64_t nxblocks = (int64_t)nx * n * n;
    int geom, jk;
    if ((int64_t)blockIdx.x < nxblocks) {
        const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
        geom = (slot / (n * n)) * 8 + xcd;
        jk = slot % (n * n);
    } else {
        const int64_t b = (int64_t)blockIdx.x - nxblocks;
        geom = nx + (int)(b / (n * n));
        jk = (int)(b % (n * n));
    }
    p += (int64_t)geom * sp;
    GsT += (int64_t)geom * sws;
    SB += (int64_t)geom * sws;
    if (Gout) Gout += (int64_t)geom * sG;
    const int j = jk / n, k = jk - j * n;
    auto P = [&](int64_t a, int64_t b) { return a >= b ? p[tri_index(a, b)] : p[tri_index(b, a)]; };
    for (int idx = threadIdx.x; idx < n * n; idx += 256) {
        const int i = idx / n, l = idx - i * n;
        const int64_t o = i * n3 + j * n2 + k * n + l;
        const int64_t R = (int64_t)i * n + j, Rt = (int64_t)j * n + i, Cc = (int64_t)k * n + l, Ct = (int64_t)l * n + k;
        const double p1 = P(R, Cc), p2 = P(Rt, Cc), p3 = P(Rt, Ct);
        SB[o] = 2.0 * (p1 + p3);
        if (Gout) Gout[o] = p1;
        tile[l * (n + 1) + i] = 2.0 * p1 + p2 + p3;
    }
    lds_barrier();
    double *dst = GsT + ((int64_t)j * n + k) * n2;
    for (int idx = threadIdx.x; idx < n * n; idx += 256) {
        const int l = idx / n, i = idx - l * n;
        dst[idx] = tile[l * (n + 1) + i];
    }
}

int launch_unpack_sym(const double *packed, int64_t sp, int n, double *GsT, double *SB, int64_t sws, double *G,
                      int64_t sG, int count, hipStream_t st) {
    const size_t lds = sizeof(double) * (size_t)n * (n + 1);
    hipLaunchKernelGGL(unpack_sym_kernel, dim3((unsigned)(n * n * count)), dim3(256), lds, st, packed, sp, n, GsT, SB,
                       sws, G, sG, count);
    EVC_LAUNCH_CHECK("unpack_sym");
    return 0;
}

// 8-fold compressed vector p8 of a fully symmetric 2-RDM (EVC_LAYOUT_SYM8) -> SB[i][j][k][l] = 4 p8(ijkl), the
// operand of both the Y2 contraction and the OAO->AO rotation (every image of (i,j,k,l) is the same element, so
// the two symmetrisations of the general path coincide), and optionally G = p8(ijkl).  One workgroup per (i,j)
// writes a contiguous n*n block; same XCD-aware geometry order as above (0.87 MB per geometry at N = 30).
// lead_half: SB is only needed for i >= j and l <= k (its consumers fold both symmetries).
__global__ __launch_bounds__(256) void unpack8_kernel(const double *__restrict__ p, int64_t sp, int n,
                                                      double *__restrict__ SB, int64_t sws,
                                                      double *__restrict__ Gout, int64_t sG, int count,
                                                      int lead_half) {
    const int64_t n2 = (int64_t)n * n;
    const int nx = count & ~7;
    const int64_t nxblocks = (int64_t)nx * n * n;
    int geom, ij;
    if ((int64_t)blockIdx.x < nxblocks) {
        const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
        geom = (slot / (n * n)) * 8 + xcd;
        ij = slot % (n * n);
    } else {
        const int64_t b = (int64_t)blockIdx.x - nxblocks;
        geom = nx + (int)(b / (n * n));
        ij = (int)(b % (n * n));
    }
    const int i = ij / n, j = ij - i * n;
    const bool want_sb = !(lead_half && i < j);
    if (!want_sb && !Gout) return;
    p += (int64_t)geom * sp;
    double *sb = SB + (int64_t)geom * sws + (int64_t)ij * n2;
    double *go = Gout ? Gout + (int64_t)geom * sG + (int64_t)ij * n2 : nullptr;
    const int64_t u = i >= j ? tri_index(i, j) : tri_index(j, i);
    for (int idx = threadIdx.x; idx < n * n; idx += 256) {
        const int k = idx / n, l = idx - k * n;
        const int64_t v = k >= l ? tri_index(k, l) : tri_index(l, k);
        const double val = u >= v ? p[tri_index(u, v)] : p[tri_index(v, u)];
        if (want_sb && !(lead_half && l > k)) sb[idx] = 4.0 * val;
        if (go) go[idx] = val;
    }
}

// The same for lead_half without G: only the quarter i >= j, l <= k of SB is written.  One WAVE per pair (i,j),
// its lanes run over v = tri(k,l) (the order of the compressed vector: the gather p8[tri(u,v)] is contiguous for
// v <= u); 4 pairs per workgroup instead of one workgroup per (i,j) with half of them idle.
__global__ __launch_bounds__(256) void unpack8_half_kernel(const double *__restrict__ p, int64_t sp, int n,
                                                           double *__restrict__ SB, int64_t sws, int count) {
    const int64_t n2 = (int64_t)n * n;
    const int npairs = n * (n + 1) / 2;
    const int bpg = (npairs + 3) / 4;   // workgroups per geometry
    const int nx = count & ~7;
    const int64_t nxblocks = (int64_t)nx * bpg;
    int geom, blk;
    if ((int64_t)blockIdx.x < nxblocks) {
        const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
        geom = (slot / bpg) * 8 + xcd;
        blk = slot % bpg;
    } else {
        const int64_t b = (int64_t)blockIdx.x - nxblocks;
        geom = nx + (int)(b / bpg);
        blk = (int)(b % bpg);
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int u = blk * 4 + wave;
    if (u >= npairs) return;
    const int i = (int)tri_row(u), j = u - i * (i + 1) / 2;
    p += (int64_t)geom * sp;
    double *sb = SB + (int64_t)geom * sws + ((int64_t)i * n + j) * n2;
    for (int v = lane; v < npairs; v += 64) {
        const int k = (int)tri_row(v), l = v - k * (k + 1) / 2;
        const double val = u >= v ? p[tri_index(u, v)] : p[tri_index(v, u)];
        sb[k * n + l] = 4.0 * val;
    }
}

// Dense (pair, pair) form of the same: SB[u][v] = 4 p8[tri(max(u,v), min(u,v))], u = tri(i,j), v = tri(k,l) -- the
// symmetric matrix the compressed vector is the lower triangle of.  One wave per row u.
__global__ __launch_bounds__(256) void unpack8_pairs_kernel(const double *__restrict__ p, int64_t sp, int n,
                                                            double *__restrict__ SB, int64_t sws, int count, int ld) {
    const int npairs = n * (n + 1) / 2;
    const int bpg = (npairs + 3) / 4;   // workgroups per geometry
    const int nx = count & ~7;
    const int64_t nxblocks = (int64_t)nx * bpg;
    int geom, blk;
    if ((int64_t)blockIdx.x < nxblocks) {
        const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
        geom = (slot / bpg) * 8 + xcd;
        blk = slot % bpg;
    } else {
        const int64_t b = (int64_t)blockIdx.x - nxblocks;
        geom = nx + (int)(b / bpg);
        blk = (int)(b % bpg);
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int u = blk * 4 + wave;
    if (u >= npairs) return;
    p += (int64_t)geom * sp;
    double *sb = SB + (int64_t)geom * sws + (int64_t)u * ld;   // (rows at the pitch pair_ld(n) of the pipeline's dense forms)
    for (int v = lane; v < npairs; v += 64) sb[v] = 4.0 * (u >= v ? p[tri_index(u, v)] : p[tri_index(v, u)]);
}

int launch_unpack8(const double *packed, int64_t sp, int n, double *SB, int64_t sws, double *G, int64_t sG, int count,
                   int lead_half, hipStream_t st) {
    if (lead_half == 2 && !G) {
        const int bpg = (n * (n + 1) / 2 + 3) / 4;
        hipLaunchKernelGGL(unpack8_pairs_kernel, dim3((unsigned)(bpg * count)), dim3(256), 0, st, packed, sp, n, SB,
                           sws, count, pair_ld(n));
        EVC_LAUNCH_CHECK("unpack8_pairs");
        return 0;
    }
    if (lead_half && !G) {
        const int bpg = (n * (n + 1) / 2 + 3) / 4;
        hipLaunchKernelGGL(unpack8_half_kernel, dim3((unsigned)(bpg * count)), dim3(256), 0, st, packed, sp, n, SB, sws,
                           count);
        EVC_LAUNCH_CHECK("unpack8_half");
        return 0;
    }
    hipLaunchKernelGGL(unpack8_kernel, dim3((unsigned)(n * n * count)), dim3(256), 0, st, packed, sp, n, SB, sws, G, sG,
                       count, lead_half);
    EVC_LAUNCH_CHECK("unpack8");
    return 0;
}

// ------------------------------------------------------------------ Y2 contraction (split-K MFMA GEMM)
// partial[slab][i][a] = sum_{k in slab} GsT[k][i] * K3[k][a],  k = (j,k,l) flattened, n^3 long.
}  // namespace evc
