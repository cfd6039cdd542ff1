// K15: int2e_ip1 diagonal contraction, dhcore dots and the Y2 slab sums (gradients_loewdin.py:234-252).
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
// Remaining blocks: y2[e] = sum_slab y2part[slab][e].
// elements of the (b,c,d) range per thread (sizes the t2part workspace)
static int ip1_per_thread() {
    static const int pt = [] {
        return 8;
    }();
    return pt;
}
int ip1_chunks(int n) {
    // (at least n: the pair-block form of the packed contraction files its partials under the partner index b)
    const int64_t n3 = (int64_t)n * n * n;
    const int c = (int)ceil_div(n3, 256 * ip1_per_thread());
    return c > n ? c : n;
}

template <int kIp1PerThread>
__global__ __launch_bounds__(256) void ip1_dh_kernel(Ip1Args a) {
    __shared__ double scr[3][4];
    __shared__ double part[4][64];
    const int n = a.n, nchunk = a.nchunk;
    const int64_t n2 = (int64_t)n * n, n3 = n2 * n, n4 = n2 * n2;
    const int64_t g = blockIdx.y;
    const bool pair_blocks = a.presym && a.fold_cd && a.ip1_s2kl;
    const int nb1 = pair_blocks ? n * (n + 1) / 2 : n * nchunk;
    if (pair_blocks && (int)blockIdx.x < nb1) {
        // int2e_ip1 packed in (c,d), c >= d, against the dense (pair, pair) AO-basis 2-RDM G[tri(m,b)][v] (rows at the
        // pitch pair_ld(n); the weight 2 of c != d is applied here): one block per unordered pair {m, b} -- the row G[tri(hi,lo)][:]
        // is read once and contracted with ip1[x][hi][lo][:] (-> t2[x][hi], filed under partner lo) and, for
        // hi != lo, with ip1[x][lo][hi][:] (-> t2[x][lo], partner hi): 7 contiguous streams of n(n+1)/2 doubles
        const double *__restrict__ ip1 = a.ip1 + g * a.sip1;
        const double *__restrict__ G = a.Gao + g * a.sws;
        const int npr = n * (n + 1) / 2;
        const int pidx = blockIdx.x, hi = tri_row_small(pidx), lo = pidx - hi * (hi + 1) / 2;
        const int64_t len = (int64_t)n * npr;          // one (x, m) block of ip1
        const double *__restrict__ gr = G + (int64_t)pidx * pair_ld(n);
        const double *__restrict__ qh = ip1 + ((int64_t)hi * n + lo) * npr;
        const double *__restrict__ ql = ip1 + ((int64_t)lo * n + hi) * npr;
        const bool both = hi != lo;
        double ah[3] = {0.0, 0.0, 0.0}, al[3] = {0.0, 0.0, 0.0};
        // two adjacent elements per lane: seven 16-byte loads in flight per lane (the int2e_ip1 rows of the caller's s2kl
        // array start on any multiple of 8 bytes: loads typed with 8-byte alignment)
        typedef double d2u __attribute__((ext_vector_type(2), aligned(8)));
        for (int v = 2 * threadIdx.x; v < npr; v += 512) {
            const int vc = tri_row_small(v), vc1 = tri_row_small(v + 1);
            // multiplicity of the pair (c,d), c >= d: 1 on the diagonal, else 2
            const double w0 = v == vc * (vc + 3) / 2 ? 1.0 : 2.0, w1 = (v + 1) == vc1 * (vc1 + 3) / 2 ? 1.0 : 2.0;
            if (v + 1 < npr) {
                const d2u gg = *reinterpret_cast<const d2u *>(gr + v);
                const double g0 = gg[0] * w0, g1 = gg[1] * w1;
#pragma unroll
                for (int x = 0; x < 3; ++x) {
                    const d2u q = *reinterpret_cast<const d2u *>(qh + (int64_t)x * n * len + v);
                    ah[x] = fma(q[1], g1, fma(q[0], g0, ah[x]));
                    if (both) {
                        const d2u r = *reinterpret_cast<const d2u *>(ql + (int64_t)x * n * len + v);
                        al[x] = fma(r[1], g1, fma(r[0], g0, al[x]));
                    }
                }
            } else {
                const double g0 = gr[v] * w0;
#pragma unroll
                for (int x = 0; x < 3; ++x) {
                    ah[x] = fma(qh[(int64_t)x * n * len + v], g0, ah[x]);
                    if (both) al[x] = fma(ql[(int64_t)x * n * len + v], g0, al[x]);
                }
            }
        }
        const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
        __shared__ double pr6[6][4];
#pragma unroll
        for (int x = 0; x < 3; ++x) {
            const double sh = wave_sum(ah[x]), sl = wave_sum(al[x]);
            if (lane == 0) {
                pr6[x][wave] = sh;
                pr6[3 + x][wave] = sl;
            }
        }
        __syncthreads();
        if (threadIdx.x < 6) {
            const int x = threadIdx.x % 3, role = threadIdx.x / 3;
            const double t = (pr6[threadIdx.x][0] + pr6[threadIdx.x][1]) + (pr6[threadIdx.x][2] + pr6[threadIdx.x][3]);
            double *tp = a.t2part + g * a.sws;
            if (role == 0) tp[((int64_t)hi * 3 + x) * nchunk + lo] = t;
            else if (both) tp[((int64_t)lo * 3 + x) * nchunk + hi] = t;
        }
        if (lo == 0 && nchunk > n) {   // slots behind the n partners (only if the chunk count exceeds n)
            double *tp = a.t2part + g * a.sws;
            for (int idx = threadIdx.x; idx < 3 * (nchunk - n); idx += 256)
                tp[((int64_t)hi * 3 + idx / (nchunk - n)) * nchunk + n + idx % (nchunk - n)] = 0.0;
        }
    } else if ((int)blockIdx.x < nb1) {
        const double *__restrict__ ip1 = a.ip1 + g * a.sip1;
        const double *__restrict__ G = a.Gao + g * a.sws;
        const int m = blockIdx.x / nchunk, ch = blockIdx.x % nchunk;
        double a0 = 0.0, a1 = 0.0, a2 = 0.0;
        const int64_t e0 = (int64_t)ch * 256 * kIp1PerThread;
        if (a.presym && a.fold_cd && (n & 1) == 0) {
            // symmetrised operand that is only valid for d <= c (and symmetric in c <-> d, like ip1 itself): the
            // dot runs over the lower triangles with weight 2 off the diagonal.  The 16-byte pairs (d, d+1), d even,
            // d <= c, of one b are numbered row by row (rows 2h and 2h+1 hold h+1 pairs each, h(h+1) pairs precede
            // row 2h), so every lane of the chunk has a live pair; n is even here.
            const int hp = n / 2, ppb = hp * (hp + 1);
            const int64_t npairs = (int64_t)n * ppb;
            const int64_t per = (npairs + nchunk - 1) / nchunk;
            const int64_t pe = min(npairs, (int64_t)(ch + 1) * per);
            for (int64_t ep = (int64_t)ch * per + threadIdx.x; ep < pe; ep += 256) {
                const int b = (int)(ep / ppb), t = (int)(ep - (int64_t)b * ppb);
                int h = (int)sqrt((double)t);
                while (h * (h + 1) > t) --h;
                while ((h + 1) * (h + 2) <= t) ++h;
                const int tp = t - h * (h + 1);
                const int up = tp >= h + 1 ? 1 : 0;
                const int c = 2 * h + up, d = 2 * (tp - up * (h + 1));
                const int64_t off = m * n3 + (int64_t)b * n2 + c * n + d;
                // the operand is also symmetric in m <-> b and only stored for b <= m
                const int64_t goff = b <= m ? off : (int64_t)b * n3 + (int64_t)m * n2 + c * n + d;
                const double2 gr = *reinterpret_cast<const double2 *>(G + goff);
                const double2 p0 = *reinterpret_cast<const double2 *>(ip1 + off);
                const double2 p1 = *reinterpret_cast<const double2 *>(ip1 + n4 + off);
                const double2 p2 = *reinterpret_cast<const double2 *>(ip1 + 2 * n4 + off);
                const double gx = d < c ? 2.0 * gr.x : gr.x;
                const double gy = d + 1 < c ? 2.0 * gr.y : (d + 1 == c ? gr.y : 0.0);
                a0 = fma(p0.y, gy, fma(p0.x, gx, a0));
                a1 = fma(p1.y, gy, fma(p1.x, gx, a1));
                a2 = fma(p2.y, gy, fma(p2.x, gx, a2));
            }
        } else if (a.presym && (n3 & 1) == 0) {
            // symmetrised operand: a plain streaming dot, 16-byte loads
#pragma unroll
            for (int u = 0; u < kIp1PerThread / 2; ++u) {
                const int64_t e = e0 + ((int64_t)u * 256 + threadIdx.x) * 2;
                if (e < n3) {
                    const int64_t off = m * n3 + e;
                    const double2 gs = *reinterpret_cast<const double2 *>(G + off);
                    const double2 p0 = *reinterpret_cast<const double2 *>(ip1 + off);
                    const double2 p1 = *reinterpret_cast<const double2 *>(ip1 + n4 + off);
                    const double2 p2 = *reinterpret_cast<const double2 *>(ip1 + 2 * n4 + off);
                    a0 = fma(p0.y, gs.y, fma(p0.x, gs.x, a0));
                    a1 = fma(p1.y, gs.y, fma(p1.x, gs.x, a1));
                    a2 = fma(p2.y, gs.y, fma(p2.x, gs.x, a2));
                }
            }
        } else
#pragma unroll
        for (int u = 0; u < kIp1PerThread; ++u) {
            const int64_t e = e0 + u * 256 + threadIdx.x;
            if (e < n3) {
                const int d = (int)(e % n);
                const int c = (int)((e / n) % n);
                const int b = (int)(e / n2);
                if (a.fold_cd && d > c) continue;
                double gs = a.presym ? (a.fold_cd && b > m ? G[(int64_t)b * n3 + (int64_t)m * n2 + c * n + d] : G[m * n3 + e])
                                     : G[m * n3 + e] + G[b * n3 + m * n2 + d * n + c] +
                                           G[c * n3 + d * n2 + m * n + b] + G[d * n3 + c * n2 + b * n + m];
                if (a.fold_cd && d < c) gs *= 2.0;
                const int64_t off = m * n3 + e;
                a0 = fma(ip1[off], gs, a0);
                a1 = fma(ip1[n4 + off], gs, a1);
                a2 = fma(ip1[2 * n4 + off], gs, a2);
            }
        }
        a0 = wave_sum(a0);
        a1 = wave_sum(a1);
        a2 = wave_sum(a2);
        const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
        if (lane == 0) {
            scr[0][wave] = a0;
            scr[1][wave] = a1;
            scr[2][wave] = a2;
        }
        __syncthreads();
        if (threadIdx.x < 3) {
            const int x = threadIdx.x;
            a.t2part[g * a.sws + ((int64_t)m * 3 + x) * nchunk + ch] =
                (scr[x][0] + scr[x][1]) + (scr[x][2] + scr[x][3]);
        }
    } else if ((int)blockIdx.x < nb1 + a.natm * 3) {
        const int ax = blockIdx.x - nb1;  // A*3 + x
        const double *p = a.dh + g * a.sdh + (int64_t)ax * n2;
        const double *Pao = a.Pao + g * a.sws;
        double s = 0.0;
        for (int64_t e = threadIdx.x; e < n2; e += 256) s = fma(p[e], Pao[e], s);
        s = block_sum<4>(s, &scr[0][0]);
        if (threadIdx.x == 0) a.term3[g * a.sws + ax] = s;
    } else {
        // 64 elements per block, 4 slab groups per element
        const int b = blockIdx.x - nb1 - a.natm * 3;
        const int e = b * 64 + (threadIdx.x & 63), grp = threadIdx.x >> 6;
        const double *y2part = a.y2part + g * a.sws;
        // (four loads in flight per thread: with one geometry per call the Y2 kernels leave > 100 slabs)
        double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
        if (e < n2) {
            int sl = grp;
            for (; sl + 12 < a.nslab; sl += 16) {
                s0 += y2part[(int64_t)sl * n2 + e];
                s1 += y2part[(int64_t)(sl + 4) * n2 + e];
                s2 += y2part[(int64_t)(sl + 8) * n2 + e];
                s3 += y2part[(int64_t)(sl + 12) * n2 + e];
            }
            for (; sl < a.nslab; sl += 4) s0 += y2part[(int64_t)sl * n2 + e];
        }
        part[grp][threadIdx.x & 63] = (s0 + s1) + (s2 + s3);
        __syncthreads();
        if (grp == 0 && e < n2)
            a.y2[g * a.sws + e] = (part[0][threadIdx.x] + part[1][threadIdx.x]) +
                                  (part[2][threadIdx.x] + part[3][threadIdx.x]);
    }
}

int launch_ip1_dh(const Ip1Args &a, int count, hipStream_t st) {
    const bool pair_blocks = a.presym && a.fold_cd && a.ip1_s2kl;
    const int blocks = (pair_blocks ? a.n * (a.n + 1) / 2 : a.n * a.nchunk) + a.natm * 3 + (a.n * a.n + 63) / 64;
    switch (ip1_per_thread()) {
        case 16: hipLaunchKernelGGL(ip1_dh_kernel<16>, dim3(blocks, (unsigned)count), dim3(256), 0, st, a); break;
        case 8: hipLaunchKernelGGL(ip1_dh_kernel<8>, dim3(blocks, (unsigned)count), dim3(256), 0, st, a); break;
        default: hipLaunchKernelGGL(ip1_dh_kernel<4>, dim3(blocks, (unsigned)count), dim3(256), 0, st, a); break;
    }
    EVC_LAUNCH_CHECK("ip1_dh");
    return 0;
}

}  // namespace evc
