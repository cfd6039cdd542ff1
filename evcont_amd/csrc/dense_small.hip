// Single-workgroup dense kernels (everything O(N^3) or O(T^3); matrices live in LDS):
//   K1/K2  Loewdin orthogonalisation: parallel cyclic Jacobi, X = S^-1/2, h1 = X^T h X
//          (electron_integral_utils.py:6-18,135; gradients_loewdin.py:336-338)
//   K6     subspace generalised eigenproblem, LAPACK dsygvd semantics
//          (evcont.py:38-90,157-173) + pair weights (gradients_loewdin.py:343-353)
//   K12/K16 one-body gradient intermediates and the adjoint Loewdin response that folds
//          K10/K11 (gradients_loewdin.py:41-134,155-187,300-303) into four N^3 products.
// These kernels are latency bound (one CU): every global operand is staged into LDS with
// coalesced loads first, inner loops carry no integer division, and the 256 threads are
// used as a 16x16 grid (tj = row group, tk = column group).
#include <stdlib.h>

#include "common.hpp"
#include "kernels.hpp"

namespace evc {

constexpr int kThreads = 256;

// ------------------------------------------------------------------ small LDS matmul
// C[i][j] = sum_k a(i,k) * b(k,j),  i,j < n;  thread (tj,tk) owns i = tj+16*, j = tk+16*.
typedef double d4s __attribute__((ext_vector_type(4)));
// Small products on the FP64 matrix cores (n <= 64; up to 32: wave w of the workgroup owns the 16 x 16 output tile
// (w >> 1, w & 1), beyond: sixteen tiles round-robin); operand maps of v_mfma_f64_16x16x4_f64: A[i][k]: lane l holds i = l & 15, k = l >> 4; B[k][j]:
// k = l >> 4, j = l & 15; D[i][j]: j = l & 15, i = (l >> 4) + 4 reg.  One eighth of the LDS traffic of the 2 x 2
// register-blocked vector version (which was LDS-bandwidth bound: ~1 us per 32^3 product against ~0.3 us).
// EVC_SMALL_MM_VALU (build flag): the vector version everywhere.
#ifndef EVC_SMALL_MM_VALU
#define EVC_SMALL_MM_MFMA 1
#endif

// C[i][j] = sum_k a(i,k) b(k,j), i, j, k < n; operands through accessors, result through store(i, j, value).
template <typename FA, typename FB, typename FC>
__device__ __forceinline__ void mm16(int n, FA a, FB b, FC store) {
#ifdef EVC_SMALL_MM_MFMA
    if (n <= 32) {
        const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, l15 = lane & 15, l4 = lane >> 4;
        const int ti = wave >> 1, tj = wave & 1;
        if (16 * ti < n && 16 * tj < n) {   // wave-uniform
            const int i = 16 * ti + l15, j = 16 * tj + l15;
            const int ic = i < n ? i : 0, jc = j < n ? j : 0;   // (rows / columns beyond n: computed on row 0, never stored)
            d4s acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int kk = 0; kk < 8; ++kk)
                if (4 * kk < n) {
                    const int k = 4 * kk + l4;
                    const bool kv = k < n;
                    const int kc = kv ? k : 0;
                    const double av = a(ic, kc), bv = b(kc, jc);
                    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(kv ? av : 0.0, kv ? bv : 0.0, acc, 0, 0, 0);
                }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int ii = 16 * ti + l4 + 4 * r;
                if (ii < n && j < n) store(ii, j, acc[r]);
            }
        }
        return;
    }
    if (n <= 64) {   // up to sixteen tiles, dealt round-robin to the four waves, K up to 64
        const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, l15 = lane & 15, l4 = lane >> 4;
        const int nt = (n + 15) >> 4;
        for (int t = wave; t < nt * nt; t += kThreads / 64) {   // wave-uniform
            const int ti = t / nt, tj = t - ti * nt;
            const int i = 16 * ti + l15, j = 16 * tj + l15;
            const int ic = i < n ? i : 0, jc = j < n ? j : 0;
            d4s acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int kk = 0; kk < 16; ++kk)
                if (4 * kk < n) {
                    const int k = 4 * kk + l4;
                    const bool kv = k < n;
                    const int kc = kv ? k : 0;
                    const double av = a(ic, kc), bv = b(kc, jc);
                    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(kv ? av : 0.0, kv ? bv : 0.0, acc, 0, 0, 0);
                }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int ii = 16 * ti + l4 + 4 * r;
                if (ii < n && j < n) store(ii, j, acc[r]);
            }
        }
        return;
    }
#endif
    const int tk = threadIdx.x & 15, tj = threadIdx.x >> 4;
    for (int i0 = 0; i0 < n; i0 += 32)
        for (int j0 = 0; j0 < n; j0 += 32) {
            const int ia = i0 + tj, ib = i0 + tj + 16, ja = j0 + tk, jb = j0 + tk + 16;
            const bool via = ia < n, vib = ib < n, vja = ja < n, vjb = jb < n;
            const int ia_ = via ? ia : 0, ib_ = vib ? ib : 0, ja_ = vja ? ja : 0, jb_ = vjb ? jb : 0;
            double c00 = 0.0, c01 = 0.0, c10 = 0.0, c11 = 0.0;
            for (int k = 0; k < n; ++k) {
                const double a0 = a(ia_, k), a1 = a(ib_, k);
                const double b0 = b(k, ja_), b1 = b(k, jb_);
                c00 = fma(a0, b0, c00);
                c01 = fma(a0, b1, c01);
                c10 = fma(a1, b0, c10);
                c11 = fma(a1, b1, c11);
            }
            if (via && vja) store(ia, ja, c00);
            if (via && vjb) store(ia, jb, c01);
            if (vib && vja) store(ib, ja, c10);
            if (vib && vjb) store(ib, jb, c11);
        }
}

__device__ __forceinline__ void copy_to_lds(double *dst, const double *__restrict__ src, int count) {
    for (int idx = threadIdx.x; idx < count; idx += kThreads) dst[idx] = src[idx];
}

// ------------------------------------------------------------------ Jacobi eigensolver (LDS)
// A (m x m, m even, symmetric, both triangles kept) is diagonalised in place; V accumulates the
// rotations (columns = eigenvectors).  Pairs follow the round-robin tournament, computed
// arithmetically: at step s pair k is (k ? (s+k) mod (m-1) : m-1, (s+m-1-k) mod (m-1)), so the m/2
// rotations of a step are disjoint.  A step is: (i) m/2 lanes compute (c,s); barrier; (ii) every 2x2
// block (pair k) x (pair k2), k<=k2, gets its row AND column rotation in registers and is mirrored;
// V gets the column rotation; barrier.  The kernel is a chain of ~7(m-1) such latency-bound steps,
// so the rotation uses the hardware rcp/rsq seeds with explicit Newton steps: the angle needs
// only ~1e-8 (it merely has to make a_pq small), while c is refined to full precision so that
// c^2+s^2 = 1 to rounding and V stays orthogonal.
__device__ __forceinline__ void pair_of(int step, int k, int m, int &p, int &q) {
    const int w = m - 1;
    p = step + k;
    if (p >= w) p -= w;
    if (k == 0) p = w;
    q = step + w - k;
    if (q >= w) q -= w;
}

__device__ __forceinline__ void jacobi_rotation(double app, double aqq, double apq, double &c, double &s) {
    c = 1.0;
    s = 0.0;
    if (fabs(apq) > 1.0e-150) {
        // t = sgn(d) b / (|d| + sqrt(d^2 + b^2)), d = aqq - app, b = 2 apq  (the smaller root)
        const double d = aqq - app, b = 2.0 * apq;
        const double h2 = fma(d, d, b * b);
        double y = __builtin_amdgcn_rsq(h2);
        y = y * fma(-0.5 * h2 * y, y, 1.5);
        const double den = fabs(d) + h2 * y;
        double r = __builtin_amdgcn_rcp(den);
        r = r * fma(-den, r, 2.0);
        const double t = copysign(b, d * b) * r;
        const double x = fma(t, t, 1.0);
        double z = __builtin_amdgcn_rsq(x);
        z = z * fma(-0.5 * x * z, z, 1.5);
        z = z * fma(-0.5 * x * z, z, 1.5);
        z = z * fma(-0.5 * x * z, z, 1.5);
        c = z;
        s = t * z;
    }
}

// init_v = false: V already holds an orthogonal matrix and A the matrix in THAT basis (warm start).
__device__ __forceinline__ void jacobi_eigh_lds(double *A, double *V, int m, double *rot, double *red, bool init_v = true) {
    const int tid = threadIdx.x;
    const int tk = tid & 15, tj = tid >> 4;
    const int half = m >> 1;
    if (init_v)
        for (int i = tj; i < m; i += 16)
            for (int j = tk; j < m; j += 16) V[i * m + j] = (i == j) ? 1.0 : 0.0;
    __syncthreads();
    for (int sweep = 0; sweep < 60; ++sweep) {
        double off = 0.0, dg = 0.0;
        for (int i = tj; i < m; i += 16)
            for (int j = tk; j < m; j += 16) {
                const double v = A[i * m + j];
                if (i == j) dg = fma(v, v, dg);
                else off = fma(v, v, off);
            }
        off = block_sum<4>(off, red);
        dg = block_sum<4>(dg, red + 4);
        if (!(off > 1.0e-32 * dg)) break;  // converged (or NaN input)
        for (int step = 0; step < m - 1; ++step) {
            if (tid < half) {
                int p, q;
                pair_of(step, tid, m, p, q);
                double c, s;
                jacobi_rotation(A[p * m + p], A[q * m + q], A[p * m + q], c, s);
                rot[2 * tid] = c;
                rot[2 * tid + 1] = s;
            }
            __syncthreads();
            for (int kb = 0; kb < half; kb += 16)
                for (int k2b = kb; k2b < half; k2b += 16) {
                    const int k = kb + tj, k2 = k2b + tk;
                    if (k2 < half && k <= k2) {
                        int p, q, p2, q2;
                        pair_of(step, k, m, p, q);
                        pair_of(step, k2, m, p2, q2);
                        const double c = rot[2 * k], s = rot[2 * k + 1], c2 = rot[2 * k2], s2 = rot[2 * k2 + 1];
                        const double b00 = A[p * m + p2], b01 = A[p * m + q2], b10 = A[q * m + p2],
                                     b11 = A[q * m + q2];
                        const double r00 = c * b00 - s * b10, r01 = c * b01 - s * b11;   // J^T B
                        const double r10 = s * b00 + c * b10, r11 = s * b01 + c * b11;
                        const double n00 = c2 * r00 - s2 * r01, n01 = s2 * r00 + c2 * r01;  // . J2
                        const double n10 = c2 * r10 - s2 * r11, n11 = s2 * r10 + c2 * r11;
                        if (k == k2) {
                            const double o = 0.5 * (n01 + n10);  // ~1e-8 |a_pq|: the angle is approximate
                            A[p * m + p] = n00; A[p * m + q] = o; A[q * m + p] = o; A[q * m + q] = n11;
                        } else {
                            A[p * m + p2] = n00; A[p * m + q2] = n01; A[q * m + p2] = n10; A[q * m + q2] = n11;
                            A[p2 * m + p] = n00; A[q2 * m + p] = n01; A[p2 * m + q] = n10; A[q2 * m + q] = n11;
                        }
                    }
                }
            for (int ib = 0; ib < m; ib += 16)
                for (int kb = 0; kb < half; kb += 16) {
                    const int i = ib + tj, k = kb + tk;
                    if (i < m && k < half) {
                        int p, q;
                        pair_of(step, k, m, p, q);
                        const double c = rot[2 * k], s = rot[2 * k + 1];
                        const double vp = V[i * m + p], vq = V[i * m + q];
                        V[i * m + p] = c * vp - s * vq;
                        V[i * m + q] = s * vp + c * vq;
                    }
                }
            __syncthreads();
        }
    }
    __syncthreads();
}

// ------------------------------------------------------------------ one-sided Jacobi on ONE wave (m <= 32)
// Hestenes' method on the columns of G = A V (V orthogonal, A symmetric positive definite): the plane rotation of a
// column pair (p,q) that makes g_p . g_q = 0 is applied to the two columns; at convergence the columns of G are
// orthogonal, g_j = lambda_j v_j.  Same round-robin pairing as above, but a step needs no workgroup barrier and no
// hand-over of rotation parameters: the four lanes of a pair read their two columns (8 rows each, 16-byte LDS loads),
// reduce the three dot products among themselves with DPP, compute (c,s) redundantly and write the rotated columns
// back; LDS operations of one wave execute in order, so the next step sees them.  ~1/3 of the two-sided step's
// latency.  Columns are stored [col][row] with pitch kJwPitch; rows >= m are zero.  Called by wave 0 only.
// Convergence: |g_p.g_q| <= 1e-9 |g_p||g_q| for every pair BEFORE the rotations of a sweep.
constexpr int kJwPitch = 34;
constexpr int kJwMax = 32;

// Sum over the four lanes of a quad, every lane gets the total (DPP moves, common.hpp).
template <int CTRL, typename T>
__device__ __forceinline__ T dpp_quad(T v) { return dpp_move<CTRL>(v); }
constexpr int kQuadXor1 = 0xB1;   // quad_perm [1,0,3,2]
constexpr int kQuadXor2 = 0x4E;   // quad_perm [2,3,0,1]
template <typename T>
__device__ __forceinline__ T quad_sum(T v) {
    v += dpp_quad<kQuadXor1>(v);
    v += dpp_quad<kQuadXor2>(v);
    return v;
}

// Timing experiments (tools/micro/loewdin_time.py; build with EVC_DEBUG_STAMPS=1): workgroup 0 stamps the phases of
// the eigen-kernels with the 100 MHz wall clock and a cap on the sweeps of the wave solvers can be set.  Compiled out
// of the product library.
#ifdef EVC_DEBUG_STAMPS
__device__ long long g_dbg_stamp[64];
__device__ double g_dbg_val[64];
#define EVC_STAMP(i_)                                                                   \
    do {                                                                                \
        if (blockIdx.x == 0 && threadIdx.x == 0) g_dbg_stamp[i_] = wall_clock64();      \
    } while (0)
#define EVC_DBGVAL(i_, v_)                                                              \
    do {                                                                                \
        if (blockIdx.x == 0 && threadIdx.x == 0) g_dbg_val[i_] = (double)(v_);          \
    } while (0)
#else
#define EVC_STAMP(i_) do { } while (0)
#define EVC_DBGVAL(i_, v_) do { } while (0)
#endif
#define EVC_FEW_STAMP(i_) EVC_STAMP(i_)
}  // namespace evc
namespace evc {
#include "few_roots.hpp"
}
namespace evc {
#ifdef EVC_DEBUG_STAMPS
__device__ int g_dbg_max_sweeps = 0;   // > 0: cap on the sweeps of the wave solvers (EVC_DBG_MAX_SWEEPS)
#else
constexpr int g_dbg_max_sweeps = 0;
#endif

__device__ __forceinline__ void jacobi_onesided_wave(double *Gc, int m) {
    const int lane = threadIdx.x & 63;
    const int k = lane >> 2, sub = lane & 3;
    const int half = m >> 1;
    const bool active = k < half;
    const int row0 = sub * 8;
    const int cap = g_dbg_max_sweeps > 0 ? g_dbg_max_sweeps : 40;
    for (int sweep = 0; sweep < cap; ++sweep) {
        bool bad = false;
        for (int step = 0; step < m - 1; ++step) {
            int p = 0, q = 1;
            if (active) pair_of(step, k, m, p, q);
            double *gp = Gc + p * kJwPitch + row0, *gq = Gc + q * kJwPitch + row0;
            double2 xg[4], yg[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                xg[u] = *reinterpret_cast<const double2 *>(gp + 2 * u);
                yg[u] = *reinterpret_cast<const double2 *>(gq + 2 * u);
            }
            double al = 0.0, be = 0.0, ga = 0.0;
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                al = fma(xg[u].x, xg[u].x, fma(xg[u].y, xg[u].y, al));
                be = fma(yg[u].x, yg[u].x, fma(yg[u].y, yg[u].y, be));
                ga = fma(xg[u].x, yg[u].x, fma(xg[u].y, yg[u].y, ga));
            }
            al = quad_sum(al);
            be = quad_sum(be);
            ga = quad_sum(ga);
            // rotate when the columns are not yet orthogonal to working precision
            const double ab = al * be, g2 = ga * ga;
            const bool rot = active && (g2 > 1.0e-30 * ab);
            bad = bad || (active && g2 > 1.0e-18 * ab);
            if (rot) {
                // t = sgn(d) b / (|d| + sqrt(d^2 + b^2)), d = beta - alpha, b = 2 gamma (the smaller root)
                // (the angle only has to make g_p . g_q small: one Newton step on the seeds; c is refined to full
                // precision so that c^2 + s^2 = 1 to rounding and the columns keep their norms)
                const double d = be - al, b = 2.0 * ga;
                const double h2 = fma(d, d, b * b);
                double y = __builtin_amdgcn_rsq(h2);
                y = y * fma(-0.5 * h2 * y, y, 1.5);
                const double den = fabs(d) + h2 * y;
                double r = __builtin_amdgcn_rcp(den);
                r = r * fma(-den, r, 2.0);
                const double t = copysign(b, d * b == 0.0 ? b : d * b) * r;
                const double x = fma(t, t, 1.0);
                double z = __builtin_amdgcn_rsq(x);
                z = z * fma(-0.5 * x * z, z, 1.5);
                z = z * fma(-0.5 * x * z, z, 1.5);
                z = z * fma(-0.5 * x * z, z, 1.5);
                const double c = z, s = t * z;
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    double2 a, b2;
                    a.x = c * xg[u].x - s * yg[u].x;
                    a.y = c * xg[u].y - s * yg[u].y;
                    b2.x = s * xg[u].x + c * yg[u].x;
                    b2.y = s * xg[u].y + c * yg[u].y;
                    *reinterpret_cast<double2 *>(gp + 2 * u) = a;
                    *reinterpret_cast<double2 *>(gq + 2 * u) = b2;
                }
            }
        }
        // every pair of this sweep was orthogonal to 1e-9 BEFORE its rotation: quadratic convergence leaves the
        // columns orthogonal to rounding after it, a confirming sweep is not needed
        if (__ballot(bad) == 0) break;
    }
}

// Eigen-decomposition of the symmetric m x m matrix A (LDS, both triangles) for m <= 32 through the wave kernel
// above: on return diag(A) holds the eigenvalues and V (row-major, V[i*m+j]) the eigenvectors as columns.
// shift: A + shift*I must be positive definite (0 for an overlap matrix): then the converged columns are
// g_j = lambda_j v_j with lambda_j = |g_j| > 0, so V = G diag(1/|g_j|) and no eigenvector matrix has to be carried
// through the rotations (its orthogonality is that of the columns of G, which is the convergence criterion).
// init_v = false: V holds an orthogonal start matrix and A the matrix in THAT basis (warm start), G0 = V (A + shift I).
// Gc: kJwMax x kJwPitch doubles of LDS.
__device__ __forceinline__ void jacobi_eigh_wave(double *A, double *V, int m, double shift, bool init_v, double *Gc, double *lam) {
    const int tid = threadIdx.x;
    for (int idx = tid; idx < kJwMax * kJwPitch; idx += kThreads) {
        const int j = idx / kJwPitch, i = idx - j * kJwPitch;
        double g = 0.0;
        if (i < m && j < m) {
            if (init_v) {
                g = A[i * m + j] + ((i == j) ? shift : 0.0);
            } else {
                double acc = shift * V[i * m + j];
                for (int kk = 0; kk < m; ++kk) acc = fma(V[i * m + kk], A[kk * m + j], acc);
                g = acc;
            }
        }
        Gc[idx] = g;
    }
    __syncthreads();
    if (tid < 64) jacobi_onesided_wave(Gc, m);
    __syncthreads();
    if (tid < m) {
        double nn = 0.0;
        for (int i = 0; i < m; ++i) nn = fma(Gc[tid * kJwPitch + i], Gc[tid * kJwPitch + i], nn);
        lam[tid] = sqrt(nn);
    }
    __syncthreads();
    for (int idx = tid; idx < m * m; idx += kThreads) {
        const int i = idx / m, j = idx - i * m;
        const double l = lam[j];
        // a zero column (the decoupled dummy dimension of an odd problem) keeps its unit vector
        V[idx] = l > 1.0e-300 ? Gc[j * kJwPitch + i] / l : (i == j ? 1.0 : 0.0);
        if (i == j) A[idx] = l - shift;
    }
    __syncthreads();
}

// ------------------------------------------------------------------ FP32 Jacobi + FP64 refinement (m <= 32)
// The FP64 wave Jacobi above is a chain of ~8 sweeps x (m-1) steps of ~1200 cycles each (f64 rsq/rcp seeds with
// Newton steps, twice the LDS bytes), and its last sweeps only square an error that is already tiny.  Two stages
// instead:
//  (1) the same one-sided Jacobi in FP32 (hardware v_sqrt/v_rcp/v_rsq_f32 without refinement, half the LDS traffic,
//      ~2.5x shorter steps), run until every column pair is orthogonal to 1e-5 before its rotation: eigenvectors to
//      ~1e-6;
//  (2) Ogita-Aishima refinement in FP64 (Japan J. Indust. Appl. Math. 35 (2018) 1007): with R = I - Z^T Z,
//      S = Z^T A Z, l_i = S_ii / (1 - R_ii),  E_ij = (S_ij + l_j R_ij) / (l_j - l_i)  (R_ij / 2 on the diagonal and
//      inside a cluster |l_i - l_j| <= delta),  Z <- Z + Z E  squares the error per pass: four small products on the
//      whole workgroup, one or two passes.
// Everything downstream (X = V f(s) V^T, the divided-difference response, the generalised eigenvectors) is a smooth
// function of invariant subspaces, so the arbitrary basis the cluster rule leaves inside a degenerate eigenspace is
// harmless -- as it is for LAPACK.  Returns false (A untouched) when the refinement does not contract: the caller
// falls back to the FP64 Jacobi.
constexpr int kJfPitch = 36;   // floats per column (16-byte aligned columns)

__device__ __forceinline__ void jacobi_onesided_wave_f32(float *Gf, int m) {
    const int lane = threadIdx.x & 63;
    const int k = lane >> 2, sub = lane & 3;
    const int half = m >> 1;
    const bool active = k < half;
    const int row0 = sub * 8;
    const int cap = g_dbg_max_sweeps > 0 ? g_dbg_max_sweeps : 30;
    for (int sweep = 0; sweep < cap; ++sweep) {
        bool bad = false;
        for (int step = 0; step < m - 1; ++step) {
            int p = 0, q = 1;
            if (active) pair_of(step, k, m, p, q);
            float *gp = Gf + p * kJfPitch + row0, *gq = Gf + q * kJfPitch + row0;
            float4 x0 = *reinterpret_cast<const float4 *>(gp), x1 = *reinterpret_cast<const float4 *>(gp + 4);
            float4 y0 = *reinterpret_cast<const float4 *>(gq), y1 = *reinterpret_cast<const float4 *>(gq + 4);
            float al = x0.x * x0.x, be = y0.x * y0.x, ga = x0.x * y0.x;
#define EVC_ACC(X_, Y_)          \
    al = fmaf(X_, X_, al);      \
    be = fmaf(Y_, Y_, be);      \
    ga = fmaf(X_, Y_, ga);
            EVC_ACC(x0.y, y0.y) EVC_ACC(x0.z, y0.z) EVC_ACC(x0.w, y0.w)
            EVC_ACC(x1.x, y1.x) EVC_ACC(x1.y, y1.y) EVC_ACC(x1.z, y1.z) EVC_ACC(x1.w, y1.w)
#undef EVC_ACC
            al = quad_sum(al);
            be = quad_sum(be);
            ga = quad_sum(ga);
            const float ab = al * be, g2 = ga * ga;
            const bool rot = active && (g2 > 1.0e-14f * ab);
            bad = bad || (active && g2 > 1.0e-10f * ab);
            if (rot) {
                const float d = be - al, b = 2.0f * ga;
                const float den = fabsf(d) + __builtin_sqrtf(fmaf(d, d, b * b));
                const float t = copysignf(b, d * b == 0.0f ? b : d * b) * __builtin_amdgcn_rcpf(den);
                const float c = __builtin_amdgcn_rsqf(fmaf(t, t, 1.0f)), s = t * c;
                float4 a0, a1, b0, b1;
#define EVC_ROT(F_)                         \
    a0.F_ = c * x0.F_ - s * y0.F_;          \
    b0.F_ = s * x0.F_ + c * y0.F_;          \
    a1.F_ = c * x1.F_ - s * y1.F_;          \
    b1.F_ = s * x1.F_ + c * y1.F_;
                EVC_ROT(x) EVC_ROT(y) EVC_ROT(z) EVC_ROT(w)
#undef EVC_ROT
                *reinterpret_cast<float4 *>(gp) = a0;
                *reinterpret_cast<float4 *>(gp + 4) = a1;
                *reinterpret_cast<float4 *>(gq) = b0;
                *reinterpret_cast<float4 *>(gq + 4) = b1;
            }
        }
        if (lane == 0) EVC_DBGVAL(20, sweep + 1);
        if (__ballot(bad) == 0) break;
    }
}

// max over the workgroup (NaN-propagating: a NaN input yields a NaN result); red: 4 doubles of LDS.
// Wave stage on DPP moves (quad xor 1, xor 2, row_half_mirror, row_mirror) + four readlanes, no LDS round trips.
__device__ __forceinline__ double nanmax(double a, double b) { return (a > b || a != a) ? a : b; }
__device__ __forceinline__ double wave_max_nan(double v) {
    v = nanmax(v, dpp_quad<kQuadXor1>(v));
    v = nanmax(v, dpp_quad<kQuadXor2>(v));
    v = nanmax(v, dpp_quad<0x141>(v));   // row_half_mirror
    v = nanmax(v, dpp_quad<0x140>(v));   // row_mirror: every lane holds the maximum of its row of 16
    return nanmax(nanmax(readlane_f64(v, 0), readlane_f64(v, 16)), nanmax(readlane_f64(v, 32), readlane_f64(v, 48)));
}
__device__ __forceinline__ double block_max_nan(double v, double *red) {
    v = wave_max_nan(v);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) red[wave] = v;
    __syncthreads();
    const double t = nanmax(nanmax(red[0], red[1]), nanmax(red[2], red[3]));
    __syncthreads();
    return t;
}
// two maxima with one pair of barriers (red: 8 doubles)
__device__ __forceinline__ void block_max_nan2(double &a, double &b, double *red) {
    a = wave_max_nan(a);
    b = wave_max_nan(b);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) {
        red[wave] = a;
        red[4 + wave] = b;
    }
    __syncthreads();
    a = nanmax(nanmax(red[0], red[1]), nanmax(red[2], red[3]));
    b = nanmax(nanmax(red[4], red[5]), nanmax(red[6], red[7]));
    __syncthreads();
}

// ------------------------------------------------------------------ FP32 tridiagonal eigensolver on ONE wave
// Start vectors for the refinement below, as LAPACK's xSYEVX would compute them, in single precision: Householder
// tridiagonalisation, eigenvalues by multisection on the Sturm count, eigenvectors of the tridiagonal matrix by
// twisted factorisation, back-transformation with the reflectors.  ~20 us for m = 30 where the Jacobi sweeps take
// 80.  Nothing here has to be accurate (the refinement squares the error and checks itself; vectors that come out
// parallel -- eigenvalues closer than single precision resolves -- make it give up, and the caller falls back to the
// Jacobi start), so there is no reorthogonalisation inside clusters and no safeguard beyond keeping pivots finite.
// Lane map: j = lane & 31 (row of the matrix, eigenvalue, eigenvector), h = lane >> 5 (column half / direction).
#ifndef EVC_STURM_ROUNDS
#define EVC_STURM_ROUNDS 6
#endif
constexpr int kSturmRounds = EVC_STURM_ROUNDS;   // multisection rounds of 17 sub-intervals each: 17^6 = 2.4e7 ~ 1 / FP32 epsilon
                                                 // (5 rounds: start error 1.5e-3 instead of 1.3e-4 at N = 30, a third refinement pass, +4 us)
constexpr int kTp = 36;    // floats per row of the matrix being reduced (16-byte aligned rows)
constexpr int kZfp = 33;   // floats per eigenvector row of the result (lane-private rows, conflict-free)

__device__ __forceinline__ float half32_sum(float v) {   // sum over the 32 lanes j (both halves hold the same values)
    v += dpp_quad<kQuadXor1>(v);
    v += dpp_quad<kQuadXor2>(v);
    v += dpp_quad<0x141>(v);
    v += dpp_quad<0x140>(v);
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 0)) +
           __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 16));
}
__device__ __forceinline__ float readlane_f32(float v, int l) {
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), l));
}

// Af: m x m matrix (pitch kTp, rows/columns >= m zero), destroyed.  Zf[j*kZfp + i] = component i of eigenvector j
// (unnormalised), zn[j] = 1 / its norm.  scr: 32*32*5 + 5*32 floats; cntbuf: [2][8][32] ints (Sturm counts of a
// multisection round).  Called by the whole workgroup (it
// contains barriers): the reduction and the eigenvectors are chains on wave 0, the multisection runs one abscissa
// per lane on all four waves.
__device__ __forceinline__ void tridiag_eig_wg_f32(float *Af, int m, float *Zf, float *zn, float *scr, int *cntbuf) {
    const int lane = threadIdx.x & 63, j = lane & 31, h = lane >> 5, wave = threadIdx.x >> 6;
    float *Vh = scr;                    // [k][r] reflector k
    float *fD = Vh + 32 * 32;           // [i][j][h] pivots of the forward / backward factorisation of lane pair j
    float *fF = fD + 2 * 32 * 32;       // [i][j][h] multipliers
    float *vv = fF + 2 * 32 * 32;       // v of the current step
    float *ww = vv + 32;                // w
    float *dd = ww + 32, *ee = dd + 32, *bb = ee + 32;   // diagonal, off-diagonal, 2 / v^T v
    const int c0 = h * 16;
    // ---- Householder reduction: A <- H_k A H_k, H_k = I - beta v v^T, v zero up to row k; the matrix stays in LDS, the
    //      two half-waves split the columns (a register-resident variant, few_roots.hpp, measured 0.97 us per step in
    //      single precision against 0.69 us for this loop: both are chains of reductions and LDS round trips, and this
    //      one has the shorter sums)
    for (int k = 0; wave == 0 && k + 2 < m; ++k) {
        const float x = (j > k && j < m) ? Af[j * kTp + k] : 0.0f;
        const float sig = half32_sum(x * x);
        const float xk1 = readlane_f32(x, k + 1);
        const float rest = sig - xk1 * xk1;            // what the reflector has to remove
        float alpha = xk1, beta = 0.0f, v = 0.0f;
        if (rest > 1.0e-30f) {
            alpha = -copysignf(__builtin_sqrtf(sig), xk1);
            beta = __builtin_amdgcn_rcpf(sig - xk1 * alpha);
            v = (j == k + 1) ? xk1 - alpha : x;
        }
        if (h == 0) {
            vv[j] = v;
            Vh[k * 32 + j] = v;
            if (j == 0) {
                dd[k] = Af[k * kTp + k];
                ee[k] = alpha;
                bb[k] = beta;
            }
        }
        float a[16], vc[16];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const float4 t = *reinterpret_cast<const float4 *>(Af + j * kTp + c0 + 4 * u);
            const float4 q = *reinterpret_cast<const float4 *>(vv + c0 + 4 * u);
            a[4 * u] = t.x; a[4 * u + 1] = t.y; a[4 * u + 2] = t.z; a[4 * u + 3] = t.w;
            vc[4 * u] = q.x; vc[4 * u + 1] = q.y; vc[4 * u + 2] = q.z; vc[4 * u + 3] = q.w;
        }
        float part = 0.0f;
#pragma unroll
        for (int c = 0; c < 16; ++c) part = fmaf(a[c], vc[c], part);
        float pr = (part + __shfl_xor(part, 32)) * beta;          // p = beta A v
        const float K = 0.5f * beta * half32_sum(pr * v);
        const float w = pr - K * v;                               // (rows <= k: v = 0, and p is not used there)
        if (h == 0) ww[j] = (j > k) ? w : 0.0f;
        const float wr = (j > k) ? w : 0.0f;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const float4 q = *reinterpret_cast<const float4 *>(ww + c0 + 4 * u);
            float4 t;
            t.x = a[4 * u] - (v * q.x + wr * vc[4 * u]);
            t.y = a[4 * u + 1] - (v * q.y + wr * vc[4 * u + 1]);
            t.z = a[4 * u + 2] - (v * q.z + wr * vc[4 * u + 2]);
            t.w = a[4 * u + 3] - (v * q.w + wr * vc[4 * u + 3]);
            *reinterpret_cast<float4 *>(Af + j * kTp + c0 + 4 * u) = t;
        }
    }
    EVC_STAMP(11);
    if (threadIdx.x == 0) {
        dd[m - 2] = Af[(m - 2) * kTp + m - 2];
        dd[m - 1] = Af[(m - 1) * kTp + m - 1];
        ee[m - 2] = Af[(m - 1) * kTp + m - 2];
        ee[m - 1] = 0.0f;
    }
    __syncthreads();
    // ---- eigenvalue j by multisection: one abscissa per lane, 8 per eigenvalue (4 waves x 2 halves) -> 9 sub-intervals
    //      per round; the Sturm counts of a round are exchanged through LDS (double-buffered: one barrier per round)
    float dr[32], e2[32];
#pragma unroll
    for (int i = 0; i < 32; ++i) {
        dr[i] = i < m ? dd[i] : 0.0f;
        const float e = (i + 1 < m) ? ee[i] : 0.0f;
        e2[i] = e * e;
    }
    float glo, ghi, emx = 0.0f;
    {
        const float ea = (j > 0 && j < m) ? fabsf(ee[j - 1]) : 0.0f, eb = (j + 1 < m) ? fabsf(ee[j]) : 0.0f;
        const float dj = j < m ? dd[j] : 0.0f;
        float lo = j < m ? dj - ea - eb : 3.0e38f, hi = j < m ? dj + ea + eb : -3.0e38f, em = fmaxf(ea, eb);
#pragma unroll
        for (int off = 1; off < 32; off <<= 1) {
            lo = fminf(lo, __shfl_xor(lo, off));
            hi = fmaxf(hi, __shfl_xor(hi, off));
            em = fmaxf(em, __shfl_xor(em, off));
        }
        const float pad = 1.0e-6f * fmaxf(fabsf(lo), fabsf(hi)) + 1.0e-30f;
        glo = lo - pad;
        ghi = hi + pad;
        emx = em;
    }
    const float pivmin = 1.0e-30f + 1.0e-14f * emx * emx;
    // (the Sturm recurrence runs unguarded: a zero pivot gives q = -inf, which counts as negative and is followed by
    //  q = d - x, as IEEE arithmetic has it; e^2 is kept away from zero so that 0 * inf cannot occur)
#pragma unroll
    for (int i = 0; i < 32; ++i) e2[i] = fmaxf(e2[i], 1.0e-36f);
    float lo = glo, hi = ghi;
    const int slot = 2 * wave + h;   // 0..7; this lane evaluates abscissae 2 slot + 1, 2 slot + 2 of 16
    for (int it = 0; it < kSturmRounds; ++it) {
        const float wd = (hi - lo) * (1.0f / 17.0f);
        const float xa = lo + wd * (float)(2 * slot + 1), xb = lo + wd * (float)(2 * slot + 2);
        float qa = dr[0] - xa, qb = dr[0] - xb;
        int ca = qa < 0.0f ? 1 : 0, cb2 = qb < 0.0f ? 1 : 0;
#pragma unroll
        for (int i = 1; i < 32; ++i) {
            if (i >= m) break;   // uniform: one test per step that is taken, none behind the end
            qa = (dr[i] - xa) - e2[i - 1] * __builtin_amdgcn_rcpf(qa);
            qb = (dr[i] - xb) - e2[i - 1] * __builtin_amdgcn_rcpf(qb);
            ca += qa < 0.0f ? 1 : 0;
            cb2 += qb < 0.0f ? 1 : 0;
        }
        int *cb = cntbuf + (it & 1) * 256;
        // eigenvalue j (ascending, 0-based) is >= x  <=>  count(x) <= j
        cb[slot * 32 + j] = (ca <= j ? 1 : 0) + (cb2 <= j ? 1 : 0);
        __syncthreads();
        int below = 0;
#pragma unroll
        for (int p = 0; p < 8; ++p) below += cb[p * 32 + j];
        lo = lo + wd * (float)below;
        hi = lo + wd;
    }
    if (wave != 0) return;
    const float lam = 0.5f * (lo + hi);
    EVC_STAMP(12);
    // ---- eigenvector of the tridiagonal matrix: twisted factorisation (h = 0: from the top, h = 1: from the bottom)
    {
        auto at = [&](int ii) { return h ? m - 1 - ii : ii; };
        float D = dd[at(0)] - lam;
        for (int ii = 0; ii + 1 < m; ++ii) {
            const int pos = at(ii), nxt = at(ii + 1), ei = pos < nxt ? pos : nxt;
            if (fabsf(D) < pivmin) D = -pivmin;
            const float e = ee[ei], F = e * __builtin_amdgcn_rcpf(D);
            fD[(pos * 32 + j) * 2 + h] = D;
            fF[(ei * 32 + j) * 2 + h] = F;
            D = (dd[nxt] - lam) - F * e;
        }
        fD[(at(m - 1) * 32 + j) * 2 + h] = D;
    }
    int kt = 0;
    {
        float best = 3.0e38f;
        for (int i = 0; i < m; ++i) {
            const float g = fabsf(fD[(i * 32 + j) * 2] + fD[(i * 32 + j) * 2 + 1] - (dd[i] - lam));
            if (g < best) {
                best = g;
                kt = i;
            }
        }
    }
    float nrm = h ? 0.0f : 1.0f;
    {
        // h = 0: z_i = -L_i z_{i+1} downwards from the twist; h = 1: z_{i+1} = -U_i z_i upwards
        float z = 1.0f;
        if (j < m) {
            if (h == 0) {
                Zf[j * kZfp + kt] = 1.0f;
                for (int i = kt - 1; i >= 0; --i) {
                    z = -fF[(i * 32 + j) * 2] * z;
                    Zf[j * kZfp + i] = z;
                    nrm = fmaf(z, z, nrm);
                }
            } else {
                for (int i = kt; i + 1 < m; ++i) {
                    z = -fF[(i * 32 + j) * 2 + 1] * z;
                    Zf[j * kZfp + i + 1] = z;
                    nrm = fmaf(z, z, nrm);
                }
            }
        }
    }
    nrm += __shfl_xor(nrm, 32);
    EVC_STAMP(13);
    if (h == 0 && j < m) zn[j] = __builtin_amdgcn_rsqf(nrm);
    // ---- back-transformation z <- H_0 H_1 ... H_{m-3} z: rows c0 .. c0+15 of eigenvector j in registers
    float z[16];
#pragma unroll
    for (int c = 0; c < 16; ++c) z[c] = (j < m && c0 + c < m) ? Zf[j * kZfp + c0 + c] : 0.0f;
    for (int k = m - 3; k >= 0; --k) {
        float vk[16];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const float4 t = *reinterpret_cast<const float4 *>(Vh + k * 32 + c0 + 4 * u);
            vk[4 * u] = t.x; vk[4 * u + 1] = t.y; vk[4 * u + 2] = t.z; vk[4 * u + 3] = t.w;
        }
        float dot = 0.0f;
#pragma unroll
        for (int c = 0; c < 16; ++c) dot = fmaf(vk[c], z[c], dot);
        dot = (dot + __shfl_xor(dot, 32)) * bb[k];
#pragma unroll
        for (int c = 0; c < 16; ++c) z[c] = fmaf(-dot, vk[c], z[c]);
    }
    if (j < m)
#pragma unroll
        for (int c = 0; c < 16; ++c)
            if (c0 + c < m) Zf[j * kZfp + c0 + c] = z[c];
}

// ---- refinement on the whole workgroup: matrices of up to 32 x 32 in LDS with row pitch kRp ------------------
// C[i][j] = sum_k P[i][k] Q[j][k] ("row . row": both operands are read along contiguous rows with 16-byte LDS loads;
// 16 consecutive rows at pitch 34 doubles fall on 16 different 4-bank groups).  Thread (tj,tk) of the 16 x 16 grid owns
// i in {tj, tj+16}, j in {tk, tk+16}; rows >= m are read as row 0 and their results dropped by the caller's store.
constexpr int kRp = 34;
constexpr int kRsz = 32 * kRp;   // doubles per matrix

template <typename Store>
__device__ __forceinline__ void mm_rowrow(int m, const double *__restrict__ P, const double *__restrict__ Q, Store store) {
#ifdef EVC_SMALL_MM_MFMA
    // (both fragments are "row l & 15, columns 4 kk + (l >> 4)" reads of a pitch-kRp matrix)
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, l15 = lane & 15, l4 = lane >> 4;
    const int ti = wave >> 1, tj = wave & 1;
    if (16 * ti >= m || 16 * tj >= m) return;   // wave-uniform
    const int i = 16 * ti + l15, j = 16 * tj + l15;
    const double *pr = P + (i < m ? i : 0) * kRp, *qr = Q + (j < m ? j : 0) * kRp;
    d4s acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int kk = 0; kk < 8; ++kk)
        if (4 * kk < m) {
            const int k = 4 * kk + l4;
            const bool kv = k < m;
            const double av = pr[kv ? k : 0], bv = qr[kv ? k : 0];
            acc = __builtin_amdgcn_mfma_f64_16x16x4f64(kv ? av : 0.0, kv ? bv : 0.0, acc, 0, 0, 0);
        }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int ii = 16 * ti + l4 + 4 * r;
        if (ii < m && j < m) store(ii, j, acc[r]);
    }
#else
    const int tk = threadIdx.x & 15, tj = threadIdx.x >> 4;
    const int ia = tj, ib = tj + 16, ja = tk, jb = tk + 16;
    const double *pa = P + (ia < m ? ia : 0) * kRp, *pb = P + (ib < m ? ib : 0) * kRp;
    const double *qa = Q + (ja < m ? ja : 0) * kRp, *qb = Q + (jb < m ? jb : 0) * kRp;
    double c00 = 0, c01 = 0, c10 = 0, c11 = 0;
    const int m2 = (m + 1) & ~1;   // columns m..m2-1 are zero padding
#pragma unroll 4
    for (int k = 0; k < m2; k += 2) {
        const double2 a0 = *reinterpret_cast<const double2 *>(pa + k), a1 = *reinterpret_cast<const double2 *>(pb + k);
        const double2 b0 = *reinterpret_cast<const double2 *>(qa + k), b1 = *reinterpret_cast<const double2 *>(qb + k);
        c00 = fma(a0.x, b0.x, c00); c00 = fma(a0.y, b0.y, c00);
        c01 = fma(a0.x, b1.x, c01); c01 = fma(a0.y, b1.y, c01);
        c10 = fma(a1.x, b0.x, c10); c10 = fma(a1.y, b0.y, c10);
        c11 = fma(a1.x, b1.x, c11); c11 = fma(a1.y, b1.y, c11);
    }
    if (ia < m && ja < m) store(ia, ja, c00);
    if (ia < m && jb < m) store(ia, jb, c01);
    if (ib < m && ja < m) store(ib, ja, c10);
    if (ib < m && jb < m) store(ib, jb, c11);
#endif
}

// Ogita-Aishima refinement of approximate eigenvectors of the symmetric matrix Ap (pitch kRp).  On entry Z holds the
// start vectors as columns and Zt = Z^T (both pitch kRp, padding zero); B1, B2, B3: scratch matrices.  On success
// (the error contracted to rounding) returns true with Z / Zt pointing at the refined pair (two of the five buffers)
// and lam = eigenvalues; false when a pass does not contract (garbage, NaN or an unresolved cluster structure).
// Rows / columns nreal..m-1 are the decoupled dummy dimension.
__device__ __forceinline__ bool oa_refine(int m, int nreal, const double *Ap, double *&Z, double *&Zt, double *B1,
                                          double *B2, double *B3, double *lam, double *red, int max_pass) {
    const int tid = threadIdx.x;
    bool ok = false;
    double prev = 1.0e300;
    for (int pass = 0; pass < max_pass; ++pass) {
        // Wt = Zt A  (A symmetric: Wt[j][i] = sum_k Zt[j][k] A[i][k])
        if (pass == 0) EVC_STAMP(14);
        mm_rowrow(m, Zt, Ap, [&](int j, int i, double v) { B1[j * kRp + i] = v; });
        __syncthreads();
        if (pass == 0) EVC_STAMP(15);
        // S = Z^T W -> B2,  R = I - Z^T Z -> B3, in one pass over the rows of Zt
#ifdef EVC_SMALL_MM_MFMA
        {
            const int lane = tid & 63, wave = tid >> 6, l15 = lane & 15, l4 = lane >> 4;
            const int ti = wave >> 1, tj = wave & 1;
            if (16 * ti < m && 16 * tj < m) {   // wave-uniform
                const int i = 16 * ti + l15, j = 16 * tj + l15;
                const double *pr = Zt + (i < m ? i : 0) * kRp;
                const double *zr = Zt + (j < m ? j : 0) * kRp, *wr = B1 + (j < m ? j : 0) * kRp;
                d4s sa = {0.0, 0.0, 0.0, 0.0}, ra = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
                for (int kk = 0; kk < 8; ++kk)
                    if (4 * kk < m) {
                        const int k = 4 * kk + l4;
                        const bool kv = k < m;
                        const int kc = kv ? k : 0;
                        const double av = kv ? pr[kc] : 0.0;
                        sa = __builtin_amdgcn_mfma_f64_16x16x4f64(av, kv ? wr[kc] : 0.0, sa, 0, 0, 0);
                        ra = __builtin_amdgcn_mfma_f64_16x16x4f64(av, kv ? zr[kc] : 0.0, ra, 0, 0, 0);
                    }
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int ii = 16 * ti + l4 + 4 * r;
                    if (ii < m && j < m) {
                        B2[ii * kRp + j] = sa[r];
                        B3[ii * kRp + j] = (ii == j ? 1.0 : 0.0) - ra[r];
                    }
                }
            }
        }
#else
        {
            const int tk = tid & 15, tj = tid >> 4;
            const int ia = tj, ib = tj + 16, ja = tk, jb = tk + 16;
            const double *pa = Zt + (ia < m ? ia : 0) * kRp, *pb = Zt + (ib < m ? ib : 0) * kRp;
            const double *za = Zt + (ja < m ? ja : 0) * kRp, *zb = Zt + (jb < m ? jb : 0) * kRp;
            const double *wa = B1 + (ja < m ? ja : 0) * kRp, *wb = B1 + (jb < m ? jb : 0) * kRp;
            double s00 = 0, s01 = 0, s10 = 0, s11 = 0, r00 = 0, r01 = 0, r10 = 0, r11 = 0;
            const int m2 = (m + 1) & ~1;
#pragma unroll 2
            for (int k = 0; k < m2; k += 2) {
                const double2 a0 = *reinterpret_cast<const double2 *>(pa + k), a1 = *reinterpret_cast<const double2 *>(pb + k);
                const double2 y0 = *reinterpret_cast<const double2 *>(za + k), y1 = *reinterpret_cast<const double2 *>(zb + k);
                const double2 w0 = *reinterpret_cast<const double2 *>(wa + k), w1 = *reinterpret_cast<const double2 *>(wb + k);
                s00 = fma(a0.x, w0.x, s00); s00 = fma(a0.y, w0.y, s00);
                s01 = fma(a0.x, w1.x, s01); s01 = fma(a0.y, w1.y, s01);
                s10 = fma(a1.x, w0.x, s10); s10 = fma(a1.y, w0.y, s10);
                s11 = fma(a1.x, w1.x, s11); s11 = fma(a1.y, w1.y, s11);
                r00 = fma(a0.x, y0.x, r00); r00 = fma(a0.y, y0.y, r00);
                r01 = fma(a0.x, y1.x, r01); r01 = fma(a0.y, y1.y, r01);
                r10 = fma(a1.x, y0.x, r10); r10 = fma(a1.y, y0.y, r10);
                r11 = fma(a1.x, y1.x, r11); r11 = fma(a1.y, y1.y, r11);
            }
            if (ia < m && ja < m) { B2[ia * kRp + ja] = s00; B3[ia * kRp + ja] = (ia == ja ? 1.0 : 0.0) - r00; }
            if (ia < m && jb < m) { B2[ia * kRp + jb] = s01; B3[ia * kRp + jb] = -r01; }
            if (ib < m && ja < m) { B2[ib * kRp + ja] = s10; B3[ib * kRp + ja] = -r10; }
            if (ib < m && jb < m) { B2[ib * kRp + jb] = s11; B3[ib * kRp + jb] = (ib == jb ? 1.0 : 0.0) - r11; }
        }
#endif
        __syncthreads();
        if (pass == 0) EVC_STAMP(16);
        const double *S = B2, *R = B3;
        // (every wave evaluates the m Rayleigh quotients itself: its maximum needs no exchange, one barrier publishes lam)
        double lm = 0.0;
        {
            const int ln = tid & 63;
            if (ln < m) {
                lm = S[ln * kRp + ln] / (1.0 - R[ln * kRp + ln]);
                if (tid < m) lam[tid] = lm;
                lm = fabs(lm);
            }
        }
        const double lmax = wave_max_nan(lm);
        __syncthreads();
        if (pass == 0) EVC_STAMP(17);
        // E^T -> B1 (Wt is consumed).  E = R/2 + a, a antisymmetric: a_ij = sh / (l_j - l_i) to first order with
        // sh = S_ij + (l_i + l_j)/2 R_ij; evaluated as the tangent of the Jacobi angle of the 2 x 2 problem
        // [[l_i, sh], [sh, l_j]], which is the same number for well separated pairs and stays bounded (|a| <= 1) for
        // close ones -- no cluster threshold.  Elements below the rounding floor are not rotated (an exactly
        // degenerate eigenspace keeps whatever orthonormal basis it has).  Convergence measure: the rotation, but
        // never more than |sh| relative to 1e-8 max|l| (a large rotation inside a numerically degenerate pair moves
        // nothing that any smooth function of A can see), and the symmetric part.
        // rounding leaves ~m eps max|l| in every element of S: below `conv_floor` an element says nothing about
        // convergence, and if the rotation it asks for is large (a numerically degenerate pair: any orthonormal basis of
        // its span is as good as any other) it is not applied at all; small rotations are applied down to `rot_floor`,
        // which is what resolves eigenvectors of tiny eigenvalues as far as the arithmetic allows.  A large rotation
        // of a significant element (a close pair found badly mixed) is limited to 0.3 per pass: the update is first
        // order, and the symmetric part repairs the t^2 loss of orthogonality in the next pass.
        const double rot_floor = 1.8e-15 * lmax, conv_floor = 1.5e-14 * (double)m * lmax;
        double emax = 0.0, rmax = 0.0;
        for (int idx = tid; idx < m * 32; idx += kThreads) {
            const int i = idx >> 5, j = idx & 31;
            if (j < m) {
                const double rv = R[i * kRp + j];
                double e = 0.5 * rv, meas = fabs(e);
                rmax = nanmax(rmax, meas);
                // (the decoupled dummy dimension of an odd problem has S_ij = R_ij = 0 exactly: no rotation -- its
                //  "eigenvalue" 0 may sit arbitrarily close to a real one)
                if (i != j && i < nreal && j < nreal) {
                    const int lo = i < j ? i : j, hi = i < j ? j : i;
                    const double d = lam[hi] - lam[lo];
                    const double sh = S[i * kRp + j] + 0.5 * (lam[i] + lam[j]) * rv;
                    double t = 0.0;
                    if (fabs(sh) > rot_floor || sh != sh) {
                        const double hd = 0.5 * d, rt = sqrt(fma(hd, hd, sh * sh));
                        t = sh / (hd + (hd < 0.0 ? -rt : rt));
                        if (fabs(sh) > conv_floor || sh != sh) {
                            meas = nanmax(meas, fabs(t));
                            t = t > 0.3 ? 0.3 : (t < -0.3 ? -0.3 : t);
                        } else if (fabs(t) > 1.0e-3) {
                            t = 0.0;
                        }
                    }
                    e += (i < j) ? t : -t;
                }
                B1[j * kRp + i] = e;
                emax = nanmax(emax, meas);
            }
        }
        if (pass == 0) EVC_STAMP(18);
        block_max_nan2(emax, rmax, red);
        EVC_STAMP(3 + pass);
        EVC_DBGVAL(pass, emax);
        // give up on NaNs, on vectors that are far from orthonormal (the first-order update cannot repair that) and
        // when the passes stop contracting; large ROTATIONS alone are fine: inside an eigenspace that is degenerate
        // to working precision they are arbitrary and harmless, elsewhere they proceed 0.3 rad per pass
        if (!(rmax < 0.5) || emax != emax || (pass >= 3 && !(emax < prev))) break;
        // Zt' = Zt + E^T Z^T: Zt'[j][i] = Zt[j][i] + sum_k Et[j][k] Z[i][k]; stored both ways (S and R are consumed)
        mm_rowrow(m, B1, Z, [&](int j, int i, double v) {
            const double z = Zt[j * kRp + i] + v;
            B2[j * kRp + i] = z;
            B3[i * kRp + j] = z;
        });
        __syncthreads();
        double *t = Zt;
        Zt = B2;
        B2 = t;
        t = Z;
        Z = B3;
        B3 = t;
        prev = emax;
        if (emax < 3.0e-8) {   // the pass just applied leaves an error of ~emax^2
            ok = true;
            // ... in the eigenvector directions; the rotations applied below the convergence floor (tiny eigenvalue
            // gaps: t = noise-level element / gap can reach 1e-3) still cost t^2 of orthogonality, which one
            // symmetric-only step Z <- Z (I + R / 2) repairs
            {
                mm_rowrow(m, Zt, Zt, [&](int i, int j, double v) { B1[j * kRp + i] = 0.5 * ((i == j ? 1.0 : 0.0) - v); });
                __syncthreads();
                mm_rowrow(m, B1, Z, [&](int j, int i, double v) {
                    const double z = Zt[j * kRp + i] + v;
                    B2[j * kRp + i] = z;
                    B3[i * kRp + j] = z;
                });
                __syncthreads();
                double *t2 = Zt;
                Zt = B2;
                B2 = t2;
                t2 = Z;
                Z = B3;
                B3 = t2;
            }
            break;
        }
    }
    return ok;
}

// Eigen-decomposition of the symmetric m x m matrix A (LDS, pitch m, both triangles; m <= 32, m even, a trailing
// decoupled dummy dimension allowed): on return diag(A) = eigenvalues, V (pitch m) = eigenvectors as columns.
//   warm: V holds the eigenvectors of a nearby problem (any garbage is detected): refinement starts from them;
//   otherwise, or when that does not contract: FP32 tridiagonal start, then FP32 Jacobi start, then FP64 Jacobi.
//   nreal: rows/columns nreal..m-1 are the decoupled dummy dimension of an odd problem.
// A + shift I must be positive definite.  Scratch R6: six matrices of kRsz doubles; Gc: kJwMax x kJwPitch doubles
// (also serves as the FP32 column buffer); lam: m; red: 8 doubles.
__device__ __forceinline__ void eigh_small(double *A, double *V, int m, int nreal, double shift, bool warm, int fast,
                                           double *R6, double *Gc, double *lam, double *red) {
    const int tid = threadIdx.x;
    bool ok = false;
    const bool tri = fast > 1;   // fast: 0 FP64 Jacobi, 1 FP32 Jacobi + refinement, 2 FP32 tridiagonal start first
    if (fast) {
        double *Ap = R6, *Z = R6 + kRsz, *Zt = R6 + 2 * kRsz, *B1 = R6 + 3 * kRsz, *B2 = R6 + 4 * kRsz, *B3 = R6 + 5 * kRsz;
        float *Gf = reinterpret_cast<float *>(Gc);
        EVC_STAMP(0);
        for (int idx = tid; idx < kRsz; idx += kThreads) {
            const int i = idx / kRp, j = idx - i * kRp;
            Ap[idx] = (i < m && j < m) ? A[i * m + j] : 0.0;
        }
        // ONE refinement call site, fed by up to three kinds of start vectors in turn (a loop that is not unrolled: every
        // inlined copy of the refinement is ~15 KB of code, and the instruction cache of a CU pair holds 64 KB):
        //   stage 0  the previous call's eigenvectors (warm start);
        //   stage 1  FP32 tridiagonalisation + multisection + twisted factorisation (tridiag_eig_wg_f32);
        //   stage 2  FP32 one-sided Jacobi on G0 = (A + shift I) / max|.|.
        double amax = 0.0;
        bool sane = true, have_amax = false;
#pragma unroll 1
        for (int stage = warm ? 0 : 1; stage < 3 && !ok && sane; ++stage) {
            if (stage == 1 && !tri) continue;
            if (stage >= 1 && !have_amax) {
                for (int idx = tid; idx < m * m; idx += kThreads) {
                    const int i = idx / m, j = idx - i * m;
                    amax = nanmax(amax, fabs(A[idx] + (i == j ? shift : 0.0)));
                }
                amax = block_max_nan(amax, red);
                have_amax = true;
                sane = amax > 0.0 && amax < 1.0e300;   // (zero, NaN or Inf input: left to the FP64 path)
                if (!sane) break;
            }
            if (stage == 0) {
                for (int idx = tid; idx < kRsz; idx += kThreads) {
                    const int i = idx / kRp, j = idx - i * kRp;
                    const bool in = i < m && j < m;
                    Z[idx] = in ? V[i * m + j] : 0.0;
                    Zt[idx] = in ? V[j * m + i] : 0.0;
                }
                __syncthreads();
            } else if (stage == 1) {
                // FP32 start 1: tridiagonalisation + multisection + twisted factorisation (scratch: the B buffers)
                float *Af = reinterpret_cast<float *>(B1), *scr = Af + 32 * kTp, *zn = scr + 32 * 32 * 5 + 5 * 32;
                float *Zf = Gf;
                const double sc = 1.0 / amax;
                for (int idx = tid; idx < 32 * kTp; idx += kThreads) {
                    const int i = idx / kTp, j = idx - i * kTp;
                    float v = 0.0f;
                    if (i < nreal && j < nreal) v = (float)(A[i * m + j] * sc);
                    else if (i == j && i < m) v = 40.0f;   // decoupled dummy dimension: an eigenvalue outside the spectrum
                    Af[idx] = v;
                }
                __syncthreads();
                EVC_STAMP(1);
                tridiag_eig_wg_f32(Af, m, Zf, zn, scr, reinterpret_cast<int *>(Gf + 34 * 32));
                __syncthreads();
                EVC_STAMP(2);
                for (int idx = tid; idx < kRsz; idx += kThreads) {
                    const int i = idx / kRp, j = idx - i * kRp;   // Zt[i][j] = component j of eigenvector i
                    Zt[idx] = (i < m && j < m) ? (double)Zf[i * kZfp + j] * (double)zn[i] : 0.0;
                }
                __syncthreads();
                for (int idx = tid; idx < kRsz; idx += kThreads) {
                    const int i = idx / kRp, j = idx - i * kRp;
                    Z[idx] = (i < m && j < m) ? Zt[j * kRp + i] : 0.0;
                }
                __syncthreads();
            } else {
                // FP32 start 2: one-sided Jacobi on G0 = (A + shift I) / max|.|, column-major with pitch kJfPitch
                const double sc = 1.0 / amax;
                for (int idx = tid; idx < kJwMax * kJfPitch; idx += kThreads) {
                    const int j = idx / kJfPitch, i = idx - j * kJfPitch;
                    Gf[idx] = (i < m && j < m) ? (float)((A[i * m + j] + (i == j ? shift : 0.0)) * sc) : 0.0f;
                }
                __syncthreads();
                if (tid < 64) jacobi_onesided_wave_f32(Gf, m);
                __syncthreads();
                // Z0 = normalised columns (a zero column = the decoupled dummy dimension keeps its unit vector)
                if (tid < m) {
                    double nn = 0.0;
                    for (int i = 0; i < m; ++i) nn = fma((double)Gf[tid * kJfPitch + i], (double)Gf[tid * kJfPitch + i], nn);
                    lam[tid] = nn > 1.0e-60 ? 1.0 / sqrt(nn) : 0.0;
                }
                __syncthreads();
                for (int idx = tid; idx < kRsz; idx += kThreads) {
                    const int i = idx / kRp, j = idx - i * kRp;   // Zt[i][j] = Z[j][i] = component j of eigenvector i
                    double v = 0.0;
                    if (i < m && j < m) v = lam[i] > 0.0 ? (double)Gf[i * kJfPitch + j] * lam[i] : (i == j ? 1.0 : 0.0);
                    Zt[idx] = v;
                }
                __syncthreads();
                for (int idx = tid; idx < kRsz; idx += kThreads) {
                    const int i = idx / kRp, j = idx - i * kRp;
                    Z[idx] = (i < m && j < m) ? Zt[j * kRp + i] : 0.0;
                }
                __syncthreads();
            }
            ok = oa_refine(m, nreal, Ap, Z, Zt, B1, B2, B3, lam, red, 10);
            EVC_DBGVAL(21, ok ? 1.0 : 0.0);
            if (!ok) {   // the buffers may have been permuted: re-establish the roles
                Z = R6 + kRsz; Zt = R6 + 2 * kRsz; B1 = R6 + 3 * kRsz; B2 = R6 + 4 * kRsz; B3 = R6 + 5 * kRsz;
            }
        }
        EVC_STAMP(10);
        if (ok) {
            for (int idx = tid; idx < m * m; idx += kThreads) {
                const int i = idx / m, j = idx - i * m;
                V[idx] = Z[i * kRp + j];
            }
            if (tid < m) A[tid * m + tid] = lam[tid];
            __syncthreads();
        }
    }
    if (!ok) jacobi_eigh_wave(A, V, m, shift, true, Gc, lam);
}

// Warm start (EVC_FLAG_WARM_START): `prev` holds the eigenvectors of the previous, nearby problem.  If they
// are orthonormal to 1e-8 (a stale or never-written buffer is not), V <- prev (padded with the identity) and
// A <- V^T A V, which is nearly diagonal, so the sweeps that follow are two or three instead of seven or eight.
// Returns whether the rotation was applied (uniform over the workgroup).  Tmp: n*n doubles of LDS.
__device__ __forceinline__ bool warm_start_rotate(double *A, double *V, double *Tmp, int n, int m, const double *__restrict__ prev,
                                  int ldp, double *red) {
    const int tid = threadIdx.x;
    for (int idx = tid; idx < m * m; idx += kThreads) {
        const int i = idx / m, j = idx - i * m;
        V[idx] = (i < n && j < n) ? prev[i * ldp + j] : (i == j ? 1.0 : 0.0);
    }
    __syncthreads();
    mm16(n, [&](int i, int k) { return V[k * m + i]; }, [&](int k, int j) { return V[k * m + j]; },
         [&](int i, int j, double v) { Tmp[i * n + j] = v - (i == j ? 1.0 : 0.0); });
    __syncthreads();
    double dev = 0.0;
    for (int idx = tid; idx < n * n; idx += kThreads) dev = fma(Tmp[idx], Tmp[idx], dev);
    dev = block_sum<4>(dev, red);
    if (!(dev < 1.0e-16)) return false;  // also catches NaN
    mm16(n, [&](int i, int k) { return A[i * m + k]; }, [&](int k, int j) { return V[k * m + j]; },
         [&](int i, int j, double v) { Tmp[i * n + j] = v; });
    __syncthreads();
    mm16(n, [&](int i, int k) { return V[k * m + i]; }, [&](int k, int j) { return Tmp[k * n + j]; },
         [&](int i, int j, double v) { A[i * m + j] = v; });
    __syncthreads();
    for (int idx = tid; idx < n * n; idx += kThreads) {  // exact symmetry, as the rotations assume
        const int i = idx / n, j = idx - i * n;
        if (i > j) {
            const double v = 0.5 * (A[i * m + j] + A[j * m + i]);
            A[i * m + j] = v;
            A[j * m + i] = v;
        }
    }
    __syncthreads();
    return true;
}

// ------------------------------------------------------------------ Loewdin: S^-1/2 without the eigensolver
// X = S^-1/2 and h1 = X h X are all the ENERGY phase needs from the Loewdin step (the eigenvectors and eigenvalues of S
// only enter the response term at the very end of the gradient, launch_grad_final).  The coupled Newton-Schulz iteration
//     Y_0 = S / c,  Z_0 = I,  T_k = (3 I - Z_k Y_k) / 2,  Y_k+1 = Y_k T_k,  Z_k+1 = Z_k T_k      (c = ||S||_inf >= lambda_max)
// (Higham, Functions of Matrices, eq. 6.35: Y -> (S/c)^1/2, Z -> (S/c)^-1/2, quadratically; all iterates are polynomials
// in S, hence symmetric and commuting) is three 32^3 products per step on the FP64 matrix cores, one 16 x 16 output tile
// per wave: ~0.8 us per step, 10-14 steps for cond(S) ~ 10^2-10^3, against ~65 us for the full eigendecomposition.
// One more step of the uncoupled form X <- X (3 I - X S X) / 2 on the ORIGINAL S removes what the coupled iterates have
// drifted and yields the residual max |I - X S X| the result is accepted on; anything else (S not positive definite,
// cond(S) beyond ~10^8, NaNs) returns false and the caller takes the eigensolver.
//
// LDS: 32 x 32 matrices at pitch 48 doubles -- the four rows a fragment read touches (k = lane >> 4) are 16 doubles
// apart modulo 32, i.e. on disjoint halves of the 64 banks; a symmetric A operand is read along the rows of A^T = A
// (lane & 15 -> consecutive addresses), so no operand is ever read with a stride.
constexpr int kNsP = 48;
constexpr int kNsSz = 32 * kNsP;
constexpr int kNsMaxIter = 64;
constexpr int kNsDoubles = 6 * kNsSz + 8;

// (kmax = 4 for n <= 16: K = 16 covers the matrix, and only the tile (0, 0) is active then)
__device__ __forceinline__ d4s ns_tile(const double *A, const double *B, int ao, int bo, int kmax) {
    d4s acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int kk = 0; kk < 8; ++kk)
        if (kk < kmax)   // uniform
            acc = __builtin_amdgcn_mfma_f64_16x16x4f64(A[kk * 4 * kNsP + ao], B[kk * 4 * kNsP + bo], acc, 0, 0, 0);
    return acc;
}

__device__ bool loewdin_ns(const double *__restrict__ S, const double *__restrict__ h, double *__restrict__ X,
                           double *__restrict__ h1, int n, double *sm) {
    double *S0 = sm, *Yc = S0 + kNsSz, *Zc = Yc + kNsSz, *Yn = Zc + kNsSz, *Zn = Yn + kNsSz, *Tm = Zn + kNsSz;
    double *red = Tm + kNsSz;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l15 = lane & 15, l4 = lane >> 4;
    const int ti = wave >> 1, tj = wave & 1;
    const int ao = l4 * kNsP + 16 * ti + l15, bo = l4 * kNsP + 16 * tj + l15;
    const int oi = 16 * ti + l4, oj = 16 * tj + l15;   // output element of register r: (oi + 4 r, oj)
    // n <= 16: one tile holds the matrix -- the other three waves idle (their padding tiles are never read: K = 16)
    const bool act = 16 * ti < n && 16 * tj < n;
    const int kmax = n <= 16 ? 4 : 8;
    double hreg[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const int idx = tid + u * kThreads, i = idx >> 5, j = idx & 31;
        const bool in = i < n && j < n;
        // (LAPACK's eigh reads the lower triangle: so does this)
        S0[i * kNsP + j] = in ? S[i >= j ? i * n + j : j * n + i] : (i == j ? 1.0 : 0.0);
        hreg[u] = (in && h) ? h[i * n + j] : 0.0;
    }
    __syncthreads();
    if (wave == 0) {
        double cs = 0.0;
        if (lane < 32)
            for (int i = 0; i < 32; ++i) cs += fabs(S0[i * kNsP + lane]);
        cs = wave_max_nan(cs);
        if (lane == 0) red[4] = cs;
    }
    __syncthreads();
    const double c = red[4];
    if (!(c > 0.0) || !(c < 1.0e300)) return false;   // (uniform: zero matrix, NaN, Inf)
    const double rc = 1.0 / c;
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const int idx = tid + u * kThreads, i = idx >> 5, j = idx & 31;
        const bool in = i < n && j < n;
        Yc[i * kNsP + j] = in ? S0[i * kNsP + j] * rc : (i == j ? 1.0 : 0.0);
        Zc[i * kNsP + j] = i == j ? 1.0 : 0.0;
    }
    __syncthreads();
    bool ok = false;
    double eprev = 2.0;
    int it = 0;
#pragma unroll 1
    for (; it < kNsMaxIter; ++it) {
        double e = 0.0;
        if (act) {
            const d4s p = ns_tile(Zc, Yc, ao, bo, kmax);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const double dlt = (oi + 4 * r == oj) ? 1.0 : 0.0;
                e = nanmax(e, fabs(dlt - p[r]));
                Tm[(oi + 4 * r) * kNsP + oj] = 1.5 * dlt - 0.5 * p[r];
            }
            e = wave_max_nan(e);
        }
        if (lane == 0) red[wave] = e;
        __syncthreads();
        e = nanmax(nanmax(red[0], red[1]), nanmax(red[2], red[3]));
        if (act) {
            d4s yn = {0.0, 0.0, 0.0, 0.0}, zn = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int kk = 0; kk < 8; ++kk)
                if (kk < kmax) {
                    const double b = Tm[kk * 4 * kNsP + bo];
                    yn = __builtin_amdgcn_mfma_f64_16x16x4f64(Yc[kk * 4 * kNsP + ao], b, yn, 0, 0, 0);
                    zn = __builtin_amdgcn_mfma_f64_16x16x4f64(Zc[kk * 4 * kNsP + ao], b, zn, 0, 0, 0);
                }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                Yn[(oi + 4 * r) * kNsP + oj] = yn[r];
                Zn[(oi + 4 * r) * kNsP + oj] = zn[r];
            }
        }
        __syncthreads();
        double *t0 = Yc;
        Yc = Yn;
        Yn = t0;
        t0 = Zc;
        Zc = Zn;
        Zn = t0;
        if (e != e) break;
        // e = max |I - Z Y| BEFORE this step; the step squares it (3/4 e^2).  Below 1e-3 a step that does not even halve
        // it has reached the rounding floor of an ill-conditioned S: the residual test below decides.
        if (e < 1.0e-8 || (e < 1.0e-3 && e > 0.5 * eprev)) {
            ok = true;
            ++it;
            break;
        }
        eprev = e;
    }
    EVC_DBGVAL(50, it);
    if (!ok) return false;
    // X = Z / sqrt(c), then one step on the original S:  W = S X,  P = X W,  X <- X (3 I - P) / 2
    const double rsq = sqrt(rc);
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const int idx = tid + u * kThreads, i = idx >> 5, j = idx & 31;
        if (i < n && j < n) Zc[i * kNsP + j] *= rsq;
    }
    __syncthreads();
    if (act) {
        const d4s wv = ns_tile(S0, Zc, ao, bo, kmax);
#pragma unroll
        for (int r = 0; r < 4; ++r) Tm[(oi + 4 * r) * kNsP + oj] = wv[r];
    }
    __syncthreads();
    double res;
    {
        double e = 0.0;
        if (act) {
            const d4s p = ns_tile(Zc, Tm, ao, bo, kmax);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const double dlt = (oi + 4 * r == oj) ? 1.0 : 0.0;
                e = nanmax(e, fabs(dlt - p[r]));
                Yn[(oi + 4 * r) * kNsP + oj] = 1.5 * dlt - 0.5 * p[r];
            }
            e = wave_max_nan(e);
        }
        if (lane == 0) red[wave] = e;
        __syncthreads();
        res = nanmax(nanmax(red[0], red[1]), nanmax(red[2], red[3]));
    }
    EVC_DBGVAL(51, res);
    if (!(res < 1.0e-7)) return false;   // (after the step: ~res^2)
    if (act) {
        const d4s xv = ns_tile(Zc, Yn, ao, bo, kmax);
#pragma unroll
        for (int r = 0; r < 4; ++r) Zn[(oi + 4 * r) * kNsP + oj] = xv[r];
    }
    __syncthreads();
    // symmetrised X -> Yc and the caller; h^T -> Tm (the A operand is read along rows of its transpose)
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const int idx = tid + u * kThreads, i = idx >> 5, j = idx & 31;
        const bool in = i < n && j < n;
        const double v = in ? 0.5 * (Zn[i * kNsP + j] + Zn[j * kNsP + i]) : (i == j ? 1.0 : 0.0);
        Yc[i * kNsP + j] = v;
        if (in) X[i * n + j] = v;
        Tm[j * kNsP + i] = hreg[u];
    }
    if (!(h && h1)) return true;
    __syncthreads();
    if (act) {
        const d4s wv = ns_tile(Tm, Yc, ao, bo, kmax);   // W = h X
#pragma unroll
        for (int r = 0; r < 4; ++r) Yn[(oi + 4 * r) * kNsP + oj] = wv[r];
    }
    __syncthreads();
    if (act) {
        const d4s hv = ns_tile(Yc, Yn, ao, bo, kmax);   // h1 = X W
#pragma unroll
        for (int r = 0; r < 4; ++r)
            if (oi + 4 * r < n && oj < n) h1[(oi + 4 * r) * n + oj] = hv[r];
    }
    return true;
}

// The same for 32 < n <= 64 (cc-pVTZ water: n = 58): 64 x 64 matrices, wave w owns the row strip 16 w ... 16 w + 15 of
// every product (four 16 x 16 tiles: one A fragment serves four MFMAs), three matrices in LDS at pitch 64 with the 16-
// double halves of odd rows swapped (idx below: the fragment rows of lane groups l4 and l4 + 1 fall on disjoint bank
// halves, as the padding does at 32), the iterates updated in place between two barriers.  ~5 us per step (192 MFMAs
// per wave), 12-16 steps.  One workgroup of 256 threads per geometry; the result flag goes to `flag[g]` (1 = X and h1
// written), which the eigensolver launch that follows in the stream (loewdin_big_kernel, part 3) reads first.
constexpr int kNs64Sz = 64 * 64;
__device__ __forceinline__ int ns64_idx(int row, int col) { return row * 64 + (col ^ ((row & 1) << 4)); }

// strip of A.B for symmetric A: acc[tj] (+)= sum_k A[k][16 w + i] B[k][16 tj + j]
__device__ __forceinline__ void ns64_strip(const double *A, const double *B, int wave, int l15, int l4, d4s (&acc)[4]) {
#pragma unroll
    for (int tj = 0; tj < 4; ++tj) acc[tj] = (d4s){0.0, 0.0, 0.0, 0.0};
#pragma unroll 4
    for (int kk = 0; kk < 16; ++kk) {
        const int row = 4 * kk + l4;
        const double av = A[ns64_idx(row, 16 * wave + l15)];
#pragma unroll
        for (int tj = 0; tj < 4; ++tj)
            acc[tj] = __builtin_amdgcn_mfma_f64_16x16x4f64(av, B[ns64_idx(row, 16 * tj + l15)], acc[tj], 0, 0, 0);
    }
}

__global__ __launch_bounds__(kThreads) void loewdin_ns64_kernel(LoewdinArgs a) {
    extern __shared__ __align__(16) double sm[];
    const int n = a.n;
    const int64_t g = blockIdx.x;
    const double *__restrict__ S = a.S + g * a.sS;
    const double *__restrict__ h = a.h ? a.h + g * a.sh : nullptr;
    double *__restrict__ X = a.X + g * a.sws;
    double *__restrict__ h1 = a.h1 ? a.h1 + g * a.sws : nullptr;
    double *__restrict__ flag = a.flag + g * a.sws;
    double *Y = sm, *Z = Y + kNs64Sz, *Tm = Z + kNs64Sz, *red = Tm + kNs64Sz;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l15 = lane & 15, l4 = lane >> 4;
    auto fail = [&]() {
        if (tid == 0) *flag = 0.0;
    };
    // S (lower triangle mirrored, identity in the padding) -> Tm; column sums -> c = ||S||_inf
    for (int idx = tid; idx < kNs64Sz; idx += kThreads) {
        const int i = idx >> 6, j = idx & 63;
        Tm[ns64_idx(i, j)] = (i < n && j < n) ? S[i >= j ? i * n + j : j * n + i] : (i == j ? 1.0 : 0.0);
    }
    __syncthreads();
    {
        double cs = 0.0;
        if (wave == 0)
            for (int i = 0; i < 64; ++i) cs += fabs(Tm[ns64_idx(i, lane)]);
        cs = wave_max_nan(cs);
        if (tid == 0) red[4] = cs;
    }
    __syncthreads();
    const double c = red[4];
    if (!(c > 0.0) || !(c < 1.0e300)) {
        fail();
        return;
    }
    const double rc = 1.0 / c;
    for (int idx = tid; idx < kNs64Sz; idx += kThreads) {
        const int i = idx >> 6, j = idx & 63;
        const int k = ns64_idx(i, j);
        Y[k] = (i < n && j < n) ? Tm[k] * rc : (i == j ? 1.0 : 0.0);
        Z[k] = i == j ? 1.0 : 0.0;
    }
    __syncthreads();
    const int oi = 16 * wave + l4;   // output element (register r of tile tj): (oi + 4 r, 16 tj + l15)
    bool ok = false;
    double eprev = 2.0;
    int it = 0;
    d4s acc[4], acc2[4];
#pragma unroll 1
    for (; it < kNsMaxIter; ++it) {
        ns64_strip(Z, Y, wave, l15, l4, acc);
        double e = 0.0;
#pragma unroll
        for (int tj = 0; tj < 4; ++tj)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int i = oi + 4 * r, j = 16 * tj + l15;
                const double dlt = i == j ? 1.0 : 0.0;
                e = nanmax(e, fabs(dlt - acc[tj][r]));
                Tm[ns64_idx(i, j)] = 1.5 * dlt - 0.5 * acc[tj][r];
            }
        e = wave_max_nan(e);
        if (lane == 0) red[wave] = e;
        __syncthreads();
        e = nanmax(nanmax(red[0], red[1]), nanmax(red[2], red[3]));
        ns64_strip(Y, Tm, wave, l15, l4, acc);
        ns64_strip(Z, Tm, wave, l15, l4, acc2);
        __syncthreads();
#pragma unroll
        for (int tj = 0; tj < 4; ++tj)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int k = ns64_idx(oi + 4 * r, 16 * tj + l15);
                Y[k] = acc[tj][r];
                Z[k] = acc2[tj][r];
            }
        __syncthreads();
        if (e != e) break;
        if (e < 1.0e-8 || (e < 1.0e-3 && e > 0.5 * eprev)) {
            ok = true;
            ++it;
            break;
        }
        eprev = e;
    }
    EVC_DBGVAL(52, it);
    if (!ok) {
        fail();
        return;
    }
    // X = Z / sqrt(c); one step on the original S:  W = S X,  P = X W,  X <- X (3 I - P) / 2
    const double rsq = sqrt(rc);
    for (int idx = tid; idx < kNs64Sz; idx += kThreads) {
        const int i = idx >> 6, j = idx & 63;
        const int k = ns64_idx(i, j);
        const bool in = i < n && j < n;
        if (in) Z[k] *= rsq;
        Tm[k] = in ? S[i >= j ? i * n + j : j * n + i] : (i == j ? 1.0 : 0.0);
    }
    __syncthreads();
    ns64_strip(Tm, Z, wave, l15, l4, acc);   // W = S X
    __syncthreads();
#pragma unroll
    for (int tj = 0; tj < 4; ++tj)
#pragma unroll
        for (int r = 0; r < 4; ++r) Tm[ns64_idx(oi + 4 * r, 16 * tj + l15)] = acc[tj][r];
    __syncthreads();
    ns64_strip(Z, Tm, wave, l15, l4, acc);   // P = X W
    double res = 0.0;
#pragma unroll
    for (int tj = 0; tj < 4; ++tj)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int i = oi + 4 * r, j = 16 * tj + l15;
            const double dlt = i == j ? 1.0 : 0.0;
            res = nanmax(res, fabs(dlt - acc[tj][r]));
            Y[ns64_idx(i, j)] = 1.5 * dlt - 0.5 * acc[tj][r];
        }
    res = wave_max_nan(res);
    if (lane == 0) red[wave] = res;
    __syncthreads();
    res = nanmax(nanmax(red[0], red[1]), nanmax(red[2], red[3]));
    EVC_DBGVAL(53, res);
    if (!(res < 1.0e-7)) {
        fail();
        return;
    }
    ns64_strip(Z, Y, wave, l15, l4, acc);    // X' = X T
    __syncthreads();
#pragma unroll
    for (int tj = 0; tj < 4; ++tj)
#pragma unroll
        for (int r = 0; r < 4; ++r) Tm[ns64_idx(oi + 4 * r, 16 * tj + l15)] = acc[tj][r];
    __syncthreads();
    // symmetrised X -> Z and the caller; h^T -> Y
    for (int idx = tid; idx < kNs64Sz; idx += kThreads) {
        const int i = idx >> 6, j = idx & 63;
        const bool in = i < n && j < n;
        const double v = in ? 0.5 * (Tm[ns64_idx(i, j)] + Tm[ns64_idx(j, i)]) : (i == j ? 1.0 : 0.0);
        Z[ns64_idx(i, j)] = v;
        if (in) X[i * n + j] = v;
        Y[ns64_idx(j, i)] = (in && h) ? h[i * n + j] : 0.0;
    }
    if (tid == 0) *flag = 1.0;
    if (!(h && h1)) return;
    __syncthreads();
    ns64_strip(Y, Z, wave, l15, l4, acc);    // W = h X  (A operand read along the rows of h^T)
    __syncthreads();
#pragma unroll
    for (int tj = 0; tj < 4; ++tj)
#pragma unroll
        for (int r = 0; r < 4; ++r) Tm[ns64_idx(oi + 4 * r, 16 * tj + l15)] = acc[tj][r];
    __syncthreads();
    ns64_strip(Z, Tm, wave, l15, l4, acc);   // h1 = X W
#pragma unroll
    for (int tj = 0; tj < 4; ++tj)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int i = oi + 4 * r, j = 16 * tj + l15;
            if (i < n && j < n) h1[i * n + j] = acc[tj][r];
        }
}

// ------------------------------------------------------------------ Loewdin
__device__ __forceinline__ void loewdin_body(LoewdinArgs a, const int64_t g) {
    const int n = a.n;
    const double *__restrict__ S = a.S + g * a.sS;
    const double *__restrict__ h = a.h ? a.h + g * a.sh : nullptr;
    double *__restrict__ X = a.X + g * a.sws;
    double *__restrict__ U = a.U + g * a.sws;
    double *__restrict__ sv = a.s + g * a.sws;
    double *__restrict__ h1 = a.h1 ? a.h1 + g * a.sws : nullptr;
    extern __shared__ __align__(16) double sm[];
    const int m = (n + 1) & ~1;
    // part = 1: X and h1 only, by Newton-Schulz (the eigensolver below only if that declines, and then without touching
    // U and s, which a part = 2 launch on another stream is writing); part = 2: U and s only; 0: everything
    const bool want_x = a.part != 2, want_u = a.part != 1;
    if (a.part == 1 && m <= kJwMax && a.fast) {
        if (loewdin_ns(S, h, X, h1, n, sm)) return;
        __syncthreads();
    }
    double *A = sm;              // m*m   (later: hcore)
    double *V = A + m * m;       // m*m   (later: T = h X)
    double *Xs = V + m * m;      // m*m   (uses n*n)
    double *rot = Xs + m * m;    // m
    double *red = rot + m;       // 8
    double *f = red + 8;         // m
    double *Gc = f + m;          // kJwMax x kJwPitch (only carved for m <= kJwMax), then the refinement's six matrices
    double *R6 = Gc + kJwMax * kJwPitch;
    const int tid = threadIdx.x, tk = tid & 15, tj = tid >> 4;
    // LAPACK's eigh reads one triangle; numpy.linalg.eigh uses the lower one.
    for (int idx = tid; idx < m * m; idx += kThreads) A[idx] = 0.0;
    __syncthreads();
    for (int idx = tid; idx < n * n; idx += kThreads) {
        const int i = idx / n, j = idx - i * n;
        const double v = S[idx];
        if (i >= j) {
            A[i * m + j] = v;
            A[j * m + i] = v;
        }
    }
    __syncthreads();
    // warm start from the eigenvectors the previous call left in U (same workspace, nearby geometry)
    // (hcore is only needed behind the eigensolver: requested now, its latency is gone by then)
    double hpre[(kRsz + kThreads - 1) / kThreads];
    if (m <= kJwMax && a.fast) {
#pragma unroll
        for (int u = 0; u < (kRsz + kThreads - 1) / kThreads; ++u) {
            const int idx = tid + u * kThreads, i = idx / kRp, j = idx - i * kRp;
            hpre[u] = (idx < kRsz && i < n && j < n && h) ? h[i * n + j] : 0.0;
        }
    }
    if (m <= kJwMax && a.fast) {
        if (a.warm) {   // refinement straight from U (a stale or never-written buffer makes it fall back)
            for (int idx = tid; idx < m * m; idx += kThreads) {
                const int i = idx / m, j = idx - i * m;
                V[idx] = (i < n && j < n) ? U[i * n + j] : (i == j ? 1.0 : 0.0);
            }
            __syncthreads();
        }
        eigh_small(A, V, m, n, 0.0, a.warm != 0, a.fast, R6, Gc, f, red);
    } else {
        const bool warm = a.warm && warm_start_rotate(A, V, Xs, n, m, U, n, red);
        if (m <= kJwMax) jacobi_eigh_wave(A, V, m, 0.0, !warm, Gc, f);
        else jacobi_eigh_lds(A, V, m, rot, red, !warm);
    }
    if (tid < m) {
        const double s = A[tid * m + tid];
        f[tid] = (tid < n && s > 1.0e-15) ? 1.0 / sqrt(s) : 0.0;
        if (tid < n && want_u) sv[tid] = s;
    }
    __syncthreads();
    if (m <= kJwMax && a.fast) {
        // X = V diag(f) V^T and h1 = X^T h X as row.row products at pitch kRp (the refinement's buffers are free)
        double *Vf = R6, *Vp = R6 + kRsz, *Xp = R6 + 2 * kRsz, *hp = R6 + 3 * kRsz, *Tt = R6 + 4 * kRsz;
#pragma unroll
        for (int u = 0; u < (kRsz + kThreads - 1) / kThreads; ++u) {
            const int idx = tid + u * kThreads;
            if (idx < kRsz) {
                const int i = idx / kRp, j = idx - i * kRp;
                const bool in = i < n && j < n;
                const double v = in ? V[i * m + j] : 0.0;
                Vp[idx] = v;
                Vf[idx] = in ? v * f[j] : 0.0;
                hp[idx] = hpre[u];
                Xp[idx] = 0.0;
                Tt[idx] = 0.0;
                if (in && want_u) U[i * n + j] = v;
            }
        }
        if (!want_x) return;
        __syncthreads();
        mm_rowrow(m, Vf, Vp, [&](int i, int j, double v) {
            if (i < n && j < n) {
                Xp[i * kRp + j] = v;
                X[i * n + j] = v;
            }
        });
        if (h && h1) {
            __syncthreads();
            // Tt[j][i] = (h X)[i][j] = sum_k X[j][k] h[i][k]  (X symmetric);  h1[i][j] = sum_k X[i][k] Tt[j][k]
            mm_rowrow(m, Xp, hp, [&](int j, int i, double v) { Tt[j * kRp + i] = v; });
            __syncthreads();
            mm_rowrow(m, Xp, Tt, [&](int i, int j, double v) {
                if (i < n && j < n) h1[i * n + j] = v;
            });
        }
        return;
    }
    // X = V diag(f) V^T  (a dummy column, if any, has f = 0)
    mm16(n, [&](int i, int k) { return V[i * m + k] * f[k]; }, [&](int k, int j) { return V[j * m + k]; },
         [&](int i, int j, double v) {
             Xs[i * n + j] = v;
             X[i * n + j] = v;
         });
    for (int i = tj; i < n; i += 16)
        for (int j = tk; j < n; j += 16) U[i * n + j] = V[i * m + j];
    if (h && h1) {
        copy_to_lds(A, h, n * n);
        __syncthreads();
        // T = h X (into V), h1 = X^T T
        mm16(n, [&](int i, int k) { return A[i * n + k]; }, [&](int k, int j) { return Xs[k * n + j]; },
             [&](int i, int j, double v) { V[i * n + j] = v; });
        __syncthreads();
        mm16(n, [&](int i, int k) { return Xs[k * n + i]; }, [&](int k, int j) { return V[k * n + j]; },
             [&](int i, int j, double v) { h1[i * n + j] = v; });
    }
}

__global__ __launch_bounds__(kThreads) void loewdin_kernel(LoewdinArgs a) { loewdin_body(a, blockIdx.x); }

static size_t jacobi_aux_bytes(int m) {
    return sizeof(double) * (size_t)(2 * m + 8) + 32 +
           (m <= kJwMax ? sizeof(double) * ((size_t)kJwMax * kJwPitch + (size_t)6 * kRsz) + 16 : 0);
}

// EVC_EIGH_F32: 0 = FP64 Jacobi, 1 = FP32 Jacobi + refinement, 2 (default) = FP32 tridiagonal start + refinement
static int eigh_fast_enabled() {
    static const int on = getenv("EVC_EIGH_F32") ? atoi(getenv("EVC_EIGH_F32")) : 2;
#ifdef EVC_DEBUG_STAMPS
    static bool dbg_done = false;
    if (!dbg_done) {
        dbg_done = true;
        if (const char *e = getenv("EVC_DBG_MAX_SWEEPS")) {
            const int v = atoi(e);
            (void)hipMemcpyToSymbol(HIP_SYMBOL(g_dbg_max_sweeps), &v, sizeof(int));
        }
    }
#endif
    return on;
}

bool loewdin_split_available(int n) { return n >= 1 && n <= 64 && (n > kJwMax || eigh_fast_enabled() != 0); }

int launch_loewdin(const LoewdinArgs &a_in, int count, hipStream_t st) {
    // 32 < n <= 64: three matrices in LDS; up to 96 with two of them in the caller's scratch (a_in.scratch)
    if (a_in.n > kJwMax && (a_in.n <= 64 || a_in.scratch)) {
        if (a_in.part && (a_in.n > 64 || !a_in.flag)) {
            set_error("loewdin: part=%d needs n <= 64 and a flag word per geometry", a_in.part);
            return -1;
        }
        if (a_in.part == 1) {
            // Newton-Schulz X and h1; then the eigensolver launch, which returns at once for every geometry whose flag
            // says the iteration has delivered (part 3)
            static LdsAttr attr;
            const size_t lds = sizeof(double) * (3 * kNs64Sz + 8);
            if (int rc = allow_dynamic_lds(loewdin_ns64_kernel, attr, 160 * 1024, "loewdin_ns64")) return rc;
            hipLaunchKernelGGL(loewdin_ns64_kernel, dim3(count), dim3(kThreads), lds, st, a_in);
            EVC_LAUNCH_CHECK("loewdin_ns64");
            LoewdinArgs b = a_in;
            b.part = 3;
            return launch_loewdin_big(b, count, st);
        }
        return launch_loewdin_big(a_in, count, st);
    }
    LoewdinArgs a = a_in;
    a.fast = eigh_fast_enabled();
    const int m = (a.n + 1) & ~1;
    size_t lds = sizeof(double) * (size_t)3 * m * m + jacobi_aux_bytes(m);
    if (a.part) {
        if (!loewdin_split_available(a.n)) {
            set_error("loewdin: part=%d needs n <= %d and the FP32-started eigensolver", a.part, kJwMax);
            return -1;
        }
        if (a.part == 1 && lds < sizeof(double) * kNsDoubles) lds = sizeof(double) * kNsDoubles;
    }
    static LdsAttr attr;
    if (int rc = allow_dynamic_lds(loewdin_kernel, attr, 160 * 1024, "loewdin")) return rc;
    hipLaunchKernelGGL(loewdin_kernel, dim3(count), dim3(kThreads), lds, st, a);
    EVC_LAUNCH_CHECK("loewdin");
    return 0;
}

// Cholesky factor of the symmetric positive definite T x T matrix S (lower triangle of Ssrc, pitch T) and its
// inverse B = L^-1, on ONE wave with everything in registers: lane i holds row i of L; the pivot and the column
// entries a step needs from other lanes travel through v_readlane (uniform operands), so the 2 T dependent steps
// carry no LDS round trip and no barrier.  Writes B (lower triangular) to Bi at pitch kRp; a matrix that is not
// positive definite yields NaNs.  T <= 32.  Called by wave 0 only.
__device__ __forceinline__ void chol_inverse_wave(const double *Ssrc, int T, double *Bi) {
    const int i = threadIdx.x & 31;
    double a[32], rinv[32];
#pragma unroll
    for (int k = 0; k < 32; ++k) {
        a[k] = (i < T && k < T) ? (k <= i ? Ssrc[i * T + k] : 0.0) : (k == i ? 1.0 : 0.0);
        rinv[k] = 1.0;
    }
#pragma unroll
    for (int j = 0; j < 32; ++j) {
        if (j < T) {   // uniform
            const double d = readlane_f64(a[j], j);
            double rs = __builtin_amdgcn_rsq(d);
            rs = rs * fma(-0.5 * d * rs, rs, 1.5);
            rs = rs * fma(-0.5 * d * rs, rs, 1.5);
            const double lij = a[j] * rs;   // lane j: sqrt(d)
            a[j] = lij;
            rinv[j] = rs;                   // 1 / L_jj (uniform)
#pragma unroll
            for (int k = j + 1; k < 32; ++k)
                if (k < T) a[k] = fma(-lij, readlane_f64(lij, k), a[k]);
        }
    }
    double b[32];
#pragma unroll
    for (int r = 0; r < 32; ++r) {
        b[r] = 0.0;
        if (r < T) {   // uniform
            double acc = (r == i) ? 1.0 : 0.0, acc2 = 0.0;
#pragma unroll
            for (int k = 0; k + 1 < r; k += 2) {
                acc = fma(-readlane_f64(a[k], r), b[k], acc);
                acc2 = fma(-readlane_f64(a[k + 1], r), b[k + 1], acc2);
            }
            if (r & 1) acc = fma(-readlane_f64(a[r - 1], r), b[r - 1], acc);
            b[r] = (acc + acc2) * rinv[r];
        }
    }
    if (threadIdx.x < 32 && i < T) {
#pragma unroll
        for (int r = 0; r < 32; ++r)
            if (r < T) Bi[r * kRp + i] = b[r];
    }
}

// ------------------------------------------------------------------ subspace solve
__device__ __forceinline__ void subspace_body(SolveArgs a, const int64_t gblk) {
    extern __shared__ __align__(16) double sm[];
    {
        const int64_t g = gblk;
        a.h1part += g * a.sh1;
        if (a.h2part) a.h2part += g * a.sh2;
        a.S += g * a.sS;
        a.evals += g * a.sev;
        a.evecs += g * a.svec;
        if (a.Hout) a.Hout += g * a.sH;
        if (a.w1) a.w1 += g * a.sw;
        if (a.w2) a.w2 += g * a.sw;
        if (a.w2t) a.w2t += (g - g % kMaxBatchG) * a.sw;
        if (a.w1t) a.w1t += (g - g % kMaxBatchG) * a.sw;
        if (a.vstd) a.vstd += g * a.sw;
        if (a.bcache) a.bcache += g * a.sw;
        if (a.e_shift_dev) a.e_shift = a.e_shift_dev[g];
    }
    EVC_STAMP(30);
    const int T = a.T;
    const int m = (T + 1) & ~1;
    double *H = sm;             // T*T  assembled H; later the coefficient vectors
    double *L = H + m * m;      // T*T  Cholesky factor (lower)
    double *Cm = L + m * m;     // m*m  standard-form matrix
    double *V = Cm + m * m;     // m*m
    double *rot = V + m * m;    // m
    double *red = rot + m;      // 8
    double *ev = red + 8;       // m
    int *order = reinterpret_cast<int *>(ev + m);                              // m
    // kJwMax x kJwPitch doubles for the single-wave eigensolver (only carved for m <= kJwMax), 16-byte aligned,
    // followed by the refinement's six matrices
    double *Gc = reinterpret_cast<double *>((reinterpret_cast<uintptr_t>(order + m) + 15) & ~(uintptr_t)15);
    double *R6 = Gc + kJwMax * kJwPitch;
    const int tid = threadIdx.x;
    const int64_t P = (int64_t)T * (T + 1) / 2;
    const bool pairs = (a.layout == EVC_LAYOUT_PAIR5 || a.layout == EVC_LAYOUT_PACK2 || a.layout == EVC_LAYOUT_SYM8);
    const int64_t rows2 = pairs ? P : (int64_t)T * T;

    // (1) one-body rows (partials are stored [span][row]: coalesced over rows)
    for (int r = tid; r < T * T; r += kThreads) {
        double s = 0.0;
        for (int k = 0; k < a.nsp1; ++k) s += a.h1part[(int64_t)k * T * T + r];
        H[r] = a.alpha1 * s;
    }
    // S lower triangle -> L
    for (int idx = tid; idx < T * T; idx += kThreads) {
        const int i = idx / T, j = idx - i * T;
        L[idx] = (i >= j) ? a.S[idx] : 0.0;
    }
    __syncthreads();
    // (2) two-body rows, placed as the reference does (evcont.py:41-68)
    for (int64_t r = tid; r < rows2; r += kThreads) {
        // (eight loads in flight per thread: up to 64 spans are summed here instead of in a launch of their own)
        double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0, s4 = 0.0, s5 = 0.0, s6 = 0.0, s7 = 0.0;
        int k = 0;
        for (; k + 8 <= a.nsp2; k += 8) {
            s0 += a.h2part[(int64_t)(k + 0) * rows2 + r];
            s1 += a.h2part[(int64_t)(k + 1) * rows2 + r];
            s2 += a.h2part[(int64_t)(k + 2) * rows2 + r];
            s3 += a.h2part[(int64_t)(k + 3) * rows2 + r];
            s4 += a.h2part[(int64_t)(k + 4) * rows2 + r];
            s5 += a.h2part[(int64_t)(k + 5) * rows2 + r];
            s6 += a.h2part[(int64_t)(k + 6) * rows2 + r];
            s7 += a.h2part[(int64_t)(k + 7) * rows2 + r];
        }
        for (; k < a.nsp2; ++k) s0 += a.h2part[(int64_t)k * rows2 + r];
        const double s = ((s0 + s1) + (s2 + s3)) + ((s4 + s5) + (s6 + s7));
        int ia, ib;
        if (pairs) {
            ia = (int)tri_row(r);
            ib = (int)(r - (int64_t)ia * (ia + 1) / 2);
        } else {
            ia = (int)(r / T);
            ib = (int)(r - (int64_t)ia * T);
        }
        H[ia * T + ib] += a.alpha2 * s;
    }
    __syncthreads();
    if (a.Hout)
        for (int idx = tid; idx < T * T; idx += kThreads) a.Hout[idx] = H[idx];
    const bool fastbase = a.fast && m <= kJwMax;   // workgroup-parallel factorisation (T <= 32)
    EVC_STAMP(31);
    if (fastbase) {
        // (3') Cholesky S = L L^T and B = L^-1 in the registers of one wave (chol_inverse_wave); (4') C = B Hsym B^T as
        //      two row.row products on the workgroup, matrices at pitch kRp in the refinement's buffers (free until the
        //      eigensolver starts).  The left-looking thread-per-row loops of the general path below are chains of
        //      ~T^2/2 dependent LDS reads each.
        double *Bi = R6 + kRsz, *Hs = R6 + 2 * kRsz, *Wt = R6 + 3 * kRsz;
        // S_train is the same for every geometry: B = L^-1 is cached in the workspace next to the matrix it was computed
        // from and reused when that matrix is bit-identical to this call's (uninitialised or stale memory: a miss)
        double sv[4];   // this thread's elements of the overlap matrix (T*T <= 1024)
        int same = a.bcache ? 1 : 0;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int idx = tid + u * kThreads;
            sv[u] = idx < T * T ? L[idx] : 0.0;
            if (a.bcache && idx < T * T) same &= (a.bcache[idx] == sv[u]) ? 1 : 0;
        }
        // (block-wide AND through the reduction scratch; __syncthreads_and would add static LDS to a kernel that asks for
        //  all 160 KB dynamically)
        const bool hit = block_max_nan(same ? 0.0 : 1.0, red) == 0.0;
        for (int idx = tid; idx < kRsz; idx += kThreads) {
            const int i = idx / kRp, j = idx - i * kRp;
            const bool in = i < T && j < T;
            Hs[idx] = in ? (i >= j ? H[i * T + j] : H[j * T + i]) : 0.0;
            Bi[idx] = (hit && in) ? a.bcache[T * T + i * T + j] : 0.0;
        }
        __syncthreads();
        if (!hit) {   // workgroup-uniform
            if (tid < 64) chol_inverse_wave(L, T, Bi);
            __syncthreads();
            if (a.bcache) {
                for (int idx = tid; idx < T * T; idx += kThreads) {
                    const int i = idx / T, j = idx - i * T;
                    a.bcache[T * T + idx] = Bi[i * kRp + j];
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int idx = tid + u * kThreads;
                    if (idx < T * T) a.bcache[idx] = sv[u];
                }
            }
        }
        EVC_STAMP(33);
        // Wt[j][k] = sum_l B[j][l] Hs[k][l];  C[i][j] = sum_k B[i][k] Wt[j][k]  (rows >= T are zero: decoupled dummy)
        mm_rowrow(m, Bi, Hs, [&](int j, int k, double v) { Wt[j * kRp + k] = v; });
        __syncthreads();
        mm_rowrow(m, Bi, Wt, [&](int i, int j, double v) { V[i * m + j] = v; });
        // keep B = L^-1 (pitch m) where the left-looking path keeps L: the back-transformation is c = B^T y
        for (int idx = tid; idx < m * m; idx += kThreads) {
            const int i = idx / m, j = idx - i * m;
            L[idx] = Bi[i * kRp + j];
        }
        __syncthreads();
    } else {
    // (3) Cholesky of S (lower triangle, as dpotrf('L')), left-looking: thread i owns row i and
        //     recomputes the pivot itself, so the column needs no barrier between pivot and scaling.
        for (int j = 0; j < T; ++j) {
            const int i = tid;
            double v = 0.0, d = 0.0;
            if (i >= j && i < T) {
                v = L[i * T + j];
                d = L[j * T + j];
                for (int k = 0; k < j; ++k) {
                    const double ljk = L[j * T + k];
                    v = fma(-L[i * T + k], ljk, v);
                    d = fma(-ljk, ljk, d);
                }
                d = sqrt(d);
            }
            __syncthreads();
            if (i >= j && i < T) L[i * T + j] = (i == j) ? d : v / d;
            __syncthreads();
        }
        // (4) C = L^-1 Hsym L^-T, Hsym from the LOWER triangle of H (dsygst).
        //     thread j solves L z = Hsym[:,j]; result in Cm[:,j]
        if (tid < T) {
            const int j = tid;
            for (int i = 0; i < T; ++i) {
                double v = (i >= j) ? H[i * T + j] : H[j * T + i];
                for (int k = 0; k < i; ++k) v = fma(-L[i * T + k], Cm[k * m + j], v);
                Cm[i * m + j] = v / L[i * T + i];
            }
        }
        __syncthreads();
        //     thread i solves L w = Z[i,:]^T; result is row i of C, kept in V[i,:]
        if (tid < T) {
            const int i = tid;
            for (int j = 0; j < T; ++j) {
                double v = Cm[i * m + j];
                for (int k = 0; k < j; ++k) v = fma(-L[j * T + k], V[i * m + k], v);
                V[i * m + j] = v / L[j * T + j];
            }
        }
        __syncthreads();
    }
    for (int idx = tid; idx < m * m; idx += kThreads) {
        const int i = idx / m, j = idx - i * m;
        double v = 0.0;
        if (i < T && j < T) v = 0.5 * (V[i * m + j] + V[j * m + i]);
        Cm[idx] = v;  // the dummy dimension (odd T) stays decoupled and is skipped below
    }
    __syncthreads();
    EVC_STAMP(34);
    // A few lowest roots (the energy+force path asks for one): double-precision tridiagonal route on ONE wave, verified
    // against the matrix (few_roots.hpp); anything it does not like falls through to the full eigensolver below.
    bool few_ok = false;
    // (a warm-started call skips it: the refinement from the previous call's eigenvectors below is faster still -- H30,
    //  one geometry per step: 4 420 -> 4 900 steps/s; the call after a cold one finds no such vectors, runs the full
    //  eigensolver once and leaves them)
    if (a.few && !(a.warm && a.vstd) && fastbase && a.nroots <= few::kMaxRoots && T >= 2) {
        if (tid < 64) {
            const int why = T <= 8    ? few::few_roots_wave<8>(Cm, m, T, a.nroots, ev, V, T, R6, red + 2)
                            : T <= 16 ? few::few_roots_wave<16>(Cm, m, T, a.nroots, ev, V, T, R6, red + 2)
                            : T <= 24 ? few::few_roots_wave<24>(Cm, m, T, a.nroots, ev, V, T, R6, red + 2)
                                      : few::few_roots_wave<32>(Cm, m, T, a.nroots, ev, V, T, R6, red + 2);
            if (tid == 0) red[0] = why == 0 ? 1.0 : 0.0;
            EVC_DBGVAL(40, why);
            EVC_DBGVAL(41, red[2]);
            EVC_DBGVAL(42, red[3]);
            EVC_DBGVAL(43, red[4]);
            EVC_DBGVAL(44, red[5]);
        }
        __syncthreads();
        few_ok = red[0] != 0.0;
        __syncthreads();
        EVC_STAMP(37);
    }
    // warm start from the standard-form eigenvectors of the previous call (H is free as scratch here)
    if (few_ok) {
        // (ev[r], V[r * T + i]: root r ascending; c = B^T y below)
    } else if (m <= kJwMax) {
        // the standard-form matrix is indefinite: shift it by a Gershgorin bound (the eigenvectors do not change)
        if (tid < m) {
            double rs = 0.0;
            for (int j = 0; j < m; ++j) rs += fabs(Cm[tid * m + j]);
            ev[tid] = rs;
        }
        __syncthreads();
        double shift = 0.0;
        for (int j = 0; j < m; ++j) shift = fmax(shift, ev[j]);
        shift = 2.0 * shift + 1.0e-300;   // eigenvalues of the shifted matrix within [1, 3] x the bound
        __syncthreads();
        if (a.fast) {
            const bool warm = a.warm && a.vstd;
            if (warm) {   // refinement straight from the previous eigenvectors (garbage makes it fall back)
                for (int idx = tid; idx < m * m; idx += kThreads) V[idx] = a.vstd[idx];
                __syncthreads();
            }
            eigh_small(Cm, V, m, T, shift, warm, a.fast, R6, Gc, ev, red);
        } else {
            const bool warm = a.warm && a.vstd && warm_start_rotate(Cm, V, H, T, m, a.vstd, m, red);
            jacobi_eigh_wave(Cm, V, m, shift, !warm, Gc, ev);
        }
    } else {
        const bool warm = a.warm && a.vstd && warm_start_rotate(Cm, V, H, T, m, a.vstd, m, red);
        jacobi_eigh_lds(Cm, V, m, rot, red, !warm);
    }
    if (a.vstd && !few_ok)
        for (int idx = tid; idx < m * m; idx += kThreads) a.vstd[idx] = V[idx];
    EVC_STAMP(35);
    // (5) ascending order
    if (!few_ok) {
        if (tid < T) ev[tid] = Cm[tid * m + tid];
        __syncthreads();
        if (tid < T) {
            int rank = 0;
            const double v = ev[tid];
            for (int j = 0; j < T; ++j) rank += (ev[j] < v || (ev[j] == v && j < tid)) ? 1 : 0;
            order[rank] = tid;
        }
        __syncthreads();
    }
    // (6) back-transform c = L^-T y for the requested roots; store into H region
    if (few_ok) {
        // c_i = sum_{k >= i} B[k][i] y_k, y = row `root` of V (pitch T)
        for (int idx = tid; idx < a.nroots * T; idx += kThreads) {
            const int root = idx / T, i = idx - root * T;
            double c = 0.0;
            for (int k = i; k < T; ++k) c = fma(L[k * m + i], V[root * T + k], c);
            H[idx] = c;
        }
        if (tid < a.nroots) a.evals[tid] = ev[tid] + a.e_shift;
    } else if (fastbase) {
        // c_i = sum_{k >= i} B[k][i] y_k with B = L^-1 kept in `L` (pitch m): one thread per (root, i)
        for (int idx = tid; idx < a.nroots * T; idx += kThreads) {
            const int root = idx / T, i = idx - root * T, col = order[root];
            double c = 0.0;
            for (int k = i; k < T; ++k) c = fma(L[k * m + i], V[k * m + col], c);
            H[idx] = c;
        }
        if (tid < a.nroots) a.evals[tid] = ev[order[tid]] + a.e_shift;
    } else if (tid < a.nroots) {
        const int col = order[tid];
        double *c = H + tid * T;
        for (int i = T - 1; i >= 0; --i) {
            double v = V[i * m + col];
            for (int k = i + 1; k < T; ++k) v = fma(-L[k * T + i], c[k], v);
            c[i] = v / L[i * T + i];
        }
        a.evals[tid] = ev[col] + a.e_shift;
    }
    __syncthreads();
    for (int idx = tid; idx < a.nroots * T; idx += kThreads) a.evecs[idx] = H[idx];
    EVC_STAMP(36);
    // (7) weights of root 0 for the predicted RDMs
    const double *c0 = H;
    if (a.w1)
        for (int idx = tid; idx < T * T; idx += kThreads) {
            const int ia = idx / T;
            const double w = c0[ia] * c0[idx - ia * T];
            a.w1[idx] = w;
            // transposed copy for the batched K8: [row][slot] in the workspace of the group's first geometry
            if (a.w1t) a.w1t[(int64_t)idx * kMaxBatchG + (int)(gblk % kMaxBatchG)] = w;
        }
    if (a.w2) {
        for (int64_t r = tid; r < a.w2_count; r += kThreads) {
            const int64_t g = r + a.w2_offset;
            double w;
            if (pairs) {
                const int ia = (int)tri_row(g), ib = (int)(g - (int64_t)ia * (ia + 1) / 2);
                w = (ia == ib) ? c0[ia] * c0[ia] : 2.0 * c0[ia] * c0[ib];
            } else {
                const int ia = (int)(g / T);
                w = c0[ia] * c0[g - (int64_t)ia * T];
            }
            a.w2[r] = w;
            // transposed copy for the batched K8: [row][slot] in the workspace of the group's first geometry
            if (a.w2t) a.w2t[r * kMaxBatchG + (int)(gblk % kMaxBatchG)] = w;
        }
    }
}

__global__ __launch_bounds__(kThreads) void subspace_kernel(SolveArgs a) { subspace_body(a, blockIdx.x); }

// The subspace solve and the eigendecomposition half of the Loewdin step (part 2: U and s, which the gradient's last
// kernel alone reads) in ONE launch: `count` workgroups each, both one workgroup per geometry and latency-bound, neither
// depending on the other -- the eigensolver (~75 us) then runs beside the subspace solve (~54 us) instead of in front of
// the whole energy phase, with no second stream (batches of 12 and more geometries; smaller calls send it to the side
// stream, pipeline.hip).
__global__ __launch_bounds__(kThreads) void subspace_loewdin_kernel(SolveArgs sa, LoewdinArgs la, int count) {
    if ((int)blockIdx.x < count) subspace_body(sa, blockIdx.x);
    else loewdin_body(la, (int64_t)blockIdx.x - count);
}

int launch_subspace_loewdin(const SolveArgs &s_in, const LoewdinArgs &l_in, int count, hipStream_t st) {
    SolveArgs a = s_in;
    LoewdinArgs l = l_in;
    a.fast = l.fast = eigh_fast_enabled();
    static const int few_on = getenv("EVC_SUBSPACE_FEW") ? atoi(getenv("EVC_SUBSPACE_FEW")) : 1;
    a.few = few_on;
    l.part = 2;
    if (a.T > kSubspaceSmallT || l.n > kJwMax || !l.fast) {
        set_error("subspace + Loewdin in one launch: T=%d, n=%d outside the small-kernel range", a.T, l.n);
        return -1;
    }
    const int m = (a.T + 1) & ~1, ml = (l.n + 1) & ~1;
    const size_t lds_s = sizeof(double) * (size_t)4 * m * m + sizeof(int) * m + jacobi_aux_bytes(m);
    const size_t lds_l = sizeof(double) * (size_t)3 * ml * ml + jacobi_aux_bytes(ml);
    static LdsAttr attr;
    if (int rc = allow_dynamic_lds(subspace_loewdin_kernel, attr, 160 * 1024, "subspace_loewdin")) return rc;
    hipLaunchKernelGGL(subspace_loewdin_kernel, dim3(2 * count), dim3(kThreads), lds_s > lds_l ? lds_s : lds_l, st, a, l,
                       count);
    EVC_LAUNCH_CHECK("subspace_loewdin");
    return 0;
}

int launch_subspace_solve(const SolveArgs &a_in, int count, hipStream_t st) {
    // (measured: 0.26 / 0.42 / 0.61 ms at T = 33 / 48 / 64 against 0.62 / 1.14 / 1.89 ms for the two-sided LDS Jacobi
    //  this file used up to T = 64 in round 2)
    if (a_in.T > kSubspaceSmallT) return launch_subspace_big(a_in, count, st);
    SolveArgs a = a_in;
    a.fast = eigh_fast_enabled();
    static const int few_on = getenv("EVC_SUBSPACE_FEW") ? atoi(getenv("EVC_SUBSPACE_FEW")) : 1;
    a.few = few_on;
    const int m = (a.T + 1) & ~1;
    const size_t lds = sizeof(double) * (size_t)4 * m * m + sizeof(int) * m + jacobi_aux_bytes(m);
    static LdsAttr attr;
    if (int rc = allow_dynamic_lds(subspace_kernel, attr, 160 * 1024, "subspace_solve")) return rc;
    hipLaunchKernelGGL(subspace_kernel, dim3(count), dim3(kThreads), lds, st, a);
    EVC_LAUNCH_CHECK("subspace_solve");
    return 0;
}

// Row weights from a coefficient vector the caller supplies (the non-Hermitian branch: the T x T pencil is solved
// with scipy.linalg.eig on the host, as the reference does, and its eigenvector comes back here).
__global__ __launch_bounds__(256) void pair_weights_kernel(const double *__restrict__ c, int T, int pairs,
                                                           double *__restrict__ w1, double *__restrict__ w2,
                                                           int64_t w2_offset, int64_t w2_count) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (w1 && i < (int64_t)T * T) {
        const int ia = (int)(i / T);
        w1[i] = c[ia] * c[i - (int64_t)ia * T];
    }
    if (w2 && i < w2_count) {
        const int64_t g = i + w2_offset;
        double w;
        if (pairs) {
            const int ia = (int)tri_row(g), ib = (int)(g - (int64_t)ia * (ia + 1) / 2);
            w = (ia == ib) ? c[ia] * c[ia] : 2.0 * c[ia] * c[ib];
        } else {
            const int ia = (int)(g / T);
            w = c[ia] * c[g - (int64_t)ia * T];
        }
        w2[i] = w;
    }
}

int launch_pair_weights(const double *c, int T, int layout, double *w1, double *w2, int64_t w2_offset,
                        int64_t w2_count, hipStream_t st) {
    const int pairs = (layout == EVC_LAYOUT_PAIR5 || layout == EVC_LAYOUT_PACK2 || layout == EVC_LAYOUT_SYM8) ? 1 : 0;
    const int64_t nmax = (int64_t)T * T > w2_count ? (int64_t)T * T : w2_count;
    hipLaunchKernelGGL(pair_weights_kernel, dim3((unsigned)ceil_div(nmax, 256)), dim3(256), 0, st, c, T, pairs, w1, w2,
                       w2_offset, w2_count);
    EVC_LAUNCH_CHECK("pair_weights");
    return 0;
}

// ------------------------------------------------------------------ gradient prep
// Pao = X D X^T ; Y1 = scale1 * hcore X (D + D^T)
__device__ __forceinline__ void grad_prep_body(GradPrepArgs a, int64_t g) {
    extern __shared__ __align__(16) double sm[];
    const int n = a.n;
    {
        a.X += g * a.sws;
        a.hcore += g * a.sh;
        a.D += g * a.sD;
        a.Pao += g * a.sws;
        a.Y1 += g * a.sws;
    }
    if (n <= 32) {
        // row.row products at pitch kRp (16-byte LDS loads, no bank conflicts): X symmetric, so
        //   T1t[j][i] = (X D)[i][j] = sum_k Dt[j][k] X[i][k],  Pao[i][j] = sum_k T1[i][k] X[j][k]   (T1 = T1t^T stored both ways)
        //   T2t[j][i] = (X (D + D^T))[i][j] = sum_k Dsym[j][k] X[i][k],  Y1[i][j] = scale1 sum_k h[i][k] T2t[j][k]
        double *Xp = sm, *Dt = Xp + kRsz, *Ds2 = Dt + kRsz, *Hp = Ds2 + kRsz, *T1 = Hp + kRsz, *T2t = T1 + kRsz;
        const int m = (n + 1) & ~1;
        for (int idx = threadIdx.x; idx < kRsz; idx += kThreads) {
            const int i = idx / kRp, j = idx - i * kRp;
            const bool in = i < n && j < n;
            const double dij = in ? a.D[i * n + j] : 0.0, dji = in ? a.D[j * n + i] : 0.0;
            Xp[idx] = in ? a.X[i * n + j] : 0.0;
            Dt[idx] = dji;
            Ds2[idx] = dij + dji;
            Hp[idx] = in ? a.hcore[i * n + j] : 0.0;
            T1[idx] = 0.0;
            T2t[idx] = 0.0;
        }
        __syncthreads();
        // T1[i][j] = (X D)[i][j] = sum_k X[i][k] Dt[j][k];  T2t[j][i] = sum_k Dsym[j][k] X[i][k]
        mm_rowrow(m, Xp, Dt, [&](int i, int j, double v) { T1[i * kRp + j] = v; });
        mm_rowrow(m, Ds2, Xp, [&](int j, int i, double v) { T2t[j * kRp + i] = v; });
        __syncthreads();
        mm_rowrow(m, T1, Xp, [&](int i, int j, double v) {
            if (i < n && j < n) a.Pao[i * n + j] = v;
        });
        mm_rowrow(m, Hp, T2t, [&](int i, int j, double v) {
            if (i < n && j < n) a.Y1[i * n + j] = a.scale1 * v;
        });
        return;
    }
    if (n > 64) {
        // four n x n matrices no longer fit LDS: one product buffer, the operands through the caches
        double *Ts = sm;   // n*n
        mm16(n, [&](int i, int k) { return a.X[i * n + k]; }, [&](int k, int j) { return a.D[k * n + j]; },
             [&](int i, int j, double v) { Ts[i * n + j] = v; });
        __syncthreads();
        mm16(n, [&](int i, int k) { return Ts[i * n + k]; }, [&](int k, int j) { return a.X[j * n + k]; },
             [&](int i, int j, double v) { a.Pao[i * n + j] = v; });
        __syncthreads();
        mm16(n, [&](int i, int k) { return a.X[i * n + k]; },
             [&](int k, int j) { return a.D[k * n + j] + a.D[j * n + k]; },
             [&](int i, int j, double v) { Ts[i * n + j] = v; });
        __syncthreads();
        mm16(n, [&](int i, int k) { return a.hcore[i * n + k]; }, [&](int k, int j) { return Ts[k * n + j]; },
             [&](int i, int j, double v) { a.Y1[i * n + j] = a.scale1 * v; });
        return;
    }
    double *Xs = sm;            // n*n
    double *Ds = Xs + n * n;    // n*n
    double *Hs = Ds + n * n;    // n*n
    double *Ts = Hs + n * n;    // n*n
    copy_to_lds(Xs, a.X, n * n);
    copy_to_lds(Ds, a.D, n * n);
    copy_to_lds(Hs, a.hcore, n * n);
    __syncthreads();
    mm16(n, [&](int i, int k) { return Xs[i * n + k]; }, [&](int k, int j) { return Ds[k * n + j]; },
         [&](int i, int j, double v) { Ts[i * n + j] = v; });
    __syncthreads();
    mm16(n, [&](int i, int k) { return Ts[i * n + k]; }, [&](int k, int j) { return Xs[j * n + k]; },
         [&](int i, int j, double v) { a.Pao[i * n + j] = v; });
    __syncthreads();
    mm16(n, [&](int i, int k) { return Xs[i * n + k]; },
         [&](int k, int j) { return Ds[k * n + j] + Ds[j * n + k]; },
         [&](int i, int j, double v) { Ts[i * n + j] = v; });
    __syncthreads();
    mm16(n, [&](int i, int k) { return Hs[i * n + k]; }, [&](int k, int j) { return Ts[k * n + j]; },
         [&](int i, int j, double v) { a.Y1[i * n + j] = a.scale1 * v; });
}

__global__ __launch_bounds__(kThreads) void grad_prep_kernel(GradPrepArgs a) { grad_prep_body(a, blockIdx.x); }

// The same launch ALSO unpacks the packed predicted 2-RDM of the compressed layout into the dense symmetric (pair, pair)
// matrix SB (pack.hip unpack8_pairs_kernel: SB[u][v] = 4 p[tri(max, min)], one wave per row u, four rows per workgroup):
// both only need what K8 has just written, so the `count` workgroups of the one and the count * ceil(npairs / 4)
// workgroups of the other share a launch instead of following each other (one kernel boundary and the shorter of the
// two durations less on the critical path of a step).  Blocks [0, count): grad_prep; the rest: unpack.
__global__ __launch_bounds__(kThreads) void unpack8_prep_kernel(GradPrepArgs a, const double *__restrict__ p, int64_t sp,
                                                                double *__restrict__ SB, int64_t sws, int count, int ld) {
    if ((int)blockIdx.x < count) {
        grad_prep_body(a, blockIdx.x);
        return;
    }
    const int n = a.n, npairs = n * (n + 1) / 2, bpg = (npairs + 3) / 4;
    const int b = (int)blockIdx.x - count;
    // (blocks of eight consecutive geometries interleaved: each XCD works through one geometry's packed vector at a time)
    const int nx = count & ~7;
    int geom, blk;
    if (b < nx * bpg) {
        const int xcd = b & 7, slot = b >> 3;
        geom = (slot / bpg) * 8 + xcd;
        blk = slot % bpg;
    } else {
        const int r = b - nx * bpg;
        geom = nx + r / bpg;
        blk = r % bpg;
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int u = blk * 4 + wave;
    if (u >= npairs) return;
    p += (int64_t)geom * sp;
    double *sb = SB + (int64_t)geom * sws + (int64_t)u * ld;
    for (int v = lane; v < npairs; v += 64) sb[v] = 4.0 * (u >= v ? p[tri_index(u, v)] : p[tri_index(v, u)]);
}

static size_t grad_prep_lds(int n) {
    return n <= 32 ? sizeof(double) * (size_t)6 * kRsz : sizeof(double) * (size_t)(n > 64 ? 1 : 4) * n * n;
}

int launch_unpack8_prep(const GradPrepArgs &a, const double *packed, int64_t sp, double *SB, int64_t sws, int count,
                        hipStream_t st) {
    const int npairs = a.n * (a.n + 1) / 2, bpg = (npairs + 3) / 4;
    static LdsAttr attr;
    if (int rc = allow_dynamic_lds(unpack8_prep_kernel, attr, 160 * 1024, "unpack8_prep")) return rc;
    hipLaunchKernelGGL(unpack8_prep_kernel, dim3((unsigned)(count + bpg * count)), dim3(kThreads), grad_prep_lds(a.n), st, a,
                       packed, sp, SB, sws, count, pair_ld(a.n));
    EVC_LAUNCH_CHECK("unpack8_prep");
    return 0;
}

int launch_grad_prep(const GradPrepArgs &a, int count, hipStream_t st) {
    const size_t lds = grad_prep_lds(a.n);
    static LdsAttr attr;
    if (int rc = allow_dynamic_lds(grad_prep_kernel, attr, 160 * 1024, "grad_prep")) return rc;
    hipLaunchKernelGGL(grad_prep_kernel, dim3(count), dim3(kThreads), lds, st, a);
    EVC_LAUNCH_CHECK("grad_prep");
    return 0;
}

// ------------------------------------------------------------------ gradient finalisation
// dE = <dX, Y> + explicit terms, with dX[A,x] = U [ (U^T dS[A,x] U) o F ] U^T (Daleckii-Krein form
// of gradients_loewdin.py:41-134).  Taking the adjoint once,  <dX,Y> = <dS, W>,
// W = U [ F o (U^T Y U) ] U^T, removes the (N,N,A,3) tensor altogether.
template <int NMAX>   // 32: n <= 32 (and n > 64, where nothing is staged); 64: 32 < n <= 64
__global__ __launch_bounds__(kThreads) void grad_final_kernel(GradFinalArgs a) {
    extern __shared__ __align__(16) double sm[];
    const int n = a.n;
    {
        const int64_t g = blockIdx.x;
        a.U += g * a.sws;
        a.s += g * a.sws;
        a.Y1 += g * a.sws;
        a.y2 += g * a.sws;
        a.t2part += g * a.sws;
        a.term3 += g * a.sws;
        a.ipovlp += g * a.sip;
        if (a.gnuc) a.gnuc += g * a.sgn;
        a.grad += g * a.sgrad;
    }
    // n > 64: four n x n matrices do not fit LDS -- Y and U are then read through the caches (`wide`), Q and W stay
    const bool wide = n > 64;
    double *Y = sm;                          // n*n   (wide: unused, zero-sized)
    double *Q = wide ? sm : Y + n * n;       // n*n
    double *W = Q + n * n;                   // n*n
    double *Us = wide ? W : W + n * n;       // n*n   (wide: unused, zero-sized)
    double *rs = (wide ? W : Us) + n * n;    // n   sqrt(s) (0 where guarded)
    double *fs = rs + n;       // n   f(s)
    double *ss = fs + n;       // n   s
    double *t2 = ss + n;       // 3*n
    double *add = t2 + 3 * n;  // 3*natm: scale1 * (term3 + gnuc)
    int *sl = reinterpret_cast<int *>(add + 3 * a.natm);   // 2*natm: AO slices
    const int tid = threadIdx.x;
    // n <= 64: the 3 n^2 overlap derivatives are fetched into registers now and parked in the three product
    // buffers once those are free, so that the per-atom loop at the end runs out of LDS (it is a chain of
    // dependent global loads otherwise: ~2 us per (atom, x) and wave)
    const bool stage_ip = !wide;          // (n <= 64: the three product buffers exist)
    constexpr int kIpf = (3 * NMAX * NMAX + kThreads - 1) / kThreads;   // 12 values per thread at n = 32, 48 at n = 64
    double ipf[kIpf];
    if (stage_ip) {
#pragma unroll
        for (int u = 0; u < kIpf; ++u)
            if (kThreads * u < 3 * n * n) {   // uniform
                const int idx = tid + kThreads * u;
                ipf[u] = idx < 3 * n * n ? a.ipovlp[idx] : 0.0;
            }
    }
    for (int idx = tid; idx < 2 * a.natm; idx += kThreads) sl[idx] = (int)a.aoslices[idx];
    for (int idx = tid; idx < 3 * a.natm; idx += kThreads) {
        double g = 0.0;
        if (a.scale1 != 0.0) {
            g = a.scale1 * a.term3[idx];
            if (a.gnuc) g += a.scale1 * a.gnuc[idx];
        }
        add[idx] = g;
    }
    if (!wide) {
        copy_to_lds(Us, a.U, n * n);
        for (int idx = tid; idx < n * n; idx += kThreads) {
            const int ai = idx / n, i = idx - ai * n;  // Y[a][i]; y2 is stored [i][a]
            Y[idx] = a.Y1[idx] + 0.5 * a.y2[i * n + ai];
        }
    }
    auto Uv = [&](int i, int j) { return wide ? a.U[i * n + j] : Us[i * n + j]; };
    auto Yv = [&](int k, int j) { return wide ? a.Y1[k * n + j] + 0.5 * a.y2[j * n + k] : Y[k * n + j]; };
    if (tid < n) {
        const double s = a.s[tid];
        const bool ok = s > 1.0e-15;
        ss[tid] = s;
        rs[tid] = ok ? sqrt(s) : 0.0;
        fs[tid] = ok ? 1.0 / sqrt(s) : 0.0;
    }
    for (int idx = tid; idx < 3 * n; idx += kThreads) {
        const int m_ = idx / 3, x = idx - 3 * m_;
        // (eight loads in flight: the partials of one (m, x) are a chain of nchunk >= n dependent round trips otherwise)
        const double *p = a.t2part + ((int64_t)m_ * 3 + x) * a.nchunk;
        double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0, s4 = 0.0, s5 = 0.0, s6 = 0.0, s7 = 0.0;
        int ch = 0;
        for (; ch + 8 <= a.nchunk; ch += 8) {
            s0 += p[ch];
            s1 += p[ch + 1];
            s2 += p[ch + 2];
            s3 += p[ch + 3];
            s4 += p[ch + 4];
            s5 += p[ch + 5];
            s6 += p[ch + 6];
            s7 += p[ch + 7];
        }
        for (; ch < a.nchunk; ++ch) s0 += p[ch];
        t2[x * n + m_] = ((s0 + s1) + (s2 + s3)) + ((s4 + s5) + (s6 + s7));
    }
    __syncthreads();
    // Q = U^T Y
    mm16(n, [&](int i, int k) { return Uv(k, i); }, [&](int k, int j) { return Yv(k, j); },
         [&](int i, int j, double v) { Q[i * n + j] = v; });
    __syncthreads();
    // W = (Q U) o F
    mm16(n, [&](int i, int k) { return Q[i * n + k]; }, [&](int k, int j) { return Uv(k, j); },
         [&](int i, int j, double v) {
             double F;
             if (rs[i] > 0.0 && rs[j] > 0.0) F = -1.0 / (rs[i] * rs[j] * (rs[i] + rs[j]));
             else if (ss[i] != ss[j]) F = (fs[i] - fs[j]) / (ss[i] - ss[j]);
             else F = 0.0;
             W[i * n + j] = v * F;
         });
    __syncthreads();
    // Q = U W
    mm16(n, [&](int i, int k) { return Uv(i, k); }, [&](int k, int j) { return W[k * n + j]; },
         [&](int i, int j, double v) { Q[i * n + j] = v; });
    __syncthreads();
    // W = Q U^T
    mm16(n, [&](int i, int k) { return Q[i * n + k]; }, [&](int k, int j) { return Uv(j, k); },
         [&](int i, int j, double v) { W[i * n + j] = v; });
    __syncthreads();
    // grad[A,x] = - sum_{mu in A} sum_nu ip[x,mu,nu] (W[mu,nu] + W[nu,mu])
    //             - 1/2 sum_{m in A} t2[x][m] + scale1 * (term3 + gnuc)
    if (stage_ip) {  // Y, Q, Us are free now: ip[x] -> {Y, Q, Us}[x]
#pragma unroll
        for (int u = 0; u < kIpf; ++u) {
            const int idx = tid + kThreads * u;
            if (kThreads * u < 3 * n * n && idx < 3 * n * n) {
                const int x = idx / (n * n);
                (x == 0 ? Y : x == 1 ? Q : Us)[idx - x * n * n] = ipf[u];
            }
        }
        __syncthreads();
    }
    // two short steps instead of one wave-wide reduction per (atom, x): q[x][mu] = sum_nu ip[x,mu,nu] (W + W^T)[mu,nu]
    // + t2[x][mu] / 2 by one thread per (x, mu) (sum_A (p1 - p0) = n: 3 n dots of length n in all), then one thread
    // per (atom, x) adds up its AOs
    for (int idx = tid; idx < 3 * n; idx += kThreads) {
        const int x = idx / n, mu = idx - x * n;
        const double *ipx = x == 0 ? Y : x == 1 ? Q : Us;
        double s0 = 0.0, s1 = 0.0;
        int nu = 0;
        for (; nu + 2 <= n; nu += 2) {
            const double i0 = stage_ip ? ipx[mu * n + nu] : a.ipovlp[(x * n + mu) * n + nu];
            const double i1 = stage_ip ? ipx[mu * n + nu + 1] : a.ipovlp[(x * n + mu) * n + nu + 1];
            s0 = fma(i0, W[mu * n + nu] + W[nu * n + mu], s0);
            s1 = fma(i1, W[mu * n + nu + 1] + W[(nu + 1) * n + mu], s1);
        }
        if (nu < n) {
            const double i0 = stage_ip ? ipx[mu * n + nu] : a.ipovlp[(x * n + mu) * n + nu];
            s0 = fma(i0, W[mu * n + nu] + W[nu * n + mu], s0);
        }
        t2[idx] = (s0 + s1) + 0.5 * t2[idx];
    }
    __syncthreads();
    for (int ax = tid; ax < a.natm * 3; ax += kThreads) {
        const int A = ax / 3, x = ax - 3 * A;
        double s = 0.0;
        for (int mu = sl[2 * A]; mu < sl[2 * A + 1]; ++mu) s += t2[x * n + mu];
        a.grad[ax] = add[ax] - s;
    }
}

int launch_grad_final(const GradFinalArgs &a, int count, hipStream_t st) {
    const size_t lds = sizeof(double) * ((size_t)(a.n > 64 ? 2 : 4) * a.n * a.n + 6 * a.n + 3 * (size_t)a.natm) +
                       sizeof(int) * 2 * (size_t)a.natm + 16;
    if (a.n > 32 && a.n <= 64) {
        static LdsAttr attr;
        if (int rc = allow_dynamic_lds(grad_final_kernel<64>, attr, 160 * 1024, "grad_final")) return rc;
        hipLaunchKernelGGL(grad_final_kernel<64>, dim3(count), dim3(kThreads), lds, st, a);
    } else {
        static LdsAttr attr;
        if (int rc = allow_dynamic_lds(grad_final_kernel<32>, attr, 160 * 1024, "grad_final")) return rc;
        hipLaunchKernelGGL(grad_final_kernel<32>, dim3(count), dim3(kThreads), lds, st, a);
    }
    EVC_LAUNCH_CHECK("grad_final");
    return 0;
}

}  // namespace evc

// ------------------------------------------------------------------ C ABI
using namespace evc;

extern "C" int evc_loewdin(const double *S, const double *hcore, int n, double *X, double *U, double *s,
                           double *h1, void *stream) {
    EVC_REQUIRE(S && X && U && s, "evc_loewdin: null pointer");
    EVC_REQUIRE(n >= 1 && n <= 80, "evc_loewdin: n=%d out of range 1..80", n);
    EVC_REQUIRE((hcore == nullptr) == (h1 == nullptr), "evc_loewdin: hcore and h1 must both be given or both NULL");
    LoewdinArgs a{};
    a.S = S;
    a.h = hcore;
    a.X = X;
    a.U = U;
    a.s = s;
    a.h1 = h1;
    a.n = n;
    return launch_loewdin(a, 1, as_stream(stream));
}

#ifdef EVC_DEBUG_STAMPS
// Debug: stamps / values written by workgroup 0 of the last eigen-kernel (timing experiments).
extern "C" int evc_debug_read(long long *stamps, double *vals, int n) {
    if (n > 64) n = 64;
    hipError_t e = hipMemcpyFromSymbol(stamps, HIP_SYMBOL(evc::g_dbg_stamp), sizeof(long long) * n);
    if (e == hipSuccess) e = hipMemcpyFromSymbol(vals, HIP_SYMBOL(evc::g_dbg_val), sizeof(double) * n);
    return (int)e;
}
#endif
