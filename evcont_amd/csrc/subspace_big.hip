// K6 for LARGE training sets (T > 32): the subspace generalised eigenproblem H c = E S c of
// ab_initio_eigenvector_continuation.py:73-88 / :157-173 with LAPACK dsygvd semantics (lower triangles,
// Cholesky of S, c^T S c = 1) and the row weights of ab_initio_gradients_loewdin.py:343-356, for training sets the
// register / 32 x 32-tile kernel of dense_small.hip cannot hold.  The reference puts no bound on T (its Zundel
// learning curve evaluates 80 and 100 training states, scripts/MD/Zundel_thermodynamics/continuation/
// 05_Zundel_test_potential_energy.py:182-210; converge_EVCont_MD grows T without bound, MD_utils.py:128-502).
//
// One workgroup of 1024 threads (16 waves) per problem.  ONE T x T matrix lives in LDS at a time (128 KB at T = 128):
//   (a) [only when the cached factor does not match] L = chol(S) right-looking in LDS, inverted in place, B = L^-1
//       written to global memory (the per-workspace cache: S_train does not depend on the geometry);
//   (b) H assembled from the span partials in LDS, symmetrised from its lower triangle;
//   (c) W = B Hs and C = W B^T as "row . row" products on the FP64 matrix cores (16 x 16 tiles dealt to the 16 waves,
//       operands read with 16-byte loads from L2-resident global scratch / LDS);
//   (d) nroots <= 4 (the energy + force path asks for ONE root): Householder tridiagonalisation in LDS, Sturm multisection,
//       twisted factorisation, back-transformation through the reflectors, and a residual / orthogonality check in the
//       original matrix (big_few_roots); otherwise, or when that check fails: eigen-decomposition of C + sigma I (sigma:
//       Gershgorin bound, makes it positive definite) by one-sided (Hestenes) Jacobi on its columns in LDS: 16 lanes per
//       column pair, 16-byte conflict-free LDS rows, the three dot products reduced over a DPP row, one barrier per
//       round-robin step; at convergence column j is (lambda_j + sigma) v_j, so no eigenvector matrix is carried through
//       the rotations;
//   (e) ascending order, back-transformation c = B^T y, weights.
// With EVC_FLAG_WARM_START the Jacobi sweeps start from G0 = (C + sigma I) V_prev (V_prev checked for orthonormality).
// The same code runs with the matrix in global memory (template parameter) for T beyond what LDS holds (Jacobi route).
// Also here: the Loewdin orthogonalisation for 32 < n <= 96 on the same Jacobi (loewdin_big_kernel).
#include <stdlib.h>

#include "common.hpp"
#include "kernels.hpp"

namespace evc {

constexpr int kBT = 1024;   // threads of the workgroup
constexpr int kBW = kBT / 64;
typedef double d4b __attribute__((ext_vector_type(4)));

// Phase stamps of workgroup 0 (tools/micro/subspace_time.py --stamps; library built with EVC_DEBUG_STAMPS=1)
#ifdef EVC_DEBUG_STAMPS
__device__ long long g_big_stamp[16];
__device__ double g_big_val[16];
#define EVC_BSTAMP(i_)                                                                  \
    do {                                                                                \
        if (blockIdx.x == 0 && threadIdx.x == 0) g_big_stamp[i_] = wall_clock64();      \
    } while (0)
#define EVC_BVAL(i_, v_)                                                                \
    do {                                                                                \
        if (blockIdx.x == 0 && threadIdx.x == 0) g_big_val[i_] = (double)(v_);          \
    } while (0)
#else
#define EVC_BSTAMP(i_) do { } while (0)
#define EVC_BVAL(i_, v_) do { } while (0)
#endif

__device__ __forceinline__ double sum8(double v) {    // over the 8 lanes of a half DPP row, every lane gets the total
    v += dpp_move<0xB1>(v);
    v += dpp_move<0x4E>(v);
    v += dpp_move<0x141>(v);
    return v;
}
__device__ __forceinline__ double sum16(double v) {   // over the 16 lanes of a DPP row
    v = sum8(v);
    v += dpp_move<0x140>(v);
    return v;
}

// block-wide reductions over the 16 waves (red: kBW doubles of LDS); all threads get the result
__device__ __forceinline__ double big_block_sum(double v, double *red) {
    v = wave_sum(v);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    double t = 0.0;
#pragma unroll
    for (int w = 0; w < kBW; ++w) t += red[w];
    __syncthreads();
    return t;
}
__device__ __forceinline__ double big_nanmax(double a, double b) { return (a > b || a != a) ? a : b; }
__device__ __forceinline__ double big_block_max(double v, double *red) {   // NaN-propagating
    v = big_nanmax(v, dpp_move<0xB1>(v));
    v = big_nanmax(v, dpp_move<0x4E>(v));
    v = big_nanmax(v, dpp_move<0x141>(v));
    v = big_nanmax(v, dpp_move<0x140>(v));
    v = big_nanmax(big_nanmax(readlane_f64(v, 0), readlane_f64(v, 16)), big_nanmax(readlane_f64(v, 32), readlane_f64(v, 48)));
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    double t = red[0];
#pragma unroll
    for (int w = 1; w < kBW; ++w) t = big_nanmax(t, red[w]);
    __syncthreads();
    return t;
}
__device__ __forceinline__ bool big_block_any(bool b, double *red) {
    const unsigned long long m = __ballot(b);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m ? 1.0 : 0.0;
    __syncthreads();
    double t = 0.0;
#pragma unroll
    for (int w = 0; w < kBW; ++w) t += red[w];
    __syncthreads();
    return t != 0.0;
}

// D[i][j] = sum_k P[i][k] Q[j][k], i, j, k < Tp (Tp a multiple of 16; pitches even, rows 16-byte aligned, padding zero).
// v_mfma_f64_16x16x4_f64: A[i][k]: lane l holds i = l & 15, k slot l >> 4; B the same with j; D[i][j]: j = l & 15,
// i = (l >> 4) + 4 reg.  A lane reads the two adjacent columns 8 kk + 2 (l >> 4), +1 of "its" row with one 16-byte load
// and feeds .x / .y to two MFMAs (the K slot of a lane may be any column as long as A and B agree).
template <typename Store>
__device__ __forceinline__ void big_mm_rr(int Tp, const double *__restrict__ P, int ldp, const double *__restrict__ Q,
                                          int ldq, Store store) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, l15 = lane & 15, l4 = lane >> 4;
    const int nt = Tp >> 4;
    for (int t = wave; t < nt * nt; t += kBW) {
        const int ti = t / nt, tj = t - ti * nt;
        const double *pr = P + (size_t)(16 * ti + l15) * ldp + 2 * l4;
        const double *qr = Q + (size_t)(16 * tj + l15) * ldq + 2 * l4;
        d4b acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll 4
        for (int k = 0; k < Tp; k += 8) {
            const double2 av = *reinterpret_cast<const double2 *>(pr + k);
            const double2 bv = *reinterpret_cast<const double2 *>(qr + k);
            acc = __builtin_amdgcn_mfma_f64_16x16x4f64(av.x, bv.x, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f64_16x16x4f64(av.y, bv.y, acc, 0, 0, 0);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) store(16 * ti + l4 + 4 * r, 16 * tj + l15, acc[r]);
    }
}

__device__ __forceinline__ void big_pair_of(int step, int k, int m, int &p, int &q) {   // round-robin tournament
    const int w = m - 1;
    p = step + k;
    if (p >= w) p -= w;
    if (k == 0) p = w;
    q = step + w - k;
    if (q >= w) q -= w;
}

// One-sided Jacobi on the m columns (m even) of G, stored column-major with pitch Pj (a multiple of 32; rows >= the
// matrix size are zero).  Pair k of a step belongs to the 16 lanes [16 k, 16 k + 16): lane s owns rows 2 s, 2 s + 1
// (+ 32 u) -- the 16 lanes of a pair read 256 contiguous bytes.  Convergence: every pair orthogonal to 1e-9 BEFORE its
// rotation in a sweep (the rotations of that sweep then leave ~1e-18).
__device__ __forceinline__ void big_jacobi(double *G, int m, int Pj, double *red) {
    const int tid = threadIdx.x, s = tid & 15;
    const int half = m >> 1, nu = Pj >> 5;
    const int npass = (half + (kBT / 16) - 1) / (kBT / 16);   // 1 for m <= 128
    for (int sweep = 0; sweep < 60; ++sweep) {
        bool bad = false;
        for (int step = 0; step < m - 1; ++step) {
            for (int pass = 0; pass < npass; ++pass) {
                const int k = (tid >> 4) + pass * (kBT / 16);
                if (k < half) {   // uniform over the DPP row
                    int p, q;
                    big_pair_of(step, k, m, p, q);
                    double *gp = G + (size_t)p * Pj + 2 * s, *gq = G + (size_t)q * Pj + 2 * s;
                    double al = 0.0, be = 0.0, ga = 0.0;
                    double2 x[4], y[4];   // the whole share of this lane when Pj <= 128 (the LDS-resident case)
#pragma unroll
                    for (int u = 0; u < 4; ++u) x[u] = y[u] = make_double2(0.0, 0.0);
                    for (int u0 = 0; u0 < nu; u0 += 4) {
#pragma unroll
                        for (int u = 0; u < 4; ++u)
                            if (u0 + u < nu) {
                                x[u] = *reinterpret_cast<const double2 *>(gp + 32 * (u0 + u));
                                y[u] = *reinterpret_cast<const double2 *>(gq + 32 * (u0 + u));
                            }
#pragma unroll
                        for (int u = 0; u < 4; ++u)
                            if (u0 + u < nu) {
                                al = fma(x[u].x, x[u].x, fma(x[u].y, x[u].y, al));
                                be = fma(y[u].x, y[u].x, fma(y[u].y, y[u].y, be));
                                ga = fma(x[u].x, y[u].x, fma(x[u].y, y[u].y, ga));
                            }
                    }
                    al = sum16(al);
                    be = sum16(be);
                    ga = sum16(ga);
                    const double ab = al * be, g2 = ga * ga;
                    const bool rot = g2 > 1.0e-30 * ab;
                    bad = bad || (g2 > 1.0e-18 * ab);
                    if (rot) {   // uniform over the row
                        // t = sgn(d) b / (|d| + sqrt(d^2 + b^2)), d = beta - alpha, b = 2 gamma (the smaller root); the
                        // angle only has to make g_p . g_q small, c is refined to full precision (c^2 + s^2 = 1)
                        const double d = be - al, b = 2.0 * ga;
                        const double h2 = fma(d, d, b * b);
                        double yv = __builtin_amdgcn_rsq(h2);
                        yv = yv * fma(-0.5 * h2 * yv, yv, 1.5);
                        const double den = fabs(d) + h2 * yv;
                        double r = __builtin_amdgcn_rcp(den);
                        r = r * fma(-den, r, 2.0);
                        const double t = copysign(b, d * b == 0.0 ? b : d * b) * r;
                        const double xx = fma(t, t, 1.0);
                        double z = __builtin_amdgcn_rsq(xx);
                        z = z * fma(-0.5 * xx * z, z, 1.5);
                        z = z * fma(-0.5 * xx * z, z, 1.5);
                        z = z * fma(-0.5 * xx * z, z, 1.5);
                        const double c = z, sn = t * z;
                        if (nu <= 4) {   // uniform: rotate the registers
#pragma unroll
                            for (int u = 0; u < 4; ++u)
                                if (u < nu) {
                                    double2 a2, b2;
                                    a2.x = c * x[u].x - sn * y[u].x;
                                    a2.y = c * x[u].y - sn * y[u].y;
                                    b2.x = sn * x[u].x + c * y[u].x;
                                    b2.y = sn * x[u].y + c * y[u].y;
                                    *reinterpret_cast<double2 *>(gp + 32 * u) = a2;
                                    *reinterpret_cast<double2 *>(gq + 32 * u) = b2;
                                }
                        } else {
                            for (int u = 0; u < nu; ++u) {
                                const double2 xv = *reinterpret_cast<const double2 *>(gp + 32 * u);
                                const double2 yw = *reinterpret_cast<const double2 *>(gq + 32 * u);
                                double2 a2, b2;
                                a2.x = c * xv.x - sn * yw.x;
                                a2.y = c * xv.y - sn * yw.y;
                                b2.x = sn * xv.x + c * yw.x;
                                b2.y = sn * xv.y + c * yw.y;
                                *reinterpret_cast<double2 *>(gp + 32 * u) = a2;
                                *reinterpret_cast<double2 *>(gq + 32 * u) = b2;
                            }
                        }
                    }
                }
            }
            __syncthreads();
        }
        EVC_BVAL(0, sweep + 1);
        if (!big_block_any(bad, red)) break;
    }
}

// ------------------------------------------------------------------ a FEW lowest eigenpairs (nroots <= kFewRoots)
// The energy + force path asks for ONE root of the T x T standard-form matrix, approximate_multistate for a handful; the
// Jacobi sweeps above compute all T eigenpairs in ~9 (T - 1) latency-bound steps.  For nroots <= kFewRoots the matrix is
// instead (i) reduced to tridiagonal form by T - 2 Householder reflections in LDS (LAPACK dsytd2: 4 barriers per step, the
// symmetric matrix kept in full so that a column is a contiguous row; the reflectors stay in the rows they annihilated),
// (ii) its lowest eigenvalues are located by multisection on the Sturm count (256 abscissae per root and round, IEEE
// division: the count is monotone, so the bracket -- and the result -- is deterministic), (iii) each eigenvector of the
// tridiagonal matrix comes from a twisted factorisation (two lanes run it from both ends at once), is carried back
// through the reflectors by one wave with the vector in registers (wave_sum, no barrier), and (iv) is CHECKED: residual
// and mutual orthogonality in the original matrix.  Anything short of that (a tight cluster among the requested roots,
// an overflow in the recurrences) makes the caller fall back to the Jacobi path.
constexpr int kFewRoots = 4;

struct BigFew {   // LDS scratch (doubles of length Tp unless noted)
    double *d, *e, *tau, *vv, *pv, *ww;   // diagonal, sub-diagonal, reflector scalars; Householder work vectors
    double *dp, *dm, *Y;                  // [kFewRoots][Tp]: forward / backward pivots, eigenvectors
    unsigned short *cnt;                  // [kBT] Sturm counts of a multisection round
    double *iv;                           // [2 kFewRoots + 4] brackets, scalars
    double *lam;                          // [kFewRoots]
};

__device__ __forceinline__ int big_sturm(const double *d, const double *e2, int n, double x, double pivmin) {
    double q = d[0] - x;
    int c = q < 0.0 ? 1 : 0;
    for (int i = 1; i < n; ++i) {
        if (fabs(q) < pivmin) q = -pivmin;
        q = (d[i] - x) - e2[i - 1] / q;
        c += q < 0.0 ? 1 : 0;
    }
    return c;
}

// A: symmetric n x n matrix at pitch P in LDS (destroyed).  Returns true (uniform) when lam[r], Y[r * P + i], r < nroots, hold
// verified eigenpairs of the matrix C (global, pitch P, symmetric) the caller loaded into A; `scale`: a bound on |C|.
__device__ __forceinline__ bool big_few_roots(double *A, const double *Cg, int n, int P, int nroots, double scale,
                                              const BigFew &w, double *red) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l8 = tid & 7, r8 = tid >> 3;
    // (i) Householder tridiagonalisation
    {
        double part = 0.0;
        for (int j = 2 + tid; j < n; j += kBT) part = fma(A[j], A[j], part);
        part = big_block_sum(part, red);
        if (tid == 0) w.iv[0] = part;   // |x[1:]|^2 of the first column
        __syncthreads();
    }
    for (int k = 0; k + 1 < n; ++k) {
        const double xn2 = w.iv[0], alpha = A[(size_t)k * P + k + 1];
        double tau = 0.0, beta = alpha, sc = 0.0;
        if (xn2 > 0.0) {
            beta = -copysign(sqrt(fma(alpha, alpha, xn2)), alpha);
            tau = (beta - alpha) / beta;
            sc = 1.0 / (alpha - beta);
        }
        // live columns k+1 .. n-1 sit in the 16-column chunks cb .. cb + nch - 1 (nch <= 8); v and w are kept ZERO on the
        // other columns of those chunks, so the row loops below need no guards (rows are zero beyond n)
        const int cb = (k + 1) >> 4, nch = ((n - 1) >> 4) - cb + 1;
        for (int j = 16 * cb + tid; j < 16 * (cb + nch); j += kBT) {
            const bool live = j > k && j < n;
            const double v = !live ? 0.0 : (j == k + 1 ? 1.0 : A[(size_t)k * P + j] * sc);
            w.vv[j] = v;
            // the reflector stays in row k; its leading 1 is implicit (A[k][k+1] is `alpha`, which the other waves may
            // still be reading)
            if (live && j > k + 1) A[(size_t)k * P + j] = v;
        }
        if (tid == 0) {
            w.d[k] = A[(size_t)k * P + k];
            w.e[k] = beta;
            w.tau[k] = tau;
        }
        __syncthreads();
        if (tau != 0.0) {   // uniform
            // p = tau A22 v: 8 lanes per row, 16 bytes per lane and chunk (odd rows take the chunks in swapped pairs: the
            // rows of an LDS phase spread over the banks), and p . v in the same pass
            double pd = 0.0;
            for (int i = k + 1 + r8; i < n; i += kBT / 8) {
                const double *row = A + (size_t)i * P + 16 * cb + 2 * l8;
                const double *vp = w.vv + 16 * cb + 2 * l8;
                double t0 = 0.0, t1 = 0.0;
#pragma unroll
                for (int u = 0; u < 8; ++u)
                    if (u < nch) {
                        int uu = u ^ (i & 1);
                        if (uu >= nch) uu = u;
                        const double2 av = *reinterpret_cast<const double2 *>(row + 16 * uu);
                        const double2 vv2 = *reinterpret_cast<const double2 *>(vp + 16 * uu);
                        t0 = fma(av.x, vv2.x, t0);
                        t1 = fma(av.y, vv2.y, t1);
                    }
                const double t = sum8(t0 + t1) * tau;
                if (l8 == 0) {
                    w.pv[i] = t;
                    pd = fma(t, w.vv[i], pd);
                }
            }
            pd = wave_sum(pd);
            if (lane == 0) red[wave] = pd;
            __syncthreads();
            double pvd = 0.0;
#pragma unroll
            for (int q = 0; q < kBW; ++q) pvd += red[q];
            const double K = 0.5 * tau * pvd;
            for (int j = 16 * cb + tid; j < 16 * (cb + nch); j += kBT) w.ww[j] = (j > k && j < n) ? w.pv[j] - K * w.vv[j] : 0.0;
            __syncthreads();
            // A22 -= v w^T + w v^T; the group that owns row k + 1 also leaves |x[1:]|^2 of the next column
            for (int i = k + 1 + r8; i < n; i += kBT / 8) {
                double *row = A + (size_t)i * P + 16 * cb + 2 * l8;
                const double *vp = w.vv + 16 * cb + 2 * l8, *wp = w.ww + 16 * cb + 2 * l8;
                const double vi = w.vv[i], wi = w.ww[i];
                double nx = 0.0;
#pragma unroll
                for (int u = 0; u < 8; ++u)
                    if (u < nch) {
                        int uu = u ^ (i & 1);
                        if (uu >= nch) uu = u;
                        double2 av = *reinterpret_cast<const double2 *>(row + 16 * uu);
                        const double2 vv2 = *reinterpret_cast<const double2 *>(vp + 16 * uu);
                        const double2 ww2 = *reinterpret_cast<const double2 *>(wp + 16 * uu);
                        av.x -= vi * ww2.x + wi * vv2.x;
                        av.y -= vi * ww2.y + wi * vv2.y;
                        *reinterpret_cast<double2 *>(row + 16 * uu) = av;
                        const int j = 16 * (cb + uu) + 2 * l8;
                        if (j > k + 2) nx = fma(av.x, av.x, nx);
                        if (j + 1 > k + 2) nx = fma(av.y, av.y, nx);
                    }
                if (i == k + 1) {   // uniform over the group (columns beyond n are zero)
                    nx = sum8(nx);
                    if (l8 == 0) w.iv[0] = nx;
                }
            }
        } else {
            // no reflection: the next column's norm from the untouched matrix
            if (tid < 64) {
                double nx = 0.0;
                for (int j = k + 3 + lane; j < n; j += 64) nx = fma(A[(size_t)(k + 1) * P + j], A[(size_t)(k + 1) * P + j], nx);
                nx = wave_sum(nx);
                if (lane == 0) w.iv[0] = nx;
            }
        }
        __syncthreads();
    }
    if (tid == 0) w.d[n - 1] = A[(size_t)(n - 1) * P + n - 1];
    __syncthreads();
    EVC_BSTAMP(8);
    // (ii) multisection: e2 -> vv, Gershgorin bracket, 256 abscissae per root and round
    double gl = 1.0e300, gu = -1.0e300, emax = 0.0;
    for (int i = tid; i < n; i += kBT) {
        const double el = i > 0 ? fabs(w.e[i - 1]) : 0.0, er = i + 1 < n ? fabs(w.e[i]) : 0.0;
        gl = fmin(gl, w.d[i] - el - er);
        gu = fmax(gu, w.d[i] + el + er);
        emax = fmax(emax, er);
        if (i + 1 < n) w.vv[i] = w.e[i] * w.e[i];
    }
    gl = -big_block_max(-gl, red);
    gu = big_block_max(gu, red);
    emax = big_block_max(emax, red);
    const double tnorm = fmax(fabs(gl), fabs(gu));
    if (!(tnorm < 1.0e300)) return false;   // NaN / Inf input (uniform)
    const double pivmin = fmax(1.0e-290, 1.0e-290 * emax * emax);
    const int root = tid >> 8, t = tid & 255;
    if (tid < 2 * kFewRoots) w.iv[2 + tid] = (tid & 1) ? gu + 2.3e-16 * tnorm * n : gl - 2.3e-16 * tnorm * n;
    __syncthreads();
    for (int round = 0; round < 7; ++round) {   // 257^7 = 7e16 > 2^53 x the guard factor
        double lo = 0.0, hi = 0.0, x = 0.0;
        if (root < nroots) {
            lo = w.iv[2 + 2 * root];
            hi = w.iv[3 + 2 * root];
            x = lo + (hi - lo) * ((double)(t + 1) * (1.0 / 257.0));
            w.cnt[tid] = (unsigned short)big_sturm(w.d, w.vv, n, x, pivmin);
        }
        __syncthreads();
        if (root < nroots) {
            const int need = root + 1, c = w.cnt[tid], cprev = t > 0 ? w.cnt[tid - 1] : 0;
            if (c >= need && (t == 0 || cprev < need)) {   // exactly one thread (the count is monotone), or none
                w.iv[3 + 2 * root] = x;
                if (t > 0) w.iv[2 + 2 * root] = lo + (hi - lo) * ((double)t * (1.0 / 257.0));
            } else if (t == 255 && c < need) {
                w.iv[2 + 2 * root] = x;
            }
        }
        __syncthreads();
    }
    if (tid < nroots) w.lam[tid] = 0.5 * (w.iv[2 + 2 * tid] + w.iv[3 + 2 * tid]);
    __syncthreads();
    EVC_BSTAMP(9);
    // (iii) eigenvectors: wave r takes root r
    if (wave < nroots) {
        const double lamr = w.lam[wave];
        double *dp = w.dp + (size_t)wave * P, *dm = w.dm + (size_t)wave * P, *Y = w.Y + (size_t)wave * P;
        if (lane < 2) {   // lane 0: pivots from the top, lane 1: from the bottom
            const int dir = lane;
            int i = dir ? n - 1 : 0;
            double q = w.d[i] - lamr;
            for (int step = 0; step < n; ++step) {
                if (fabs(q) < pivmin) q = -pivmin;
                (dir ? dm : dp)[i] = q;
                const int in = dir ? i - 1 : i + 1;
                if (in < 0 || in >= n) break;
                const double ee = w.e[dir ? in : i];
                q = (w.d[in] - lamr) - (ee / q) * ee;
                i = in;
            }
        }
        // (LDS operations of one wave complete in order)
        double g0 = 1.0e300, g1 = 1.0e300;
        if (lane < n) g0 = fabs(dp[lane] + dm[lane] - (w.d[lane] - lamr));
        if (lane + 64 < n) g1 = fabs(dp[lane + 64] + dm[lane + 64] - (w.d[lane + 64] - lamr));
        double gm = fmin(g0, g1);
        gm = fmin(gm, dpp_move<0xB1>(gm));
        gm = fmin(gm, dpp_move<0x4E>(gm));
        gm = fmin(gm, dpp_move<0x141>(gm));
        gm = fmin(gm, dpp_move<0x140>(gm));
        gm = fmin(fmin(readlane_f64(gm, 0), readlane_f64(gm, 16)), fmin(readlane_f64(gm, 32), readlane_f64(gm, 48)));
        const unsigned long long b0 = __ballot(g0 == gm), b1 = __ballot(g1 == gm);
        const int kt = b0 ? __builtin_ctzll(b0) : (b1 ? 64 + __builtin_ctzll(b1) : 0);   // (NaNs: 0, caught by the check)
        if (lane < 2) {   // lane 0 walks up from the twist index, lane 1 down
            double z = 1.0;
            if (lane == 0) {
                Y[kt] = 1.0;
                for (int j = kt; j > 0; --j) {
                    z = -(w.e[j - 1] / dp[j - 1]) * z;
                    Y[j - 1] = z;
                }
            } else {
                for (int j = kt; j + 1 < n; ++j) {
                    z = -(w.e[j] / dm[j + 1]) * z;
                    Y[j + 1] = z;
                }
            }
        }
        double z0 = lane < n ? Y[lane] : 0.0, z1 = lane + 64 < n ? Y[lane + 64] : 0.0;
        {
            const double nn = wave_sum(fma(z0, z0, z1 * z1));
            const double inv = 1.0 / sqrt(nn);
            z0 *= inv;
            z1 *= inv;
        }
        // back through the reflectors: z <- H_0 H_1 ... H_{n-3} z (H_k acts on indices k+1 .. n-1; row k of A holds v_k)
        for (int k = n - 3; k >= 0; --k) {
            const double tk = w.tau[k];
            const double *vr = A + (size_t)k * P;
            const double v0 = lane == k + 1 ? 1.0 : ((lane > k + 1 && lane < n) ? vr[lane] : 0.0);
            const double v1 = lane + 64 == k + 1 ? 1.0 : ((lane + 64 > k + 1 && lane + 64 < n) ? vr[lane + 64] : 0.0);
            const double sdot = wave_sum(fma(v0, z0, v1 * z1)) * tk;
            z0 = fma(-sdot, v0, z0);
            z1 = fma(-sdot, v1, z1);
        }
        if (lane < n) Y[lane] = z0;
        if (lane + 64 < n) Y[lane + 64] = z1;
    }
    __syncthreads();
    EVC_BSTAMP(10);
    // (iv) check in the original matrix: residuals and mutual orthogonality
    double worst = 0.0;
    for (int r = 0; r < nroots; ++r) {
        const double *Y = w.Y + (size_t)r * P;
        const double lamr = w.lam[r];
        for (int i = r8; i < n; i += kBT / 8) {
            double t2 = 0.0;
            for (int j = l8; j < n; j += 8) t2 = fma(Cg[(size_t)i * P + j], Y[j], t2);
            t2 = sum8(t2);
            worst = big_nanmax(worst, fabs(t2 - lamr * Y[i]));
        }
        for (int r2 = 0; r2 < r; ++r2) {
            if (wave == 0) {
                const double *Y2 = w.Y + (size_t)r2 * P;
                double dd = 0.0;
                for (int j = lane; j < n; j += 64) dd = fma(Y[j], Y2[j], dd);
                dd = wave_sum(dd);
                worst = big_nanmax(worst, fabs(dd) * 10.0 * scale);   // |y_r . y_r'| <= 1e-12 (close roots: eps / gap)
            }
        }
    }
    worst = big_block_max(worst, red);
    EVC_BSTAMP(11);
    EVC_BVAL(2, worst / scale);
    EVC_BVAL(3, w.lam[0]);
    EVC_BVAL(4, w.iv[3] - w.iv[2]);
    return worst <= 1.0e-11 * scale;   // (NaN: false)
}

// In-place Cholesky factor of the lower triangle in M (pitch Tp) followed by its in-place inverse: on return the lower
// triangle of M holds B = L^-1.  A matrix that is not positive definite yields NaNs.  dinv, tmp: T doubles each.
__device__ __forceinline__ void big_chol_inverse(double *M, int T, int Tp, double *dinv, double *tmp) {
    const int tid = threadIdx.x, tx = tid & 31, ty = tid >> 5;
    for (int j = 0; j < T; ++j) {
        const double d = M[(size_t)j * Tp + j];
        double rs = __builtin_amdgcn_rsq(d);
        rs = rs * fma(-0.5 * d * rs, rs, 1.5);
        rs = rs * fma(-0.5 * d * rs, rs, 1.5);
        if (tid == 0) dinv[j] = rs;   // 1 / L_jj (the diagonal itself is fixed up below)
        for (int i = j + 1 + tid; i < T; i += kBT) M[(size_t)i * Tp + j] *= rs;
        __syncthreads();
        for (int i = j + 1 + ty; i < T; i += 32) {
            const double lij = M[(size_t)i * Tp + j];
            for (int k = j + 1 + tx; k <= i; k += 32) M[(size_t)i * Tp + k] = fma(-lij, M[(size_t)k * Tp + j], M[(size_t)i * Tp + k]);
        }
        __syncthreads();
    }
    // inverse, column by column from the last: new column j below the diagonal = -(L22^-1 c) / L_jj
    const int l8 = tid & 7, r8 = tid >> 3;
    for (int j = T - 1; j >= 0; --j) {
        for (int i = j + 1 + r8; i < T; i += kBT / 8) {   // (the eight lanes of a group share i)
            double t = 0.0;
            for (int k = j + 1 + l8; k <= i; k += 8) t = fma(M[(size_t)i * Tp + k], M[(size_t)k * Tp + j], t);
            t = sum8(t);
            if (l8 == 0) tmp[i] = t;
        }
        __syncthreads();
        const double ajj = dinv[j];
        for (int i = j + 1 + tid; i < T; i += kBT) M[(size_t)i * Tp + j] = -tmp[i] * ajj;
        if (tid == 0) M[(size_t)j * Tp + j] = ajj;
        __syncthreads();
    }
}

template <bool kLds>
__global__ __launch_bounds__(kBT) void subspace_big_kernel(SolveArgs a) {
    extern __shared__ __align__(16) double sm[];
    {
        const int64_t g = blockIdx.x;
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
        a.scratch += g * a.sscratch;
        if (a.e_shift_dev) a.e_shift = a.e_shift_dev[g];
    }
    const int T = a.T, m = (T + 1) & ~1, Tp = (T + 15) & ~15, Pj = (m + 31) & ~31;
    const size_t Tp2 = (size_t)Tp * Tp;
    const size_t msz = (size_t)m * Pj > Tp2 ? (size_t)m * Pj : Tp2;
    // global scratch: [B (when there is no cache) | W | C | the matrix itself when it does not fit LDS]
    double *Bg = a.bcache ? a.bcache + Tp2 : a.scratch;
    double *Wg = a.scratch + Tp2, *Cg = a.scratch + 2 * Tp2;
    double *M = kLds ? sm : a.scratch + 3 * Tp2;
    double *aux = kLds ? sm + msz : sm;
    double *ev = aux, *tmp = ev + Tp, *c0 = tmp + Tp, *dinv = c0 + Tp, *red = dinv + Tp;
    int *order = reinterpret_cast<int *>(red + 2 * kBW);
    BigFew fw;   // (the Householder work vectors share the buffers of the phases that are over / not yet reached)
    fw.d = ev;
    fw.vv = tmp;
    fw.pv = dinv;
    fw.ww = c0;
    fw.e = reinterpret_cast<double *>(order + Tp + (Tp & 1));
    fw.tau = fw.e + Tp;
    fw.dp = fw.tau + Tp;
    fw.dm = fw.dp + (size_t)kFewRoots * Tp;
    fw.Y = fw.dm + (size_t)kFewRoots * Tp;
    fw.iv = fw.Y + (size_t)kFewRoots * Tp;
    fw.lam = fw.iv + 2 * kFewRoots + 4;
    fw.cnt = reinterpret_cast<unsigned short *>(fw.lam + kFewRoots);
    const int tid = threadIdx.x;
    const int64_t P = (int64_t)T * (T + 1) / 2;
    const bool pairs = (a.layout == EVC_LAYOUT_PAIR5 || a.layout == EVC_LAYOUT_PACK2 || a.layout == EVC_LAYOUT_SYM8);
    const int64_t rows2 = pairs ? P : (int64_t)T * T;

    EVC_BSTAMP(0);
    // (a) B = L^-1: from the cache when the overlap matrix it was computed from is bit-identical to this call's
    bool hit = false;
    if (a.bcache) {
        bool diff = false;
        for (int idx = tid; idx < T * T; idx += kBT) {
            const int i = idx / T, j = idx - i * T;
            if (i >= j) diff = diff || !(a.bcache[idx] == a.S[idx]);
        }
        hit = !big_block_any(diff, red);
    }
    if (!hit) {   // uniform
        for (size_t idx = tid; idx < Tp2; idx += kBT) {
            const int i = (int)(idx / Tp), j = (int)(idx - (size_t)i * Tp);
            M[idx] = (i < T && j <= i) ? a.S[(size_t)i * T + j] : 0.0;
        }
        __syncthreads();
        big_chol_inverse(M, T, Tp, dinv, tmp);
        for (size_t idx = tid; idx < Tp2; idx += kBT) {
            const int i = (int)(idx / Tp), j = (int)(idx - (size_t)i * Tp);
            Bg[idx] = (i < T && j <= i) ? M[idx] : 0.0;
        }
        if (a.bcache)
            for (int idx = tid; idx < T * T; idx += kBT) a.bcache[idx] = a.S[idx];
        __syncthreads();
    }
    EVC_BSTAMP(1);
    // (b) H from the span partials (stored [span][row]), placed as the reference does (evcont.py:41-68)
    for (size_t idx = tid; idx < Tp2; idx += kBT) M[idx] = 0.0;
    __syncthreads();
    for (int r = tid; r < T * T; r += kBT) {
        double s = 0.0;
        for (int k = 0; k < a.nsp1; ++k) s += a.h1part[(int64_t)k * T * T + r];
        const int i = r / T;
        M[(size_t)i * Tp + (r - i * T)] = a.alpha1 * s;
    }
    __syncthreads();
    for (int64_t r = tid; r < rows2; r += kBT) {
        double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
        int k = 0;
        for (; k + 4 <= a.nsp2; k += 4) {
            s0 += a.h2part[(int64_t)(k + 0) * rows2 + r];
            s1 += a.h2part[(int64_t)(k + 1) * rows2 + r];
            s2 += a.h2part[(int64_t)(k + 2) * rows2 + r];
            s3 += a.h2part[(int64_t)(k + 3) * rows2 + r];
        }
        for (; k < a.nsp2; ++k) s0 += a.h2part[(int64_t)k * rows2 + r];
        const double s = (s0 + s1) + (s2 + s3);
        int ia, ib;
        if (pairs) {
            ia = (int)tri_row(r);
            ib = (int)(r - (int64_t)ia * (ia + 1) / 2);
        } else {
            ia = (int)(r / T);
            ib = (int)(r - (int64_t)ia * T);
        }
        M[(size_t)ia * Tp + ib] += a.alpha2 * s;
    }
    __syncthreads();
    if (a.Hout)
        for (int idx = tid; idx < T * T; idx += kBT) {
            const int i = idx / T;
            a.Hout[idx] = M[(size_t)i * Tp + (idx - i * T)];
        }
    __syncthreads();
    for (int idx = tid; idx < T * T; idx += kBT) {   // Hs from the LOWER triangle (dsygst)
        const int i = idx / T, j = idx - i * T;
        if (j > i) M[(size_t)i * Tp + j] = M[(size_t)j * Tp + i];
    }
    __syncthreads();
    EVC_BSTAMP(2);
    // (c) W = B Hs (W[i][k] = sum_l B[i][l] Hs[k][l]);  C = W B^T (C[i][j] = sum_k W[i][k] B[j][k])
    big_mm_rr(Tp, Bg, Tp, M, Tp, [&](int i, int k, double v) { Wg[(size_t)i * Tp + k] = v; });
    __syncthreads();
    big_mm_rr(Tp, Wg, Tp, Bg, Tp, [&](int i, int j, double v) { Cg[(size_t)i * Tp + j] = v; });
    __syncthreads();
    EVC_BSTAMP(3);
    // exact symmetry (as the rotations assume) and the Gershgorin bound of the shift
    for (size_t idx = tid; idx < Tp2; idx += kBT) {
        const int i = (int)(idx / Tp), j = (int)(idx - (size_t)i * Tp);
        if (j < i) {
            const double v = 0.5 * (Cg[idx] + Cg[(size_t)j * Tp + i]);
            Cg[idx] = v;
            Cg[(size_t)j * Tp + i] = v;
        }
    }
    __syncthreads();
    double rmax = 0.0;
    for (int i = tid; i < T; i += kBT) {
        double rs = 0.0;
        for (int j = 0; j < T; ++j) rs += fabs(Cg[(size_t)i * Tp + j]);
        rmax = big_nanmax(rmax, rs);
    }
    rmax = big_block_max(rmax, red);
    const double shift = 2.0 * rmax + 1.0e-300;   // eigenvalues of the shifted matrix within [1, 3] x the bound
    EVC_BSTAMP(4);
    // (d') a few roots: tridiagonal route, verified; anything else (or a failed check): all eigenpairs by Jacobi
    bool few = false;
    if (kLds && a.few && a.nroots <= kFewRoots && T <= 128) {   // uniform
        for (size_t idx = tid; idx < Tp2; idx += kBT) M[idx] = Cg[idx];
        __syncthreads();
        few = big_few_roots(M, Cg, T, Tp, a.nroots, rmax + 1.0e-300, fw, red);
        EVC_BVAL(1, few ? 1.0 : 0.0);
        __syncthreads();
    }
    EVC_BSTAMP(5);
    if (!few) {
    // (d) G0, column-major with pitch Pj
    bool warm = false;
    if (a.warm && a.vstd) {   // uniform: the previous eigenvectors (rows of vstd, pitch Tp) must be orthonormal
        double dev = 0.0;
        big_mm_rr(Tp, a.vstd, Tp, a.vstd, Tp, [&](int i, int j, double v) {
            const double e = v - ((i == j && i < T) ? 1.0 : 0.0);
            dev = fma(e, e, dev);
        });
        dev = big_block_sum(dev, red);
        warm = dev < 1.0e-16;   // (NaN: false)
    }
    for (size_t idx = tid; idx < (size_t)m * Pj; idx += kBT) {
        const int j = (int)(idx / Pj), i = (int)(idx - (size_t)j * Pj);
        double v = 0.0;
        if (!warm && i < T && j < T) v = Cg[(size_t)i * Tp + j] + (i == j ? shift : 0.0);
        M[idx] = v;
    }
    __syncthreads();
    if (warm) {   // G[j][i] = sum_k Vt[j][k] C[i][k] + shift Vt[j][i]
        big_mm_rr(Tp, a.vstd, Tp, Cg, Tp, [&](int j, int i, double v) {
            if (j < T && i < T) M[(size_t)j * Pj + i] = v + shift * a.vstd[(size_t)j * Tp + i];
        });
        __syncthreads();
    }
    big_jacobi(M, m, Pj, red);
    // column norms = eigenvalues + shift; normalised columns = eigenvectors
    for (int j = tid >> 4; j < m; j += kBT / 16) {
        const int s = tid & 15;
        double nn = 0.0;
        for (int i = 2 * s; i < Pj; i += 32) {
            const double2 x = *reinterpret_cast<const double2 *>(M + (size_t)j * Pj + i);
            nn = fma(x.x, x.x, fma(x.y, x.y, nn));
        }
        nn = sum16(nn);
        const double l = sqrt(nn);
        const double inv = l > 1.0e-300 ? 1.0 / l : 0.0;   // (the decoupled dummy column of an odd problem stays zero)
        for (int i = 2 * s; i < Pj; i += 32) {
            double2 x = *reinterpret_cast<const double2 *>(M + (size_t)j * Pj + i);
            x.x *= inv;
            x.y *= inv;
            *reinterpret_cast<double2 *>(M + (size_t)j * Pj + i) = x;
        }
        if (s == 0 && j < Tp) ev[j] = l - shift;
    }
    __syncthreads();
    if (a.vstd)
        for (size_t idx = tid; idx < Tp2; idx += kBT) {
            const int j = (int)(idx / Tp), i = (int)(idx - (size_t)j * Tp);
            a.vstd[idx] = (j < T && i < T) ? M[(size_t)j * Pj + i] : 0.0;
        }
    // (e) ascending order
    for (int j = tid; j < T; j += kBT) {
        int rank = 0;
        const double v = ev[j];
        for (int k = 0; k < T; ++k) rank += (ev[k] < v || (ev[k] == v && k < j)) ? 1 : 0;
        order[rank] = j;
    }
    __syncthreads();
    }   // !few
    EVC_BSTAMP(6);
    // back-transformation c_i = sum_{k >= i} B[k][i] y_k for the requested roots
    for (int idx = tid; idx < a.nroots * T; idx += kBT) {
        const int root = idx / T, i = idx - root * T;
        const double *y = few ? fw.Y + (size_t)root * Tp : M + (size_t)order[root] * Pj;
        double c = 0.0;
        for (int k = i; k < T; ++k) c = fma(Bg[(size_t)k * Tp + i], y[k], c);
        a.evecs[idx] = c;
        if (root == 0) tmp[i] = c;   // (c0 may still hold Householder work: staged in tmp, copied below)
    }
    for (int r = tid; r < a.nroots; r += kBT) a.evals[r] = (few ? fw.lam[r] : ev[order[r]]) + a.e_shift;
    __syncthreads();
    for (int i = tid; i < T; i += kBT) c0[i] = tmp[i];
    __syncthreads();
    // weights of root 0 for the predicted RDMs (gradients_loewdin.py:343-356)
    if (a.w1)
        for (int idx = tid; idx < T * T; idx += kBT) {
            const int ia = idx / T;
            const double w = c0[ia] * c0[idx - ia * T];
            a.w1[idx] = w;
            // transposed copy for the batched K8: [row][slot] in the workspace of the group's first geometry
            if (a.w1t) a.w1t[(int64_t)idx * kMaxBatchG + (int)(blockIdx.x % kMaxBatchG)] = w;
        }
    if (a.w2) {
        for (int64_t r = tid; r < a.w2_count; r += kBT) {
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
            if (a.w2t) a.w2t[r * kMaxBatchG + (int)(blockIdx.x % kMaxBatchG)] = w;
        }
    }
    EVC_BSTAMP(7);
}

// ------------------------------------------------------------------ Loewdin orthogonalisation, 32 < n <= 64
// S = U diag(s) U^T by the same one-sided Jacobi (S is positive definite: no shift), X = U diag(s > 1e-15 ? s^-1/2 : 0) U^T,
// h1 = X^T h X (electron_integral_utils.py:6-18,135; gradients_loewdin.py:336-338), three matrices in LDS.  Replaces the
// two-sided LDS Jacobi of dense_small.hip for these sizes (1.4 ms at n = 58: cc-pVTZ water).
__global__ __launch_bounds__(kBT) void loewdin_big_kernel(LoewdinArgs a) {
    extern __shared__ __align__(16) double sm[];
    const int n = a.n, m = (n + 1) & ~1, Tp = (n + 15) & ~15, Pj = (m + 31) & ~31;
    const int64_t g = blockIdx.x;
    const double *__restrict__ S = a.S + g * a.sS;
    const double *__restrict__ h = a.h ? a.h + g * a.sh : nullptr;
    double *__restrict__ X = a.X + g * a.sws;
    double *__restrict__ U = a.U + g * a.sws;
    double *__restrict__ sv = a.s + g * a.sws;
    double *__restrict__ h1 = a.h1 ? a.h1 + g * a.sws : nullptr;
    const size_t Tp2 = (size_t)Tp * Tp;
    const size_t gsz = (size_t)m * Pj > Tp2 ? (size_t)m * Pj : Tp2;
    // n <= 64: G, A, B in LDS; beyond: A and B in the caller's scratch (global memory, L2 resident)
    const bool ext = n > 64;
    double *G = sm;
    double *A = ext ? a.scratch + g * a.sscratch : G + gsz, *B = A + Tp2;
    double *f = ext ? G + gsz : B + Tp2, *red = f + Tp;
    const int tid = threadIdx.x;
    // part 2: U and s only (the response half of a split step, pipeline.hip); part 3: X and h1 only, and only where the
    // Newton-Schulz launch in front of this one declined (its flag word)
    const bool want_u = a.part != 1 && a.part != 3, want_x = a.part != 2;
    if (a.part == 3 && a.flag[g * a.sws] != 0.0) return;
    // numpy.linalg.eigh reads the lower triangle
    bool warm = false;
    if (a.warm && want_u) {   // A = Vt (rows = previous eigenvectors), B = S symmetrised; G0 = Vt S
        for (size_t idx = tid; idx < Tp2; idx += kBT) {
            const int i = (int)(idx / Tp), j = (int)(idx - (size_t)i * Tp);
            const bool in = i < n && j < n;
            A[idx] = in ? U[(size_t)j * n + i] : 0.0;
            B[idx] = in ? S[(size_t)(i > j ? i : j) * n + (i > j ? j : i)] : 0.0;
        }
        __syncthreads();
        double dev = 0.0;
        big_mm_rr(Tp, A, Tp, A, Tp, [&](int i, int j, double v) {
            const double e = v - ((i == j && i < n) ? 1.0 : 0.0);
            dev = fma(e, e, dev);
        });
        dev = big_block_sum(dev, red);
        warm = dev < 1.0e-16;
    }
    for (size_t idx = tid; idx < (size_t)m * Pj; idx += kBT) {
        const int j = (int)(idx / Pj), i = (int)(idx - (size_t)j * Pj);
        G[idx] = (!warm && i < n && j < n) ? S[(size_t)(i > j ? i : j) * n + (i > j ? j : i)] : 0.0;
    }
    __syncthreads();
    if (warm) {
        big_mm_rr(Tp, A, Tp, B, Tp, [&](int j, int i, double v) {
            if (j < n && i < n) G[(size_t)j * Pj + i] = v;
        });
        __syncthreads();
    }
    big_jacobi(G, m, Pj, red);
    for (int j = tid >> 4; j < m; j += kBT / 16) {
        const int s = tid & 15;
        double nn = 0.0;
        for (int i = 2 * s; i < Pj; i += 32) {
            const double2 x = *reinterpret_cast<const double2 *>(G + (size_t)j * Pj + i);
            nn = fma(x.x, x.x, fma(x.y, x.y, nn));
        }
        nn = sum16(nn);
        const double l = sqrt(nn);
        const double inv = l > 1.0e-300 ? 1.0 / l : 0.0;
        for (int i = 2 * s; i < Pj; i += 32) {
            double2 x = *reinterpret_cast<const double2 *>(G + (size_t)j * Pj + i);
            x.x *= inv;
            x.y *= inv;
            *reinterpret_cast<double2 *>(G + (size_t)j * Pj + i) = x;
        }
        if (s == 0 && j < Tp) {
            f[j] = (j < n && l > 1.0e-15) ? 1.0 / sqrt(l) : 0.0;
            if (j < n && want_u) sv[j] = l;
        }
    }
    __syncthreads();
    // (a zero column -- the decoupled dummy dimension of an odd n -- stays zero and has f = 0)
    for (size_t idx = tid; idx < Tp2; idx += kBT) {
        const int i = (int)(idx / Tp), k = (int)(idx - (size_t)i * Tp);
        const bool in = i < n && k < n;
        const double v = in ? G[(size_t)k * Pj + i] : 0.0;
        A[idx] = v;
        B[idx] = in ? v * f[k] : 0.0;
        if (in && want_u) U[(size_t)i * n + k] = v;
    }
    if (!want_x) return;
    __syncthreads();
    // X = (V f) V^T, kept at pitch Tp in the G region (free now)
    for (size_t idx = tid; idx < Tp2; idx += kBT) G[idx] = 0.0;
    __syncthreads();
    big_mm_rr(Tp, B, Tp, A, Tp, [&](int i, int j, double v) {
        if (i < n && j < n) {
            G[(size_t)i * Tp + j] = v;
            X[(size_t)i * n + j] = v;
        }
    });
    if (!(h && h1)) return;
    __syncthreads();
    for (size_t idx = tid; idx < Tp2; idx += kBT) {
        const int i = (int)(idx / Tp), j = (int)(idx - (size_t)i * Tp);
        A[idx] = (i < n && j < n) ? h[(size_t)i * n + j] : 0.0;
    }
    __syncthreads();
    // Tt[j][i] = (h X)[i][j] = sum_k X[j][k] h[i][k] (X symmetric);  h1[i][j] = sum_k X[i][k] Tt[j][k]
    big_mm_rr(Tp, G, Tp, A, Tp, [&](int j, int i, double v) { B[(size_t)j * Tp + i] = v; });
    __syncthreads();
    big_mm_rr(Tp, G, Tp, B, Tp, [&](int i, int j, double v) {
        if (i < n && j < n) h1[(size_t)i * n + j] = v;
    });
}

int launch_loewdin_big(const LoewdinArgs &a, int count, hipStream_t st) {
    const size_t n = a.n, m = (n + 1) & ~(size_t)1, Tp = (n + 15) & ~(size_t)15, Pj = (m + 31) & ~(size_t)31;
    const size_t gsz = m * Pj > Tp * Tp ? m * Pj : Tp * Tp;
    if (n > 64 && !a.scratch) {
        set_error("loewdin: n=%d needs a scratch buffer", a.n);
        return -1;
    }
    const size_t lds = sizeof(double) * (gsz + (n > 64 ? 0 : 2 * Tp * Tp) + Tp + 2 * kBW) + 64;
    static LdsAttr attr;
    if (int rc = allow_dynamic_lds(loewdin_big_kernel, attr, 160 * 1024, "loewdin_big")) return rc;
    hipLaunchKernelGGL(loewdin_big_kernel, dim3(count), dim3(kBT), lds, st, a);
    EVC_LAUNCH_CHECK("loewdin_big");
    return 0;
}

static size_t big_aux_bytes(int T) {
    const int Tp = (T + 15) & ~15;
    return sizeof(double) * ((size_t)4 * Tp + 2 * kBW) + sizeof(int) * (size_t)(Tp + 2) +
           sizeof(double) * ((size_t)(2 + 3 * kFewRoots) * Tp + 3 * kFewRoots + 8) + sizeof(unsigned short) * kBT + 64;
}
static size_t big_matrix_doubles(int T) {
    const size_t m = (T + 1) & ~1, Tp = (T + 15) & ~15, Pj = (m + 31) & ~(size_t)31;
    return m * Pj > Tp * Tp ? m * Pj : Tp * Tp;
}
static bool big_fits_lds(int T) { return sizeof(double) * big_matrix_doubles(T) + big_aux_bytes(T) <= 160 * 1024; }

size_t subspace_big_scratch_doubles(int T) {
    const size_t Tp = (T + 15) & ~15;
    return 3 * Tp * Tp + (big_fits_lds(T) ? 0 : big_matrix_doubles(T));
}

int launch_subspace_big(const SolveArgs &a_in, int count, hipStream_t st) {
    SolveArgs a = a_in;
    // EVC_SUBSPACE_FEW=0: every call through the Jacobi sweeps (A/B timing, tests of that path)
    static const int few_on = getenv("EVC_SUBSPACE_FEW") ? atoi(getenv("EVC_SUBSPACE_FEW")) : 1;
    a.few = few_on;
    if (!a.scratch) {
        set_error("subspace solve: T=%d needs a scratch buffer (evc_subspace_solve_ws_bytes)", a.T);
        return -1;
    }
    if (big_fits_lds(a.T)) {
        const size_t lds = sizeof(double) * big_matrix_doubles(a.T) + big_aux_bytes(a.T);
        static LdsAttr attr;
        if (int rc = allow_dynamic_lds(subspace_big_kernel<true>, attr, 160 * 1024, "subspace_big")) return rc;
        hipLaunchKernelGGL(subspace_big_kernel<true>, dim3(count), dim3(kBT), lds, st, a);
    } else {
        hipLaunchKernelGGL(subspace_big_kernel<false>, dim3(count), dim3(kBT), big_aux_bytes(a.T), st, a);
    }
    EVC_LAUNCH_CHECK("subspace_big");
    return 0;
}

}  // namespace evc

#ifdef EVC_DEBUG_STAMPS
extern "C" int evc_debug_read_big(long long *stamps, double *vals, int n) {
    if (n > 16) n = 16;
    hipError_t e = hipMemcpyFromSymbol(stamps, HIP_SYMBOL(evc::g_big_stamp), sizeof(long long) * n);
    if (e == hipSuccess) e = hipMemcpyFromSymbol(vals, HIP_SYMBOL(evc::g_big_val), sizeof(double) * n);
    return (int)e;
}
#endif
