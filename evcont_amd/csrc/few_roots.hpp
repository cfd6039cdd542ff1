// A few lowest eigenpairs of a small symmetric matrix (m <= 32) on ONE wave, in double precision throughout, without
// any refinement pass: what the subspace problem of the energy+force path asks for (its ground state;
// ab_initio_eigenvector_continuation.py:73-88 takes argmin of eigh, approximate_multistate a handful of roots).
//
//   1. Householder tridiagonalisation with the matrix in REGISTERS: lane j holds row j (lanes 32..63 mirror 0..31, so
//      every wave-wide reduction is twice the sum over the rows), the loop over the columns is unrolled, so the column
//      a step works on is a register, not an LDS round trip; v and w are broadcast through LDS (one 16-byte broadcast
//      read per two columns), and the work of a step shrinks with the trailing matrix.
//   2. The nroots lowest eigenvalues of the tridiagonal matrix by multisection on the Sturm count (IEEE division: the
//      count is monotone, results are deterministic), 64 / nroots abscissae per root and round.
//   3. Their eigenvectors by twisted factorisation (the two directions on two lanes), carried back through the
//      reflectors with the vector distributed over the lanes (one wave_sum per reflector).
//   4. Verification in the ORIGINAL matrix: residual and mutual overlap; the caller falls back to the full
//      eigensolver when it fails (a cluster among the requested roots, an overflow in a recurrence).
// Same scheme as the large-T kernel's few-roots route (subspace_big.hip: big_few_roots), which runs on 16 waves with
// the matrix in LDS; here nothing waits for a barrier.
#pragma once
#include "common.hpp"

// (dense_small.hip includes this file inside namespace evc, behind its phase-stamp macros)
#ifndef EVC_FEW_STAMP
#define EVC_FEW_STAMP(i_) do { } while (0)
#endif

namespace few {

constexpr int kMaxRoots = 4;
// LDS scratch (doubles): vb[32] wb[32] dd[32] ee[32] bb[32] Vh[32][32] fD[2][4][32] fF[2][4][32] zb[4][32]
constexpr int kScratch = 5 * 32 + 32 * 32 + 2 * 2 * kMaxRoots * 32 + kMaxRoots * 32;

__device__ __forceinline__ double sum32m(double v) { return 0.5 * wave_sum(v); }   // lanes 32..63 mirror 0..31
__device__ __forceinline__ float sum32m(float v) {
    v += dpp_move<0xB1>(v);
    v += dpp_move<0x4E>(v);
    v += dpp_move<0x141>(v);
    v += dpp_move<0x140>(v);
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 0)) +
           __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 16));
}
__device__ __forceinline__ double wave_sum_t(double v) { return wave_sum(v); }
__device__ __forceinline__ float wave_sum_t(float v) {
    v += dpp_move<0xB1>(v);
    v += dpp_move<0x4E>(v);
    v += dpp_move<0x141>(v);
    v += dpp_move<0x140>(v);
    const auto rl = [&](int l) { return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), l)); };
    return (rl(0) + rl(16)) + (rl(32) + rl(48));
}
__device__ __forceinline__ double rdlane(double v, int l) { return readlane_f64(v, l); }
__device__ __forceinline__ float rdlane(float v, int l) {
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), l));
}
__device__ __forceinline__ double hh_sqrt(double x) { return sqrt(x); }
__device__ __forceinline__ float hh_sqrt(float x) { return __builtin_sqrtf(x); }
__device__ __forceinline__ double hh_fma(double a, double b, double c) { return __builtin_fma(a, b, c); }
__device__ __forceinline__ float hh_fma(float a, float b, float c) { return __builtin_fmaf(a, b, c); }
__device__ __forceinline__ double hh_rcp(double x) { return 1.0 / x; }
__device__ __forceinline__ float hh_rcp(float x) { return __builtin_amdgcn_rcpf(x); }

// Householder tridiagonalisation of a symmetric m x m matrix (m <= MR <= 32) whose row j sits in the registers a[0..MR-1]
// of lane j (lanes 32..63 mirror lanes 0..31; entries >= m zero): A <- H_k A H_k, H_k = I - beta_k v_k v_k^T (v_k zero
// up to row k), k = 0 .. m-3.  The loop over k is ROLLED (an unrolled one is tens of KB of straight-line code that is
// executed once: instruction fetch then costs more than the arithmetic -- measured, round 4): after every step the
// registers are shifted by one, so a[0] always is the column the next step works on; v and w travel through LDS in the
// same rotated frame (lane j writes slot (j - k) mod MR, everybody reads slots 0 .. MR-1 at static offsets).  Results
// in LDS: dd[i] diagonal, ee[i] off-diagonal (ee[m-1] = 0), bb[k] = beta_k, Vh[k * 32 + j] = component j of v_k.
// vrel / wrel: MR elements each.  Called by one whole wave.  (A variant that splits the columns between the two
// half-waves -- half the arithmetic per lane -- is left for later: tools/micro/hh_test.hip is its unit test.)
template <typename T, int MR>
__device__ __forceinline__ void householder_tridiag_wave(T (&a)[MR], int m, T *vrel, T *wrel, T *dd, T *ee, T *bb, T *Vh,
                                                         T tiny_abs, T tiny_rel) {
    typedef T T2 __attribute__((ext_vector_type(2)));
    const int lane = threadIdx.x & 63, j = lane & 31, h = lane >> 5;
#pragma unroll 1
    for (int k = 0; k + 2 < m; ++k) {
        const T x = (j > k) ? a[0] : (T)0;
        const T sig = sum32m(x * x);
        const T xk1 = rdlane(x, k + 1);
        const T rest = sig - xk1 * xk1;
        T alpha = xk1, beta = (T)0, v = (T)0;
        if (rest > tiny_abs && rest > tiny_rel * sig) {   // (uniform) something to remove below the subdiagonal
            const T rt = hh_sqrt(sig);
            alpha = xk1 > (T)0 ? -rt : rt;
            beta = hh_rcp(sig - xk1 * alpha);
            v = (j == k + 1) ? xk1 - alpha : x;
        }
        if (lane == 0) {
            ee[k] = alpha;
            bb[k] = beta;
        }
        if (lane == k) dd[k] = a[0];
        const int slot = j >= k ? j - k : MR - k + j;   // rotated frame: slot c <-> column k + c
        if (h == 0) {
            if (j < MR) vrel[slot] = v;
            Vh[k * 32 + j] = v;
        }
        // (the slots are written as scalars by other lanes and read as 16-byte vectors: the compiler barrier keeps type-based
        //  alias analysis from moving the vector loads in front of the stores)
        asm volatile("" ::: "memory");
        // p = beta A v over the trailing columns (slot 0 = column k: v = 0 there; slots beyond the matrix hold zeros)
        T p0 = (T)0, p1 = (T)0;
#pragma unroll
        for (int c = 0; c < MR; c += 2) {
            const T2 vc = *reinterpret_cast<const T2 *>(vrel + c);
            p0 = hh_fma(a[c], vc[0], p0);
            p1 = hh_fma(a[c + 1], vc[1], p1);
        }
        const T p = (p0 + p1) * beta;
        const T K = (T)0.5 * beta * sum32m(p * v);
        const T w = (j > k) ? p - K * v : (T)0;
        if (h == 0 && j < MR) wrel[slot] = w;
        asm volatile("" ::: "memory");
        // A <- A - v w^T - w v^T, and the shift by one column
        {
            const T2 vc = *reinterpret_cast<const T2 *>(vrel), wc = *reinterpret_cast<const T2 *>(wrel);
            a[0] = a[1] - (v * wc[1] + w * vc[1]);
        }
#pragma unroll
        for (int c = 2; c < MR; c += 2) {
            const T2 vc = *reinterpret_cast<const T2 *>(vrel + c), wc = *reinterpret_cast<const T2 *>(wrel + c);
            a[c - 1] = a[c] - (v * wc[0] + w * vc[0]);
            a[c] = a[c + 1] - (v * wc[1] + w * vc[1]);
        }
        asm volatile("" ::: "memory");
        a[MR - 1] = (T)0;
    }
    // the trailing 2 x 2 block (m >= 2; a[0], a[1] = its columns) / the single element (m == 1)
    if (m >= 2) {
        if (lane == m - 2) dd[m - 2] = a[0];
        if (lane == m - 1) {
            dd[m - 1] = a[1];
            ee[m - 2] = a[0];
        }
    } else if (lane == 0) {
        dd[0] = a[0];
    }
    if (lane == 0) ee[m - 1] = (T)0;
}

// Cs: the matrix (LDS, pitch ldc, both triangles); on success lam[r], Y[r * ldy + i] (unit vectors), r < nroots.
// Called by one whole wave (64 lanes); everything is wave-uniform control flow.
// Returns 0 on success, else why not (1: empty / non-finite matrix, 2: residual, 3: norm, 4: overlap of two roots,
// 5: the Rayleigh-quotient iteration left its bracket, 6: the value is not eigenvalue number r);
// dbg (may be NULL): [0] worst residual / |C|, [1] lowest eigenvalue.
template <int MR>
__device__ __forceinline__ int few_roots_wave(const double *Cs, int ldc, int m, int nroots, double *lam, double *Y, int ldy,
                                              double *scr, double *dbg = nullptr) {
    const int lane = threadIdx.x & 63, j = lane & 31, h = lane >> 5;
    double *vb = scr, *wb = vb + 32, *dd = wb + 32, *ee = dd + 32, *bb = ee + 32, *Vh = bb + 32;
    double *fD = Vh + 32 * 32, *fF = fD + 2 * kMaxRoots * 32, *zb = fF + 2 * kMaxRoots * 32;
    double a[MR];
#pragma unroll
    for (int c = 0; c < MR; ++c) a[c] = (j < m && c < m) ? Cs[j * ldc + c] : 0.0;
    double anorm = 0.0;
#pragma unroll
    for (int c = 0; c < MR; ++c) anorm = fmax(anorm, fabs(a[c]));
    {
        // max over the lanes (values identical in both halves)
        double t = anorm;
        t = fmax(t, dpp_move<0xB1>(t));
        t = fmax(t, dpp_move<0x4E>(t));
        t = fmax(t, dpp_move<0x141>(t));
        t = fmax(t, dpp_move<0x140>(t));
        anorm = fmax(fmax(readlane_f64(t, 0), readlane_f64(t, 16)), fmax(readlane_f64(t, 32), readlane_f64(t, 48)));
    }
    if (!(anorm > 0.0) || !(anorm < 1.0e300)) return 1;
    EVC_FEW_STAMP(50);
    householder_tridiag_wave<double, MR>(a, m, vb, wb, dd, ee, bb, Vh, 1.0e-290, 1.0e-30);
    EVC_FEW_STAMP(51);
    // ---- the nroots lowest eigenvalues: multisection on the Sturm count
    double dr = j < m ? dd[j] : 0.0;
    const double er = (j + 1 < m) ? ee[j] : 0.0, el = (j > 0 && j < m) ? ee[j - 1] : 0.0;
    double glo, ghi;
    {
        double lo = j < m ? dr - fabs(el) - fabs(er) : 1.0e300, hi = j < m ? dr + fabs(el) + fabs(er) : -1.0e300;
        lo = fmin(lo, dpp_move<0xB1>(lo));
        lo = fmin(lo, dpp_move<0x4E>(lo));
        lo = fmin(lo, dpp_move<0x141>(lo));
        lo = fmin(lo, dpp_move<0x140>(lo));
        hi = fmax(hi, dpp_move<0xB1>(hi));
        hi = fmax(hi, dpp_move<0x4E>(hi));
        hi = fmax(hi, dpp_move<0x141>(hi));
        hi = fmax(hi, dpp_move<0x140>(hi));
        glo = fmin(readlane_f64(lo, 0), readlane_f64(lo, 16));
        ghi = fmax(readlane_f64(hi, 0), readlane_f64(hi, 16));
        const double pad = 1.0e-12 * fmax(fabs(glo), fabs(ghi)) + 1.0e-300;
        glo -= pad;
        ghi += pad;
    }
    // (2a) brackets of the nroots lowest eigenvalues by multisection on the Sturm count in SINGLE precision (every lane
    //      keeps the tridiagonal matrix in registers; 64 / nroots abscissae per root and round): good to ~1e-7 of the
    //      spectral range, which is all the Rayleigh-quotient iteration below needs as a start
    const double tiny = 1.0e-290 + 1.0e-30 * fmax(fabs(glo), fabs(ghi)) * fmax(fabs(glo), fabs(ghi));
    const double span = ghi - glo;
    float df[MR], e2f[MR];
    {
        const double sc = 1.0 / span;   // abscissae and matrix relative to (glo, span): t = (x - glo) / span in [0, 1]
#pragma unroll
        for (int i = 0; i < MR; ++i) {
            df[i] = i < m ? (float)((dd[i] - glo) * sc) : 0.0f;
            const double e = (i + 1 < m) ? ee[i] * sc : 0.0;
            e2f[i] = fmaxf((float)(e * e), 1.0e-36f);
        }
    }
    const int per = nroots == 1 ? 64 : (nroots == 2 ? 32 : 16);
    const int myr = lane / per, mys = lane - myr * per;
    float flo = 0.0f, fwd = 1.0f;   // bracket of MY root in units of the span (uniform within a root's lane group)
    const int rounds = nroots == 1 ? 4 : (nroots == 2 ? 5 : 6);
#pragma unroll 1
    for (int it = 0; it < rounds; ++it) {
        const float wd = fwd / (float)(per + 1);
        const float xa = flo + wd * (float)(mys + 1);
        float q = df[0] - xa;
        int cnt = q < 0.0f ? 1 : 0;
#pragma unroll
        for (int i = 1; i < MR; ++i) {
            if (i >= m) break;   // uniform: one test per step that is taken, none behind the end
            q = (df[i] - xa) - e2f[i - 1] * __builtin_amdgcn_rcpf(q);   // (a zero pivot gives -inf: counted, then d - x)
            cnt += q < 0.0f ? 1 : 0;
        }
        // eigenvalue r (ascending, 0-based) is >= x  <=>  count(x) <= r: number of such abscissae within my group
        const unsigned long long mask = __ballot(cnt <= myr);
        const unsigned long long grp = per == 64 ? ~0ull : (((1ull << per) - 1ull) << (myr * per));
        const int below = __popcll(mask & grp);
        flo = flo + wd * (float)below;
        fwd = wd;
    }
    const double mylam = glo + span * ((double)flo + 0.5 * (double)fwd);   // (uniform within the group of root myr)
    const double mywid = span * (double)fwd + 4.0e-7 * span;              // what the single-precision count can be off by
    EVC_FEW_STAMP(52);
    // (2b / 3) per root, in DOUBLE precision: twisted factorisation at the bracket's midpoint (lanes (r, dir): the two
    //      directions of root r on two lanes), twist at the smallest |gamma|, z by the two recurrences; the Rayleigh
    //      quotient of that z, lambda + gamma_k / |z|^2, is good to the SQUARE of the start error: a second factorisation
    //      at it gives the vector, and its own correction the eigenvalue.  A correction that leaves the bracket, or a
    //      final Sturm count that does not put the value at index r, fails the call.
    const double pivmin = tiny;
    double lamr[kMaxRoots], widr[kMaxRoots], lam0r[kMaxRoots];
#pragma unroll
    for (int r = 0; r < kMaxRoots; ++r) {
        lamr[r] = readlane_f64(mylam, (r < nroots ? r : 0) * per);
        widr[r] = readlane_f64(mywid, (r < nroots ? r : 0) * per);
        lam0r[r] = lamr[r];
    }
    double zr[kMaxRoots];
    bool rq_ok = true;
#pragma unroll 1
    for (int pass = 0; pass < 2; ++pass) {
        {
            const int r = (lane >> 1) < nroots ? (lane >> 1) : 0, dir = lane & 1;
            double l = lamr[0];
#pragma unroll
            for (int q = 1; q < kMaxRoots; ++q) l = (r == q) ? lamr[q] : l;
            if (lane < 2 * nroots) {
                auto at = [&](int ii) { return dir ? m - 1 - ii : ii; };
                double D = dd[at(0)] - l;
                for (int ii = 0; ii + 1 < m; ++ii) {
                    const int pos = at(ii), nxt = at(ii + 1), ei = pos < nxt ? pos : nxt;
                    if (fabs(D) < pivmin) D = -pivmin;
                    double rD = __builtin_amdgcn_rcp(D);
                    rD = rD * fma(-D, rD, 2.0);
                    rD = rD * fma(-D, rD, 2.0);
                    const double e = ee[ei], F = e * rD;
                    fD[(dir * kMaxRoots + r) * 32 + pos] = D;
                    fF[(dir * kMaxRoots + r) * 32 + ei] = F;
                    D = (dd[nxt] - l) - F * e;
                }
                fD[(dir * kMaxRoots + r) * 32 + at(m - 1)] = D;
            }
        }
        // twist index (smallest |gamma_i|), z by the two recurrences, per root; component j of root r ends in lane j
#pragma unroll
        for (int r = 0; r < kMaxRoots; ++r) {
            zr[r] = 0.0;
            if (r < nroots) {   // uniform
                const double gs = (j < m) ? fD[r * 32 + j] + fD[(kMaxRoots + r) * 32 + j] - (dd[j] - lamr[r]) : 1.0e300;
                const double g = fabs(gs);
                // argmin over the lanes (ties: the smallest index)
                double best = g;
                best = fmin(best, dpp_move<0xB1>(best));
                best = fmin(best, dpp_move<0x4E>(best));
                best = fmin(best, dpp_move<0x141>(best));
                best = fmin(best, dpp_move<0x140>(best));
                best = fmin(readlane_f64(best, 0), readlane_f64(best, 16));
                const unsigned long long hit = __ballot(g == best && lane < 32);
                const int kt = __ffsll((long long)hit) - 1;
                const double gk = readlane_f64(gs, kt);
                // lane 0 walks up from the twist, lane 1 down (serial chains of at most m steps), through LDS
                if (lane == 0) {
                    double z = 1.0;
                    zb[r * 32 + kt] = 1.0;
                    for (int i = kt - 1; i >= 0; --i) {
                        z = -fF[r * 32 + i] * z;
                        zb[r * 32 + i] = z;
                    }
                } else if (lane == 1) {
                    double z = 1.0;
                    for (int i = kt; i + 1 < m; ++i) {
                        z = -fF[(kMaxRoots + r) * 32 + i] * z;
                        zb[r * 32 + i + 1] = z;
                    }
                }
                const double z = (j < m) ? zb[r * 32 + j] : 0.0;
                const double nrm = sum32m(z * z);
                zr[r] = z / sqrt(nrm);
                const double lnew = lamr[r] + gk / nrm;   // Rayleigh quotient of z (z_k = 1: T z = lam z + gamma_k e_k)
                if (!(fabs(lnew - lam0r[r]) <= widr[r])) rq_ok = false;   // left the bracket (or NaN)
                lamr[r] = lnew;
            }
        }
    }
    if (!rq_ok) return 5;
    {
        // the value is eigenvalue number r: Sturm counts just below and just above it, in double precision (lanes
        // (r, side) = (0 .. nroots-1, 0 / 1))
        const int r = (lane >> 1) < nroots ? (lane >> 1) : 0, side = lane & 1;
        double l = lamr[0];
#pragma unroll
        for (int q = 1; q < kMaxRoots; ++q) l = (r == q) ? lamr[q] : l;
        const double tol = 4.0e-13 * fmax(fabs(glo), fabs(ghi)) + 1.0e-300;
        const double xa = side ? l + tol : l - tol;
        double q = dd[0] - xa;
        int cnt = q < 0.0 ? 1 : 0;
        for (int i = 1; i < m; ++i) {
            if (fabs(q) < pivmin) q = -pivmin;
            const double e = ee[i - 1];
            q = (dd[i] - xa) - e * e / q;
            cnt += q < 0.0 ? 1 : 0;
        }
        const bool good = lane >= 2 * nroots || cnt == r + side;
        if (__ballot(!good) != 0ull) return 6;
    }
    EVC_FEW_STAMP(53);
    // ---- back-transformation z <- H_0 H_1 ... H_{m-3} z
#pragma unroll 1
    for (int k = m - 3; k >= 0; --k) {
        const double vk = Vh[k * 32 + j], bk = bb[k];
#pragma unroll
        for (int r = 0; r < kMaxRoots; ++r)
            if (r < nroots) {
                const double dot = sum32m(vk * zr[r]) * bk;
                zr[r] = fma(-dot, vk, zr[r]);
            }
    }
    EVC_FEW_STAMP(54);
    // ---- verification in the original matrix: |C z - lambda z| <= 1e-11 |C|, |z_r . z_s| <= 1e-12, |z| = 1
    int bad = 0;
    double worst = 0.0;
#pragma unroll
    for (int r = 0; r < kMaxRoots; ++r)
        if (r < nroots) {
            if (h == 0) zb[r * 32 + j] = zr[r];
        }
#pragma unroll
    for (int r = 0; r < kMaxRoots; ++r)
        if (r < nroots) {
            double acc = 0.0;
            for (int c = 0; c < m; ++c) acc = fma((j < m) ? Cs[j * ldc + c] : 0.0, zb[r * 32 + c], acc);
            double res = fabs(acc - lamr[r] * zr[r]);
            res = fmax(res, dpp_move<0xB1>(res));
            res = fmax(res, dpp_move<0x4E>(res));
            res = fmax(res, dpp_move<0x141>(res));
            res = fmax(res, dpp_move<0x140>(res));
            res = fmax(readlane_f64(res, 0), readlane_f64(res, 16));
            const double nn = sum32m(zr[r] * zr[r]);
            worst = fmax(worst, res / anorm);
            if (!(res <= 1.0e-11 * anorm)) bad = bad ? bad : 2;
            if (!(fabs(nn - 1.0) < 1.0e-10)) bad = bad ? bad : 3;
#pragma unroll
            for (int s2 = 0; s2 < kMaxRoots; ++s2)
                if (s2 < r) {
                    const double ov = sum32m(zr[r] * zr[s2]);
                    if (!(fabs(ov) <= 1.0e-12)) bad = bad ? bad : 4;
                }
        }
    EVC_FEW_STAMP(55);
    if (dbg && lane == 0) {
        dbg[0] = worst;
        dbg[1] = lamr[0];
    }
    if (bad) return bad;
#pragma unroll
    for (int r = 0; r < kMaxRoots; ++r)
        if (r < nroots) {
            if (lane == 0) lam[r] = lamr[r];
            if (h == 0 && j < m) Y[r * ldy + j] = zr[r];
        }
    return 0;
}

}  // namespace few
