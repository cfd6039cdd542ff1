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
#include "common.hpp"
#include "kernels.hpp"

namespace evc {

constexpr int kThreads = 256;

// ------------------------------------------------------------------ small LDS matmul
// C[i][j] = sum_k a(i,k) * b(k,j),  i,j < n;  thread (tj,tk) owns i = tj+16*, j = tk+16*.
template <typename FA, typename FB, typename FC>
__device__ __forceinline__ void mm16(int n, FA a, FB b, FC store) {
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
__device__ void jacobi_eigh_lds(double *A, double *V, int m, double *rot, double *red, bool init_v = true) {
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

__device__ __forceinline__ double quad_sum(double v) {
    v += __shfl_xor(v, 1);
    v += __shfl_xor(v, 2);
    return v;
}

__device__ void jacobi_onesided_wave(double *Gc, int m) {
    const int lane = threadIdx.x & 63;
    const int k = lane >> 2, sub = lane & 3;
    const int half = m >> 1;
    const bool active = k < half;
    const int row0 = sub * 8;
    for (int sweep = 0; sweep < 40; ++sweep) {
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
__device__ void jacobi_eigh_wave(double *A, double *V, int m, double shift, bool init_v, double *Gc, double *lam) {
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

// Warm start (EVC_FLAG_WARM_START): `prev` holds the eigenvectors of the previous, nearby problem.  If they
// are orthonormal to 1e-8 (a stale or never-written buffer is not), V <- prev (padded with the identity) and
// A <- V^T A V, which is nearly diagonal, so the sweeps that follow are two or three instead of seven or eight.
// Returns whether the rotation was applied (uniform over the workgroup).  Tmp: n*n doubles of LDS.
__device__ bool warm_start_rotate(double *A, double *V, double *Tmp, int n, int m, const double *__restrict__ prev,
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

// ------------------------------------------------------------------ Loewdin
__global__ __launch_bounds__(kThreads) void loewdin_kernel(LoewdinArgs a) {
    const int n = a.n;
    const int64_t g = blockIdx.x;
    const double *__restrict__ S = a.S + g * a.sS;
    const double *__restrict__ h = a.h ? a.h + g * a.sh : nullptr;
    double *__restrict__ X = a.X + g * a.sws;
    double *__restrict__ U = a.U + g * a.sws;
    double *__restrict__ sv = a.s + g * a.sws;
    double *__restrict__ h1 = a.h1 ? a.h1 + g * a.sws : nullptr;
    extern __shared__ __align__(16) double sm[];
    const int m = (n + 1) & ~1;
    double *A = sm;              // m*m   (later: hcore)
    double *V = A + m * m;       // m*m   (later: T = h X)
    double *Xs = V + m * m;      // m*m   (uses n*n)
    double *rot = Xs + m * m;    // m
    double *red = rot + m;       // 8
    double *f = red + 8;         // m
    double *Gc = f + m;          // kJwMax x kJwPitch (only carved for m <= kJwMax)
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
    const bool warm = a.warm && warm_start_rotate(A, V, Xs, n, m, U, n, red);
    if (m <= kJwMax) jacobi_eigh_wave(A, V, m, 0.0, !warm, Gc, f);
    else jacobi_eigh_lds(A, V, m, rot, red, !warm);
    if (tid < m) {
        const double s = A[tid * m + tid];
        f[tid] = (tid < n && s > 1.0e-15) ? 1.0 / sqrt(s) : 0.0;
        if (tid < n) sv[tid] = s;
    }
    __syncthreads();
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

static size_t jacobi_aux_bytes(int m) {
    return sizeof(double) * (size_t)(2 * m + 8) + 32 +
           (m <= kJwMax ? sizeof(double) * kJwMax * kJwPitch + 16 : 0);
}

int launch_loewdin(const LoewdinArgs &a, int count, hipStream_t st) {
    const int m = (a.n + 1) & ~1;
    const size_t lds = sizeof(double) * (size_t)3 * m * m + jacobi_aux_bytes(m);
    static LdsAttr attr;
    if (int rc = allow_dynamic_lds(loewdin_kernel, attr, 160 * 1024, "loewdin")) return rc;
    hipLaunchKernelGGL(loewdin_kernel, dim3(count), dim3(kThreads), lds, st, a);
    EVC_LAUNCH_CHECK("loewdin");
    return 0;
}

// ------------------------------------------------------------------ subspace solve
__global__ __launch_bounds__(kThreads) void subspace_kernel(SolveArgs a) {
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
        if (a.vstd) a.vstd += g * a.sw;
        if (a.e_shift_dev) a.e_shift = a.e_shift_dev[g];
    }
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
    // kJwMax x kJwPitch doubles for the single-wave eigensolver (only carved for m <= kJwMax), 16-byte aligned
    double *Gc = reinterpret_cast<double *>((reinterpret_cast<uintptr_t>(order + m) + 15) & ~(uintptr_t)15);
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
        H[ia * T + ib] += a.alpha2 * s;
    }
    __syncthreads();
    if (a.Hout)
        for (int idx = tid; idx < T * T; idx += kThreads) a.Hout[idx] = H[idx];
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
    for (int idx = tid; idx < m * m; idx += kThreads) {
        const int i = idx / m, j = idx - i * m;
        double v = 0.0;
        if (i < T && j < T) v = 0.5 * (V[i * m + j] + V[j * m + i]);
        Cm[idx] = v;  // the dummy dimension (odd T) stays decoupled and is skipped below
    }
    __syncthreads();
    // warm start from the standard-form eigenvectors of the previous call (H is free as scratch here)
    const bool warm = a.warm && a.vstd && warm_start_rotate(Cm, V, H, T, m, a.vstd, m, red);
    if (m <= kJwMax) {
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
        jacobi_eigh_wave(Cm, V, m, shift, !warm, Gc, ev);
    } else {
        jacobi_eigh_lds(Cm, V, m, rot, red, !warm);
    }
    if (a.vstd)
        for (int idx = tid; idx < m * m; idx += kThreads) a.vstd[idx] = V[idx];
    // (5) ascending order
    if (tid < T) ev[tid] = Cm[tid * m + tid];
    __syncthreads();
    if (tid < T) {
        int rank = 0;
        const double v = ev[tid];
        for (int j = 0; j < T; ++j) rank += (ev[j] < v || (ev[j] == v && j < tid)) ? 1 : 0;
        order[rank] = tid;
    }
    __syncthreads();
    // (6) back-transform c = L^-T y for the requested roots (thread per root); store into H region
    if (tid < a.nroots) {
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
    // (7) weights of root 0 for the predicted RDMs
    const double *c0 = H;
    if (a.w1)
        for (int idx = tid; idx < T * T; idx += kThreads) {
            const int ia = idx / T;
            a.w1[idx] = c0[ia] * c0[idx - ia * T];
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
            if (a.w2t) a.w2t[r * kMaxBatchG + (int)(blockIdx.x % kMaxBatchG)] = w;
        }
    }
}

int launch_subspace_solve(const SolveArgs &a, int count, hipStream_t st) {
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
__global__ __launch_bounds__(kThreads) void grad_prep_kernel(GradPrepArgs a) {
    extern __shared__ __align__(16) double sm[];
    const int n = a.n;
    {
        const int64_t g = blockIdx.x;
        a.X += g * a.sws;
        a.hcore += g * a.sh;
        a.D += g * a.sD;
        a.Pao += g * a.sws;
        a.Y1 += g * a.sws;
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

int launch_grad_prep(const GradPrepArgs &a, int count, hipStream_t st) {
    const size_t lds = sizeof(double) * (size_t)4 * a.n * a.n;
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
    double *Y = sm;            // n*n
    double *Q = Y + n * n;     // n*n
    double *W = Q + n * n;     // n*n
    double *Us = W + n * n;    // n*n
    double *rs = Us + n * n;   // n   sqrt(s) (0 where guarded)
    double *fs = rs + n;       // n   f(s)
    double *ss = fs + n;       // n   s
    double *t2 = ss + n;       // 3*n
    double *add = t2 + 3 * n;  // 3*natm: scale1 * (term3 + gnuc)
    int *sl = reinterpret_cast<int *>(add + 3 * a.natm);   // 2*natm: AO slices
    const int tid = threadIdx.x;
    // n <= 32: the 3 n^2 overlap derivatives are fetched into registers now and parked in the three product
    // buffers once those are free, so that the per-atom loop at the end runs out of LDS (it is a chain of
    // dependent global loads otherwise: ~2 us per (atom, x) and wave)
    const bool stage_ip = n <= 32;
    double ipf[12];
    if (stage_ip) {
#pragma unroll
        for (int u = 0; u < 12; ++u) {
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
    copy_to_lds(Us, a.U, n * n);
    for (int idx = tid; idx < n * n; idx += kThreads) {
        const int ai = idx / n, i = idx - ai * n;  // Y[a][i]; y2 is stored [i][a]
        Y[idx] = a.Y1[idx] + 0.5 * a.y2[i * n + ai];
    }
    if (tid < n) {
        const double s = a.s[tid];
        const bool ok = s > 1.0e-15;
        ss[tid] = s;
        rs[tid] = ok ? sqrt(s) : 0.0;
        fs[tid] = ok ? 1.0 / sqrt(s) : 0.0;
    }
    for (int idx = tid; idx < 3 * n; idx += kThreads) {
        const int m_ = idx / 3, x = idx - 3 * m_;
        const double *p = a.t2part + ((int64_t)m_ * 3 + x) * a.nchunk;
        double s0 = 0.0, s1 = 0.0;
        int ch = 0;
        for (; ch + 2 <= a.nchunk; ch += 2) {
            s0 += p[ch];
            s1 += p[ch + 1];
        }
        if (ch < a.nchunk) s0 += p[ch];
        t2[x * n + m_] = s0 + s1;
    }
    __syncthreads();
    // Q = U^T Y
    mm16(n, [&](int i, int k) { return Us[k * n + i]; }, [&](int k, int j) { return Y[k * n + j]; },
         [&](int i, int j, double v) { Q[i * n + j] = v; });
    __syncthreads();
    // W = (Q U) o F
    mm16(n, [&](int i, int k) { return Q[i * n + k]; }, [&](int k, int j) { return Us[k * n + j]; },
         [&](int i, int j, double v) {
             double F;
             if (rs[i] > 0.0 && rs[j] > 0.0) F = -1.0 / (rs[i] * rs[j] * (rs[i] + rs[j]));
             else if (ss[i] != ss[j]) F = (fs[i] - fs[j]) / (ss[i] - ss[j]);
             else F = 0.0;
             W[i * n + j] = v * F;
         });
    __syncthreads();
    // Q = U W
    mm16(n, [&](int i, int k) { return Us[i * n + k]; }, [&](int k, int j) { return W[k * n + j]; },
         [&](int i, int j, double v) { Q[i * n + j] = v; });
    __syncthreads();
    // W = Q U^T
    mm16(n, [&](int i, int k) { return Q[i * n + k]; }, [&](int k, int j) { return Us[j * n + k]; },
         [&](int i, int j, double v) { W[i * n + j] = v; });
    __syncthreads();
    // grad[A,x] = - sum_{mu in A} sum_nu ip[x,mu,nu] (W[mu,nu] + W[nu,mu])
    //             - 1/2 sum_{m in A} t2[x][m] + scale1 * (term3 + gnuc)
    if (stage_ip) {  // Y, Q, Us are free now: ip[x] -> {Y, Q, Us}[x]
#pragma unroll
        for (int u = 0; u < 12; ++u) {
            const int idx = tid + kThreads * u;
            if (idx < 3 * n * n) {
                const int x = idx / (n * n);
                (x == 0 ? Y : x == 1 ? Q : Us)[idx - x * n * n] = ipf[u];
            }
        }
        __syncthreads();
    }
    const int lane = tid & 63, wave = tid >> 6;
    for (int ax = wave; ax < a.natm * 3; ax += 4) {
        const int A = ax / 3, x = ax - 3 * A;
        const int p0 = sl[2 * A], p1 = sl[2 * A + 1];
        const double *ipx = x == 0 ? Y : x == 1 ? Q : Us;
        double s = 0.0;
        for (int mu = p0; mu < p1; ++mu)
            for (int nu = lane; nu < n; nu += 64) {
                const double ip = stage_ip ? ipx[mu * n + nu] : a.ipovlp[(x * n + mu) * n + nu];
                s = fma(ip, W[mu * n + nu] + W[nu * n + mu], s);
            }
        s = -s;
        for (int m_ = p0 + lane; m_ < p1; m_ += 64) s -= 0.5 * t2[x * n + m_];
        s = wave_sum(s);
        if (lane == 0) a.grad[ax] = s + add[ax];
    }
}

int launch_grad_final(const GradFinalArgs &a, int count, hipStream_t st) {
    const size_t lds = sizeof(double) * ((size_t)4 * a.n * a.n + 6 * a.n + 3 * (size_t)a.natm) +
                       sizeof(int) * 2 * (size_t)a.natm + 16;
    static LdsAttr attr;
    if (int rc = allow_dynamic_lds(grad_final_kernel, attr, 160 * 1024, "grad_final")) return rc;
    hipLaunchKernelGGL(grad_final_kernel, dim3(count), dim3(kThreads), lds, st, a);
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
