// Single-workgroup dense kernels (everything O(N^3) or O(T^3); matrices live in LDS):
//   K1/K2  Loewdin orthogonalisation: parallel cyclic Jacobi, X = S^-1/2, h1 = X^T h X
//          (electron_integral_utils.py:6-18,135; gradients_loewdin.py:336-338)
//   K6     subspace generalised eigenproblem, LAPACK dsygvd semantics
//          (evcont.py:38-90,157-173) + pair weights (gradients_loewdin.py:343-353)
//   K12/K16 one-body gradient intermediates and the adjoint Loewdin response that folds
//          K10/K11 (gradients_loewdin.py:41-134,155-187,300-303) into three N^3 products.
#include "common.hpp"
#include "kernels.hpp"

namespace evc {

constexpr int kThreads = 256;

// ------------------------------------------------------------------ Jacobi eigensolver (LDS)
// A (m x m, m even, symmetric, both triangles kept) is diagonalised in place; V accumulates the
// rotations (columns = eigenvectors).  Pairs follow the round-robin tournament so the m/2 rotations
// of a step are disjoint; a step is: (i) m/2 lanes compute (c,s); barrier; (ii) every 2x2 block
// (pair k) x (pair k') gets its row AND column rotation in registers and V gets the column
// rotation; barrier.  rot: 2*(m/2) doubles; pq: 2*(m/2) ints; red: 8 doubles.
__device__ void jacobi_eigh_lds(double *A, double *V, int m, double *rot, int *pq, double *red) {
    const int tid = threadIdx.x;
    const int half = m >> 1;
    for (int idx = tid; idx < m * m; idx += kThreads) V[idx] = (idx / m == idx % m) ? 1.0 : 0.0;
    __syncthreads();
    for (int sweep = 0; sweep < 60; ++sweep) {
        double off = 0.0, dg = 0.0;
        for (int idx = tid; idx < m * m; idx += kThreads) {
            const double v = A[idx];
            if (idx / m == idx % m) dg = fma(v, v, dg);
            else off = fma(v, v, off);
        }
        off = block_sum<4>(off, red);
        dg = block_sum<4>(dg, red + 4);
        if (!(off > 1.0e-34 * dg)) break;  // also leaves on NaN
        for (int step = 0; step < m - 1; ++step) {
            if (tid < half) {
                int p = (tid == 0) ? m - 1 : (step + tid) % (m - 1);
                int q = (step + m - 1 - tid) % (m - 1);
                if (p > q) { const int t = p; p = q; q = t; }
                const double apq = A[p * m + q];
                double c = 1.0, s = 0.0;
                if (apq != 0.0) {
                    const double app = A[p * m + p], aqq = A[q * m + q];
                    const double tau = (aqq - app) / (2.0 * apq);
                    const double t = copysign(1.0, tau) / (fabs(tau) + sqrt(fma(tau, tau, 1.0)));
                    c = 1.0 / sqrt(fma(t, t, 1.0));
                    s = t * c;
                }
                rot[2 * tid] = c;
                rot[2 * tid + 1] = s;
                pq[2 * tid] = p;
                pq[2 * tid + 1] = q;
            }
            __syncthreads();
            // A <- J^T A J on the block upper triangle (k <= k2), mirrored to keep A symmetric
            const int nblk = half * (half + 1) / 2;
            for (int bidx = tid; bidx < nblk + m * half; bidx += kThreads) {
                if (bidx < nblk) {
                    int k = 0, rem = bidx;
                    while (rem >= half - k) { rem -= half - k; ++k; }
                    const int k2 = k + rem;
                    const int p = pq[2 * k], q = pq[2 * k + 1], p2 = pq[2 * k2], q2 = pq[2 * k2 + 1];
                    const double c = rot[2 * k], s = rot[2 * k + 1], c2 = rot[2 * k2], s2 = rot[2 * k2 + 1];
                    const double b00 = A[p * m + p2], b01 = A[p * m + q2], b10 = A[q * m + p2], b11 = A[q * m + q2];
                    // rows: (J^T B)
                    const double r00 = c * b00 - s * b10, r01 = c * b01 - s * b11;
                    const double r10 = s * b00 + c * b10, r11 = s * b01 + c * b11;
                    // cols: (. J2)
                    double n00 = c2 * r00 - s2 * r01, n01 = s2 * r00 + c2 * r01;
                    double n10 = c2 * r10 - s2 * r11, n11 = s2 * r10 + c2 * r11;
                    if (k == k2) { n01 = 0.0; n10 = 0.0; }
                    A[p * m + p2] = n00; A[p * m + q2] = n01; A[q * m + p2] = n10; A[q * m + q2] = n11;
                    if (k != k2) {
                        A[p2 * m + p] = n00; A[q2 * m + p] = n01; A[p2 * m + q] = n10; A[q2 * m + q] = n11;
                    }
                } else {
                    const int e = bidx - nblk;
                    const int i = e / half, k = e % half;
                    const int p = pq[2 * k], q = pq[2 * k + 1];
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

// C[i][j] = sum_k opA(i,k) * opB(k,j) for n x n matrices; generic small matmul over the block.
template <typename FA, typename FB, typename FC>
__device__ __forceinline__ void small_mm(int n, FA a, FB b, FC store) {
    for (int idx = threadIdx.x; idx < n * n; idx += kThreads) {
        const int i = idx / n, j = idx % n;
        double acc = 0.0;
        for (int k = 0; k < n; ++k) acc = fma(a(i, k), b(k, j), acc);
        store(i, j, acc);
    }
}

// ------------------------------------------------------------------ Loewdin
__global__ __launch_bounds__(kThreads) void loewdin_kernel(const double *__restrict__ S, const double *__restrict__ h,
                                                           int n, double *__restrict__ X, double *__restrict__ U,
                                                           double *__restrict__ sv, double *__restrict__ h1) {
    extern __shared__ __align__(16) double sm[];
    const int m = (n + 1) & ~1;
    double *A = sm;              // m*m   (later: T = h X)
    double *V = A + m * m;       // m*m
    double *Xs = V + m * m;      // n*n
    double *rot = Xs + m * m;    // m
    double *red = rot + m;       // 8
    double *f = red + 8;         // m
    int *pq = reinterpret_cast<int *>(f + m);  // m ints
    const int tid = threadIdx.x;
    for (int idx = tid; idx < m * m; idx += kThreads) {
        const int i = idx / m, j = idx % m;
        // LAPACK's eigh reads one triangle; numpy.linalg.eigh uses the lower one.
        double v = 0.0;
        if (i < n && j < n) v = (i >= j) ? S[i * n + j] : S[j * n + i];
        A[idx] = v;
    }
    __syncthreads();
    jacobi_eigh_lds(A, V, m, rot, pq, red);
    if (tid < m) {
        const double s = A[tid * m + tid];
        f[tid] = (tid < n && s > 1.0e-15) ? 1.0 / sqrt(s) : 0.0;
        if (tid < n) sv[tid] = s;
    }
    __syncthreads();
    // X = V diag(f) V^T  (dummy column, if any, has f = 0)
    for (int idx = tid; idx < n * n; idx += kThreads) {
        const int a = idx / n, b = idx % n;
        double acc = 0.0;
        for (int i = 0; i < m; ++i) acc = fma(V[a * m + i] * f[i], V[b * m + i], acc);
        Xs[idx] = acc;
        X[idx] = acc;
        U[idx] = V[a * m + b];
    }
    __syncthreads();
    if (h && h1) {
        // T = h X (into A), h1 = X^T T
        small_mm(n, [&](int i, int k) { return h[i * n + k]; }, [&](int k, int j) { return Xs[k * n + j]; },
                 [&](int i, int j, double v) { A[i * n + j] = v; });
        __syncthreads();
        small_mm(n, [&](int i, int k) { return Xs[k * n + i]; }, [&](int k, int j) { return A[k * n + j]; },
                 [&](int i, int j, double v) { h1[i * n + j] = v; });
    }
}

static size_t loewdin_lds_bytes(int n) {
    const int m = (n + 1) & ~1;
    return sizeof(double) * ((size_t)3 * m * m + m + 8 + m) + sizeof(int) * m + 16;
}

int launch_loewdin(const double *S, const double *hcore, int n, double *X, double *U, double *s, double *h1,
                   hipStream_t st) {
    const size_t lds = loewdin_lds_bytes(n);
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(loewdin_kernel),
                            hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        attr_set = true;
    }
    hipLaunchKernelGGL(loewdin_kernel, dim3(1), dim3(kThreads), lds, st, S, hcore, n, X, U, s, h1);
    EVC_LAUNCH_CHECK("loewdin");
    return 0;
}

// ------------------------------------------------------------------ subspace solve
__global__ __launch_bounds__(kThreads) void subspace_kernel(SolveArgs a) {
    extern __shared__ __align__(16) double sm[];
    const int T = a.T;
    const int m = (T + 1) & ~1;
    double *H = sm;             // T*T  assembled H, later Z, later C (m*m)
    double *L = H + m * m;      // T*T  Cholesky factor (lower)
    double *Cm = L + m * m;     // m*m  standard-form matrix
    double *V = Cm + m * m;     // m*m
    double *rot = V + m * m;    // m
    double *red = rot + m;      // 8
    double *ev = red + 8;       // m
    int *pq = reinterpret_cast<int *>(ev + m);  // m
    int *order = pq + m;                        // m
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int64_t P = (int64_t)T * (T + 1) / 2;
    const bool pairs = (a.layout == EVC_LAYOUT_PAIR5 || a.layout == EVC_LAYOUT_PACK2);

    // (1) one-body rows: H = alpha1 * sum_sp h1part
    for (int r = wave; r < T * T; r += 4) {
        double s = 0.0;
        for (int k = lane; k < a.nsp1; k += 64) s += a.h1part[(int64_t)r * a.nsp1 + k];
        s = wave_sum(s);
        if (lane == 0) H[r] = a.alpha1 * s;
    }
    __syncthreads();
    // (2) two-body rows, placed as the reference does (evcont.py:41-68)
    const int64_t rows2 = pairs ? P : (int64_t)T * T;
    for (int64_t r = wave; r < rows2; r += 4) {
        double s = 0.0;
        for (int k = lane; k < a.nsp2; k += 64) s += a.h2part[r * a.nsp2 + k];
        s = wave_sum(s);
        if (lane == 0) {
            int ia, ib;
            if (pairs) {
                ia = (int)tri_row(r);
                ib = (int)(r - (int64_t)ia * (ia + 1) / 2);
            } else {
                ia = (int)(r / T);
                ib = (int)(r % T);
            }
            H[ia * T + ib] += a.alpha2 * s;
        }
    }
    __syncthreads();
    if (a.Hout)
        for (int idx = tid; idx < T * T; idx += kThreads) a.Hout[idx] = H[idx];
    // (3) Cholesky of S (lower triangle, as dpotrf('L'))
    for (int idx = tid; idx < T * T; idx += kThreads) {
        const int i = idx / T, j = idx % T;
        L[idx] = (i >= j) ? a.S[idx] : 0.0;
    }
    __syncthreads();
    for (int j = 0; j < T; ++j) {
        if (tid == 0) L[j * T + j] = sqrt(L[j * T + j]);
        __syncthreads();
        const double d = L[j * T + j];
        for (int i = j + 1 + tid; i < T; i += kThreads) L[i * T + j] /= d;
        __syncthreads();
        const int rem = T - j - 1;
        for (int idx = tid; idx < rem * rem; idx += kThreads) {
            const int i = j + 1 + idx / rem, k = j + 1 + idx % rem;
            if (k <= i) L[i * T + k] -= L[i * T + j] * L[k * T + j];
        }
        __syncthreads();
    }
    // (4) C = L^-1 Hsym L^-T, Hsym from the LOWER triangle of H (dsygst).
    //     thread j solves L z = Hsym[:,j]  (column j), result stored in Cm[:,j]
    if (tid < T) {
        const int j = tid;
        for (int i = 0; i < T; ++i) {
            double v = (i >= j) ? H[i * T + j] : H[j * T + i];
            for (int k = 0; k < i; ++k) v -= L[i * T + k] * Cm[k * m + j];
            Cm[i * m + j] = v / L[i * T + i];
        }
    }
    __syncthreads();
    //     thread i solves L w = Z[i,:]^T (row i), result is row i of C -> store into H region (m*m)
    if (tid < T) {
        const int i = tid;
        for (int j = 0; j < T; ++j) {
            double v = Cm[i * m + j];
            for (int k = 0; k < j; ++k) v -= L[j * T + k] * V[i * m + k];
            V[i * m + j] = v / L[j * T + j];
        }
    }
    __syncthreads();
    for (int idx = tid; idx < m * m; idx += kThreads) {
        const int i = idx / m, j = idx % m;
        double v = 0.0;
        if (i < T && j < T) v = 0.5 * (V[i * m + j] + V[j * m + i]);
        Cm[idx] = v;  // the dummy dimension (odd T) stays decoupled and is skipped below
    }
    __syncthreads();
    jacobi_eigh_lds(Cm, V, m, rot, pq, red);
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
            for (int k = i + 1; k < T; ++k) v -= L[k * T + i] * c[k];
            c[i] = v / L[i * T + i];
        }
        a.evals[tid] = ev[col] + a.e_shift;
    }
    __syncthreads();
    for (int idx = tid; idx < a.nroots * T; idx += kThreads) a.evecs[idx] = H[idx];
    // (7) weights of root 0 for the predicted RDMs
    const double *c0 = H;
    if (a.w1)
        for (int idx = tid; idx < T * T; idx += kThreads) a.w1[idx] = c0[idx / T] * c0[idx % T];
    if (a.w2) {
        for (int64_t r = tid; r < a.w2_count; r += kThreads) {
            const int64_t g = r + a.w2_offset;
            double w;
            if (pairs) {
                const int ia = (int)tri_row(g), ib = (int)(g - (int64_t)ia * (ia + 1) / 2);
                w = (ia == ib) ? c0[ia] * c0[ia] : 2.0 * c0[ia] * c0[ib];
            } else {
                w = c0[g / T] * c0[g % T];
            }
            a.w2[r] = w;
        }
    }
}

int launch_subspace_solve(const SolveArgs &a, hipStream_t st) {
    const int m = (a.T + 1) & ~1;
    const size_t lds = sizeof(double) * ((size_t)4 * m * m + 2 * m + 8) + sizeof(int) * 2 * m + 16;
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(subspace_kernel),
                            hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        attr_set = true;
    }
    hipLaunchKernelGGL(subspace_kernel, dim3(1), dim3(kThreads), lds, st, a);
    EVC_LAUNCH_CHECK("subspace_solve");
    return 0;
}

// ------------------------------------------------------------------ gradient prep
// Pao = X D X^T ; Y1 = scale1 * hcore X (D + D^T)
__global__ __launch_bounds__(kThreads) void grad_prep_kernel(GradPrepArgs a) {
    extern __shared__ __align__(16) double sm[];
    const int n = a.n;
    double *T1 = sm;           // X D
    double *T2 = T1 + n * n;   // X (D + D^T)
    const double *X = a.X, *D = a.D, *h = a.hcore;
    small_mm(n, [&](int i, int k) { return X[i * n + k]; }, [&](int k, int j) { return D[k * n + j]; },
             [&](int i, int j, double v) { T1[i * n + j] = v; });
    small_mm(n, [&](int i, int k) { return X[i * n + k]; },
             [&](int k, int j) { return D[k * n + j] + D[j * n + k]; },
             [&](int i, int j, double v) { T2[i * n + j] = v; });
    __syncthreads();
    small_mm(n, [&](int i, int k) { return T1[i * n + k]; }, [&](int k, int j) { return X[j * n + k]; },
             [&](int i, int j, double v) { a.Pao[i * n + j] = v; });
    small_mm(n, [&](int i, int k) { return h[i * n + k]; }, [&](int k, int j) { return T2[k * n + j]; },
             [&](int i, int j, double v) { a.Y1[i * n + j] = a.scale1 * v; });
}

int launch_grad_prep(const GradPrepArgs &a, hipStream_t st) {
    const size_t lds = sizeof(double) * (size_t)2 * a.n * a.n;
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(grad_prep_kernel),
                            hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        attr_set = true;
    }
    hipLaunchKernelGGL(grad_prep_kernel, dim3(1), dim3(kThreads), lds, st, a);
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
    double *Y = sm;            // n*n
    double *Q = Y + n * n;     // n*n
    double *W = Q + n * n;     // n*n
    double *rs = W + n * n;    // n   sqrt(s) (0 where guarded)
    double *fs = rs + n;       // n   f(s)
    double *t2 = fs + n;       // 3*n
    const int tid = threadIdx.x;
    const double *U = a.U;
    for (int idx = tid; idx < n * n; idx += kThreads) {
        const int ai = idx / n, i = idx % n;  // Y[a][i]
        double y2 = 0.0;
        for (int sl = 0; sl < a.nslab; ++sl) y2 += a.y2part[(int64_t)sl * n * n + i * n + ai];
        Y[idx] = a.Y1[idx] + 0.5 * y2;
    }
    if (tid < n) {
        const double s = a.s[tid];
        const bool ok = s > 1.0e-15;
        rs[tid] = ok ? sqrt(s) : 0.0;
        fs[tid] = ok ? 1.0 / sqrt(s) : 0.0;
    }
    for (int idx = tid; idx < 3 * n; idx += kThreads) {
        const int m_ = idx / 3, x = idx % 3;
        double s = 0.0;
        for (int ch = 0; ch < a.nchunk; ++ch) s += a.t2part[((int64_t)m_ * 3 + x) * a.nchunk + ch];
        t2[x * n + m_] = s;
    }
    __syncthreads();
    // Q = U^T Y
    small_mm(n, [&](int i, int k) { return U[k * n + i]; }, [&](int k, int j) { return Y[k * n + j]; },
             [&](int i, int j, double v) { Q[i * n + j] = v; });
    __syncthreads();
    // W = (Q U) o F
    small_mm(n, [&](int i, int k) { return Q[i * n + k]; }, [&](int k, int j) { return U[k * n + j]; },
             [&](int i, int j, double v) {
                 double F;
                 const double si = a.s[i], sj = a.s[j];
                 if (rs[i] > 0.0 && rs[j] > 0.0) F = -1.0 / (rs[i] * rs[j] * (rs[i] + rs[j]));
                 else if (si != sj) F = (fs[i] - fs[j]) / (si - sj);
                 else F = 0.0;
                 W[i * n + j] = v * F;
             });
    __syncthreads();
    // Q = U W
    small_mm(n, [&](int i, int k) { return U[i * n + k]; }, [&](int k, int j) { return W[k * n + j]; },
             [&](int i, int j, double v) { Q[i * n + j] = v; });
    __syncthreads();
    // W = Q U^T
    small_mm(n, [&](int i, int k) { return Q[i * n + k]; }, [&](int k, int j) { return U[j * n + k]; },
             [&](int i, int j, double v) { W[i * n + j] = v; });
    __syncthreads();
    // grad[A,x] = - sum_{mu in A} sum_nu ip[x,mu,nu] (W[mu,nu] + W[nu,mu])
    //             - 1/2 sum_{m in A} t2[x][m] + scale1 * (term3 + gnuc)
    const int lane = tid & 63, wave = tid >> 6;
    for (int ax = wave; ax < a.natm * 3; ax += 4) {
        const int A = ax / 3, x = ax % 3;
        const int p0 = (int)a.aoslices[2 * A], p1 = (int)a.aoslices[2 * A + 1];
        double s = 0.0;
        for (int e = lane; e < (p1 - p0) * n; e += 64) {
            const int mu = p0 + e / n, nu = e % n;
            s = fma(a.ipovlp[(x * n + mu) * n + nu], W[mu * n + nu] + W[nu * n + mu], s);
        }
        s = -s;
        for (int m_ = p0 + lane; m_ < p1; m_ += 64) s -= 0.5 * t2[x * n + m_];
        s = wave_sum(s);
        if (lane == 0) {
            double g = s;
            if (a.scale1 != 0.0) {
                g += a.scale1 * a.term3[ax];
                if (a.gnuc) g += a.scale1 * a.gnuc[ax];
            }
            a.grad[ax] = g;
        }
    }
}

int launch_grad_final(const GradFinalArgs &a, hipStream_t st) {
    const size_t lds = sizeof(double) * ((size_t)3 * a.n * a.n + 5 * a.n);
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(grad_final_kernel),
                            hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        attr_set = true;
    }
    hipLaunchKernelGGL(grad_final_kernel, dim3(1), dim3(kThreads), lds, st, a);
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
    return launch_loewdin(S, hcore, n, X, U, s, h1, as_stream(stream));
}
