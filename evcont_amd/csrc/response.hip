// Explicit (tensor-valued) forms of the Loewdin response, for the public building blocks of
// ab_initio_gradients_loewdin.py that return whole tensors:
//   loewdin_trafo_grad            (:41-112)   -> evc_loewdin_trafo_grad      (N,N,N,N)
//   get_derivative_ao_mo_trafo    (:115-134)  -> evc_derivative_ao_mo_trafo  (N,N,A,3)
//   get_one_el_grad               (:155-187)  -> evc_one_el_grad             (N,N,A,3)
//   two_el_grad                   (:190-252)  -> evc_two_el_grad             (A,3)
// The fused energy+force path never materialises these (it uses the adjoint form in
// dense_small.hip); they exist so that scripts calling the pieces get device results too.
#include <string.h>

#include "common.hpp"
#include "kernels.hpp"

namespace evc {

constexpr int kT = 256;

template <typename FA, typename FB, typename FC>
__device__ __forceinline__ void mm16r(int n, FA a, FB b, FC store) {
    const int tk = threadIdx.x & 15, tj = threadIdx.x >> 4;
    for (int i0 = 0; i0 < n; i0 += 16)
        for (int j0 = 0; j0 < n; j0 += 16) {
            const int i = i0 + tj, j = j0 + tk;
            if (i < n && j < n) {
                double c = 0.0;
                for (int k = 0; k < n; ++k) c = fma(a(i, k), b(k, j), c);
                store(i, j, c);
            }
        }
}

// Divided differences of f(s) = s^-1/2 with the reference's 1e-15 eigenvalue guard.
__device__ __forceinline__ double dk_factor(double si, double sj) {
    const bool oi = si > 1.0e-15, oj = sj > 1.0e-15;
    if (oi && oj) {
        const double ri = sqrt(si), rj = sqrt(sj);
        return -1.0 / (ri * rj * (ri + rj));
    }
    if (si != sj) {
        const double fi = oi ? 1.0 / sqrt(si) : 0.0, fj = oj ? 1.0 / sqrt(sj) : 0.0;
        return (fi - fj) / (si - sj);
    }
    return 0.0;
}

// dX[k,l,A,x] = ( U [ (U^T dS[A,x] U) o F ] U^T )[k,l],  one workgroup per (A,x)
__global__ __launch_bounds__(kT) void dx_tensor_kernel(const double *__restrict__ U, const double *__restrict__ s,
                                                       const double *__restrict__ ipovlp,
                                                       const int64_t *__restrict__ aoslices, int n, int natm,
                                                       double *__restrict__ dX) {
    extern __shared__ __align__(16) double sm[];
    double *Us = sm, *M1 = Us + n * n, *M2 = M1 + n * n, *ss = M2 + n * n;
    const int A = blockIdx.x / 3, x = blockIdx.x - 3 * A;
    const int p0 = (int)aoslices[2 * A], p1 = (int)aoslices[2 * A + 1];
    for (int idx = threadIdx.x; idx < n * n; idx += kT) {
        Us[idx] = U[idx];
        const int mu = idx / n, nu = idx - mu * n;
        double v = 0.0;
        if (mu >= p0 && mu < p1) v -= ipovlp[(x * n + mu) * n + nu];
        if (nu >= p0 && nu < p1) v -= ipovlp[(x * n + nu) * n + mu];
        M1[idx] = v;
    }
    if (threadIdx.x < n) ss[threadIdx.x] = s[threadIdx.x];
    __syncthreads();
    mm16r(n, [&](int i, int k) { return Us[k * n + i]; }, [&](int k, int j) { return M1[k * n + j]; },
          [&](int i, int j, double v) { M2[i * n + j] = v; });
    __syncthreads();
    mm16r(n, [&](int i, int k) { return M2[i * n + k]; }, [&](int k, int j) { return Us[k * n + j]; },
          [&](int i, int j, double v) { M1[i * n + j] = v * dk_factor(ss[i], ss[j]); });
    __syncthreads();
    mm16r(n, [&](int i, int k) { return Us[i * n + k]; }, [&](int k, int j) { return M1[k * n + j]; },
          [&](int i, int j, double v) { M2[i * n + j] = v; });
    __syncthreads();
    mm16r(n, [&](int i, int k) { return M2[i * n + k]; }, [&](int k, int j) { return Us[j * n + k]; },
          [&](int i, int j, double v) { dX[((int64_t)(i * n + j) * natm + A) * 3 + x] = v; });
}

// h1_jac[j,n,A,x] = (dX^T h X)[j,n] + (dX^T h X)[n,j] + (X^T dh[A,x] X)[j,n]
__global__ __launch_bounds__(kT) void one_el_grad_kernel(const double *__restrict__ X, const double *__restrict__ h,
                                                         const double *__restrict__ dh, const double *__restrict__ dX,
                                                         int n, int natm, double *__restrict__ out) {
    extern __shared__ __align__(16) double sm[];
    double *Xs = sm, *Hs = Xs + n * n, *Ds = Hs + n * n, *T1 = Ds + n * n, *T2 = T1 + n * n;
    const int A = blockIdx.x / 3, x = blockIdx.x - 3 * A;
    const double *dhp = dh + (int64_t)blockIdx.x * n * n;  // (A,3,n,n)
    for (int idx = threadIdx.x; idx < n * n; idx += kT) {
        Xs[idx] = X[idx];
        Hs[idx] = h[idx];
        Ds[idx] = dX[((int64_t)idx * natm + A) * 3 + x];
    }
    __syncthreads();
    mm16r(n, [&](int i, int k) { return Hs[i * n + k]; }, [&](int k, int j) { return Xs[k * n + j]; },
          [&](int i, int j, double v) { T1[i * n + j] = v; });                   // h X
    __syncthreads();
    mm16r(n, [&](int i, int k) { return Ds[k * n + i]; }, [&](int k, int j) { return T1[k * n + j]; },
          [&](int i, int j, double v) { T2[i * n + j] = v; });                   // dX^T h X
    __syncthreads();
    for (int idx = threadIdx.x; idx < n * n; idx += kT) Hs[idx] = dhp[idx];
    __syncthreads();
    mm16r(n, [&](int i, int k) { return Hs[i * n + k]; }, [&](int k, int j) { return Xs[k * n + j]; },
          [&](int i, int j, double v) { T1[i * n + j] = v; });                   // dh X
    __syncthreads();
    mm16r(n, [&](int i, int k) { return Xs[k * n + i]; }, [&](int k, int j) { return T1[k * n + j]; },
          [&](int i, int j, double v) {
              out[((int64_t)(i * n + j) * natm + A) * 3 + x] = (T2[i * n + j] + T2[j * n + i]) + v;
          });
}

// LG[p,q,a,b] = 1/2 sum_ij U_pi U_qj F_ij (U_ai U_bj + U_bi U_aj); one workgroup per p
__global__ __launch_bounds__(kT) void loewdin_trafo_grad_kernel(const double *__restrict__ U,
                                                                const double *__restrict__ s, int n,
                                                                double *__restrict__ LG) {
    extern __shared__ __align__(16) double sm[];
    double *Us = sm, *Fm = Us + n * n, *Vp = Fm + n * n;
    const int p = blockIdx.x;
    for (int idx = threadIdx.x; idx < n * n; idx += kT) {
        Us[idx] = U[idx];
        const int i = idx / n, j = idx - i * n;
        Fm[idx] = dk_factor(s[i], s[j]);
    }
    __syncthreads();
    // Vp[a][j] = sum_i U_pi U_ai F_ij
    mm16r(n, [&](int a, int i) { return Us[p * n + i] * Us[a * n + i]; }, [&](int i, int j) { return Fm[i * n + j]; },
          [&](int a, int j, double v) { Vp[a * n + j] = v; });
    __syncthreads();
    const int n3 = n * n * n;
    for (int idx = threadIdx.x; idx < n3; idx += kT) {
        const int q = idx / (n * n), r = idx - q * n * n, a = r / n, b = r - a * n;
        double acc = 0.0;
        for (int j = 0; j < n; ++j)
            acc = fma(Us[q * n + j], fma(Vp[a * n + j], Us[b * n + j], Vp[b * n + j] * Us[a * n + j]), acc);
        LG[(int64_t)p * n3 + idx] = 0.5 * acc;
    }
}

// out[A,x] = scale * sum_{ij} T[i,j,A,x] * (transposed ? M[j,i] : M[i,j])
//            - t2scale * sum_{m in A} sum_ch t2part[(m*3+x)*nchunk + ch]        (t2part may be NULL)
__global__ __launch_bounds__(kT) void contract_kernel(const double *__restrict__ T, const double *__restrict__ M,
                                                      int transposed, double scale, int n, int natm,
                                                      const double *__restrict__ t2part, int nchunk, double t2scale,
                                                      const int64_t *__restrict__ aoslices,
                                                      double *__restrict__ out) {
    __shared__ double scr[4];
    const int A = blockIdx.x / 3, x = blockIdx.x - 3 * A;
    double s = 0.0;
    for (int idx = threadIdx.x; idx < n * n; idx += kT) {
        const int i = idx / n, j = idx - i * n;
        s = fma(T[((int64_t)idx * natm + A) * 3 + x], transposed ? M[j * n + i] : M[idx], s);
    }
    s *= scale;
    if (t2part) {
        const int p0 = (int)aoslices[2 * A], p1 = (int)aoslices[2 * A + 1];
        const int cnt = (p1 - p0) * nchunk;
        double t = 0.0;
        for (int e = threadIdx.x; e < cnt; e += kT) {
            const int m_ = p0 + e / nchunk, ch = e - (e / nchunk) * nchunk;
            t += t2part[((int64_t)m_ * 3 + x) * nchunk + ch];
        }
        s -= t2scale * t;
    }
    s = block_sum<4>(s, scr);
    if (threadIdx.x == 0) out[blockIdx.x] = s;
}

}  // namespace evc

using namespace evc;

extern "C" int evc_loewdin_trafo_grad(const double *S, int n, double *LG, double *ws, void *stream) {
    EVC_REQUIRE(S && LG && ws, "evc_loewdin_trafo_grad: null pointer");
    EVC_REQUIRE(n >= 1 && n <= 64, "evc_loewdin_trafo_grad: n=%d out of range 1..64", n);
    hipStream_t st = as_stream(stream);
    double *X = ws, *U = X + n * n, *s = U + n * n;  // ws: 2 n^2 + n doubles
    LoewdinArgs la{};
    la.S = S;
    la.X = X;
    la.U = U;
    la.s = s;
    la.n = n;
    int rc = launch_loewdin(la, 1, st);
    if (rc) return rc;
    static LdsAttr attr;
    if (int rca = allow_dynamic_lds(loewdin_trafo_grad_kernel, attr, 160 * 1024, "loewdin_trafo_grad")) return rca;
    hipLaunchKernelGGL(loewdin_trafo_grad_kernel, dim3(n), dim3(kT), sizeof(double) * 3 * n * n, st, U, s, n, LG);
    EVC_LAUNCH_CHECK("loewdin_trafo_grad");
    return 0;
}

extern "C" int evc_derivative_ao_mo_trafo(const double *S, const double *ipovlp, const int64_t *aoslices, int n,
                                          int natm, double *dX, double *ws, void *stream) {
    EVC_REQUIRE(S && ipovlp && aoslices && dX && ws, "evc_derivative_ao_mo_trafo: null pointer");
    EVC_REQUIRE(n >= 1 && n <= 64 && natm >= 1, "evc_derivative_ao_mo_trafo: n=%d natm=%d out of range", n, natm);
    hipStream_t st = as_stream(stream);
    double *X = ws, *U = X + n * n, *s = U + n * n;
    LoewdinArgs la{};
    la.S = S;
    la.X = X;
    la.U = U;
    la.s = s;
    la.n = n;
    int rc = launch_loewdin(la, 1, st);
    if (rc) return rc;
    static LdsAttr attr;
    if (int rca = allow_dynamic_lds(dx_tensor_kernel, attr, 160 * 1024, "dx_tensor")) return rca;
    hipLaunchKernelGGL(dx_tensor_kernel, dim3(natm * 3), dim3(kT), sizeof(double) * (3 * n * n + n), st, U, s, ipovlp,
                       aoslices, n, natm, dX);
    EVC_LAUNCH_CHECK("dx_tensor");
    return 0;
}

extern "C" int evc_one_el_grad(const double *X, const double *hcore, const double *dhcore, const double *dX, int n,
                               int natm, double *out, void *stream) {
    EVC_REQUIRE(X && hcore && dhcore && dX && out, "evc_one_el_grad: null pointer");
    EVC_REQUIRE(n >= 1 && n <= 60 && natm >= 1, "evc_one_el_grad: n=%d natm=%d out of range (n<=60)", n, natm);
    static LdsAttr attr;
    if (int rca = allow_dynamic_lds(one_el_grad_kernel, attr, 160 * 1024, "one_el_grad")) return rca;
    hipLaunchKernelGGL(one_el_grad_kernel, dim3(natm * 3), dim3(kT), sizeof(double) * 5 * n * n, as_stream(stream), X,
                       hcore, dhcore, dX, n, natm, out);
    EVC_LAUNCH_CHECK("one_el_grad");
    return 0;
}

extern "C" int evc_contract_nnA3(const double *T, const double *M, int transposed, int n, int natm, double *out,
                                 void *stream) {
    EVC_REQUIRE(T && M && out, "evc_contract_nnA3: null pointer");
    EVC_REQUIRE(n >= 1 && natm >= 1, "evc_contract_nnA3: n=%d natm=%d", n, natm);
    hipLaunchKernelGGL(contract_kernel, dim3(natm * 3), dim3(kT), 0, as_stream(stream), T, M, transposed, 1.0, n, natm,
                       (const double *)nullptr, 0, 0.0, (const int64_t *)nullptr, out);
    EVC_LAUNCH_CHECK("contract_nnA3");
    return 0;
}

extern "C" size_t evc_two_el_grad_ws_bytes(int n) {
    if (n < 1 || n > 64) return 0;
    const size_t n2 = (size_t)n * n, n4 = n2 * n2;
    return sizeof(double) * (3 * n4 + (size_t)y2_slabs(n) * n2 + n2 + (size_t)n * 3 * ip1_chunks(n) + 64);
}

extern "C" int evc_two_el_grad(const double *h2_ao, const double *two_rdm, const double *X, const double *dX,
                               const double *ip1, const int64_t *aoslices, int n, int natm, double *out, void *ws,
                               size_t ws_bytes, void *stream) {
    EVC_REQUIRE(h2_ao && two_rdm && X && dX && ip1 && aoslices && out && ws, "evc_two_el_grad: null pointer");
    EVC_REQUIRE(n >= 1 && n <= 64 && natm >= 1, "evc_two_el_grad: n=%d natm=%d out of range", n, natm);
    EVC_REQUIRE(ws_bytes >= evc_two_el_grad_ws_bytes(n), "evc_two_el_grad: workspace too small");
    hipStream_t st = as_stream(stream);
    const size_t n2 = (size_t)n * n, n4 = n2 * n2;
    double *B1 = static_cast<double *>(ws), *B2 = B1 + n4, *K3 = B2 + n4;
    double *y2part = K3 + n4, *y2 = y2part + (size_t)y2_slabs(n) * n2, *t2part = y2 + n2;
    int rc;
    auto qt = [&](const double *src, int ct, double *dst) {
        return launch_quarter_transform(src, 0, X, 0, ct, n, dst, 0, 1, st);
    };
    if ((rc = qt(h2_ao, 0, B1))) return rc;
    if ((rc = qt(B1, 0, B2))) return rc;
    if ((rc = qt(B2, 0, K3))) return rc;
    if ((rc = launch_sym_oao_t(two_rdm, 0, n, B2, 0, 1, st))) return rc;
    if ((rc = launch_y2(B2, K3, n, y2part, 0, 1, st))) return rc;
    if ((rc = qt(two_rdm, 1, B1))) return rc;
    if ((rc = qt(B1, 1, B2))) return rc;
    if ((rc = qt(B2, 1, B1))) return rc;
    if ((rc = qt(B1, 1, B2))) return rc;
    Ip1Args ia;
    memset(&ia, 0, sizeof(ia));
    ia.ip1 = ip1;
    ia.Gao = B2;
    ia.t2part = t2part;
    ia.y2part = y2part;
    ia.y2 = y2;
    ia.n = n;
    ia.natm = 0;
    ia.presym = 0;
    ia.nslab = y2_slabs(n);
    ia.nchunk = ip1_chunks(n);
    if ((rc = launch_ip1_dh(ia, 1, st))) return rc;
    // out[A,x] = sum_ai dX[a,i,A,x] y2[i][a] - sum_{m in A} t2[x,m]
    hipLaunchKernelGGL(contract_kernel, dim3(natm * 3), dim3(kT), 0, st, dX, (const double *)y2, 1, 1.0, n, natm,
                       (const double *)t2part, ip1_chunks(n), 1.0, aoslices, out);
    EVC_LAUNCH_CHECK("two_el_contract");
    return 0;
}
