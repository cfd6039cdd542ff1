// Internal launch interfaces shared by the translation units of libevcont_hip.so.
#pragma once
#include "common.hpp"

namespace evc {

// ---- gemv_stream.hip ---------------------------------------------------------------
struct RowProblem {
    const double *A;   // (rows, ld)
    const double *v;   // (cols)
    double *partial;   // (rows, nspans)
    int64_t rows, cols, ld;
    int nspans, cps, nblocks;
};
struct GemvRowsLaunch {
    RowProblem p[2];
    int nblk0;
};
struct ColProblem {
    const double *A;  // (rows, ld)
    const double *w;  // (rows)
    double *out;      // (cols)
    int64_t rows, cols, ld;
};
struct GemvColsLaunch {
    ColProblem p[2];
    int nblk0;
};
void plan_rows(RowProblem &P);
size_t rows_ws_doubles(int64_t rows, int64_t cols);
int launch_gemv_rows(RowProblem p0, RowProblem p1, hipStream_t st);
int launch_gemv_cols(ColProblem p0, ColProblem p1, hipStream_t st);

// ---- transform.hip -----------------------------------------------------------------
int launch_quarter_transform(const double *in, const double *C, int c_transposed, int n, double *out,
                             hipStream_t st);
int launch_pack(const double *h2, int n, double diag_mult, double *out, int64_t out_len, hipStream_t st);
int launch_unpack(const double *packed, int n, double *out, hipStream_t st);
// Gs^T[jkl][i] = G[i,j,k,l] + G[j,i,k,l] + G[l,k,j,i] + G[k,l,i,j]   (gradients_loewdin.py:213-215)
int launch_sym_oao_t(const double *G, int n, double *out, hipStream_t st);
// partial[b][i][a] = sum_{k in slab b} GsT[k][i] * K3[k][a]   (k = jkl);  returns #slabs via nslabs
int y2_slabs(int n);
int launch_y2(const double *GsT, const double *K3, int n, double *partial, hipStream_t st);
// ip1 contraction with on-the-fly AO symmetrisation (gradients_loewdin.py:234-252) + dhcore:P_ao dots
int ip1_chunks(int n);
// ... and the fixed-order sum of the Y2 slabs (third block family of the same launch)
int launch_ip1_dh(const double *ip1, const double *Gao, int n, double *t2_partial, const double *dhcore,
                  const double *Pao, int natm, double *term3, const double *y2part, int nslab, double *y2,
                  hipStream_t st);

// ---- dense_small.hip ---------------------------------------------------------------
int launch_loewdin(const double *S, const double *hcore, int n, double *X, double *U, double *s, double *h1,
                   hipStream_t st);
struct SolveArgs {
    const double *h1part;  // (T*T, nsp1) partial sums of the one-body rows
    int nsp1;
    double alpha1;
    const double *h2part;  // (rows2_total, nsp2) partial sums of the two-body rows
    int nsp2;
    double alpha2;
    const double *S;  // (T,T)
    int T, layout, nroots;
    double e_shift;
    double *evals, *evecs, *w2, *w1, *Hout;
    int64_t w2_offset, w2_count;  // slice of the global weight vector to write (multi-GPU)
};
int launch_subspace_solve(const SolveArgs &a, hipStream_t st);
struct GradPrepArgs {
    int n;
    const double *X, *hcore, *D;  // D = predicted 1-RDM (n,n)
    double *Pao;                  // X D X^T
    double *Y1;                   // hcore X (D + D^T)
    double scale1;                // 1 on the rank that owns the one-body part, else 0
};
int launch_grad_prep(const GradPrepArgs &a, hipStream_t st);
struct GradFinalArgs {
    int n, natm;
    const double *U, *s;          // eigen-decomposition of S_AO
    const double *Y1;             // (n,n) [a][i]
    const double *y2;             // (n, n) as [i][a]  (slabs already summed)
    const double *ipovlp;         // (3,n,n)
    const int64_t *aoslices;      // (A,2)
    const double *t2part;         // (n, 3, nchunk) partial sums of T2diag[x,m]
    int nchunk;
    const double *term3;          // (A*3)
    const double *gnuc;           // (A,3) or NULL
    double scale1;                // 1: include term3 + gnuc
    double *grad;                 // (A,3)
};
int launch_grad_final(const GradFinalArgs &a, hipStream_t st);

}  // namespace evc
