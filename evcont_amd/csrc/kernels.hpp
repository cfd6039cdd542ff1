// Internal launch interfaces shared by the translation units of libevcont_hip.so.
//
// Batching: every kernel of the per-geometry pipeline carries a batch index (blockIdx.y for the
// multi-workgroup kernels, blockIdx.x for the single-workgroup ones).  A pointer `p` that belongs
// to geometry 0 is paired with a stride `sp` in doubles; geometry g uses p + g*sp.  Workspace
// buffers all share one stride (the per-geometry workspace size), inputs/outputs have their own.
#pragma once
#include "common.hpp"

namespace evc {

constexpr int kMaxOrbitals = 96;   // N of the fused pipeline (quarter transforms: 16-wide tiles up to 96 padded columns)
constexpr int kMaxBatchG = 32;  // geometries contracted per pass of the streaming kernels (matrix-core variants)

// ---- gemv_stream.hip ---------------------------------------------------------------
struct RowProblem {
    const double *A;   // (rows, ld)                       shared by the batch
    const double *v;   // (cols)            + g*vstride
    double *partial;   // (nspans, rows)    + g*pstride
    int64_t rows, cols, ld, vstride, pstride;
    int nspans, nblocks;
    int64_t span_cols;  // columns per span (a multiple of 32; the last span may be shorter): span s = [s, s+1) * span_cols
    int lds_plan;       // spans planned for the LDS-staged matrix-core kernel (gemv_lds.hip: plan_rows_lds)
};
struct GemvRowsLaunch {
    RowProblem p[2];
    int nblk0;
    int nblk1;  // matrix-core variant: blocks of p[1] (padded to a multiple of 8), dispatched first
    // matrix-core variant: the 16-row tiles of p[k] are dealt to nrg[k] balanced row groups, the first
    // trem[k] of which have tpg[k] + 1 tiles and the others tpg[k]
    int tpg[2], trem[2], nrg[2];
};
struct ColProblem {
    const double *A;  // (rows, ld)                        shared by the batch
    const double *w;  // (rows)             + g*wstride
    const double *wt; // (rows, kMaxBatchG) + (g - g % kMaxBatchG)*wstride: the same weights of a group of geometries,
                      // transposed (geometry g in column g % kMaxBatchG), or NULL
    double *out;      // (cols)             + g*ostride
    int64_t rows, cols, ld, wstride, ostride;
    double *part;     // (kColSlabs, ld) + g*pstride or NULL: scratch for the row-slab form of a NARROW problem with many
    int64_t pstride;  // rows (the one-body t-RDM of a large training set: T^2 rows of N^2 columns)
};
constexpr int kColSlabs = 64;   // row slabs of that form
struct GemvColsLaunch {
    ColProblem p[2];
    int nblk0;
};
// gemv_mfma.hip: matrix-core variants, G <= kMaxBatchG geometries [g0, g0+G) per launch
int launch_gemv_rows_mfma(const GemvRowsLaunch &L, int g0, int G, int tiles, hipStream_t st);
int launch_gemv_cols_mfma(GemvColsLaunch L, int g0, int G, hipStream_t st);
void plan_rows(RowProblem &P, bool batched);
// gemv_lds.hip: the LDS-staged rows kernel for groups of 12 .. 32 geometries and its span plan
bool rows_lds_applicable(const RowProblem &p0, const RowProblem &p1);
void plan_rows_lds(RowProblem &p0, RowProblem &p1);
int rows_max_spans(const RowProblem &P, bool small);
int launch_gemv_rows_lds(const GemvRowsLaunch &L, int g0, int G, hipStream_t st);
int rows_lds_max_g(const RowProblem &p0, const RowProblem &p1);
int cols_lds_mode(const ColProblem &p0, const ColProblem &p1, int G);
int launch_gemv_cols_lds(const GemvColsLaunch &L, int g0, int G, hipStream_t st);
int launch_gemv_cols_lds_slab(const GemvColsLaunch &L, int g0, int G, hipStream_t st);
bool rows_groups_all_mfma(int count);   // gemv_stream.hip: every group of a batch of `count` runs on the matrix cores
size_t rows_ws_doubles(int64_t rows, int64_t cols);
// `count` geometries; launched in groups of up to kMaxBatchG that share one read of A.
int launch_gemv_rows(RowProblem p0, RowProblem p1, int count, hipStream_t st);
int launch_gemv_cols(ColProblem p0, ColProblem p1, int count, hipStream_t st);

// ---- transform.hip, pack.hip, y2.hip, ip1.hip ----------------------------------------
int launch_quarter_transform(const double *in, int64_t sin, const double *C, int64_t sC, int c_transposed, int n,
                             double *out, int64_t sout, int count, hipStream_t st);
// Two quarter steps fused (n <= 32): out[r'][s'][p][q] = sum_rs in[p][q][r][s] C[r][r'] C[s][s'].
struct PairTransformArgs {
    const double *in;   // (n^4)   + g*sin
    const double *C;    // (n,n)   + g*sC
    double *out;        // (n^4) full result or NULL                          + g*sout
    double *packed;     // packed lower triangle (diag * diag_mult) or NULL   + g*spacked
    double *k3;         // half result K3[s'][p][q][r] = sum_s in[p][q][r][s] C[s][s'] or NULL  + g*sk3
    int64_t sin, sC, sout, spacked, sk3, packed_len;
    double diag_mult;
    int ct, n;
    int tiles_per_wg;   // set by launch_pair_transform: consecutive 8-wide q tiles per workgroup
    int sym8;           // `packed` is the 8-fold compressed vector (EVC_LAYOUT_SYM8), multiplicities folded in
    int lead_sym;       // in[p][q][..] = in[q][p][..]: only q <= p (whole 8-wide q tiles) is computed and stored
    int in_lower;       // in[p][q][r][s] = in[p][q][s][r] and only r >= s is valid (output of a lead_sym step)
    int rs_lower;       // only the rows (r',s'), s' <= r', of the result are needed (out rows / packed sym8 vector)
    int out_pairs;      // (with lead_sym and rs_lower) `out` is the dense (pair, pair) matrix out[tri(r',s')][tri(p,q)]
                        // instead of rows of an N^4 tensor (multiplicities are the consumers' business)
    int in_pairs;       // (with lead_sym) `in` is such a matrix: in[tri(p,q)][tri(r,s)]
    int in_ld, out_ld;  // row pitch (doubles) of the dense (pair, pair) operand / result; 0: n(n+1)/2 (the caller's s4 `int2e`);
                        // the pipeline's own intermediates use pair_ld(n): rows start on 128-byte lines
};
// Row pitch of the pipeline's dense (pair, pair) intermediates: n(n+1)/2 rounded up to 16 doubles.
__host__ __device__ inline int pair_ld(int n) { return (n * (n + 1) / 2 + 15) & ~15; }
constexpr int kPairTransformMaxN = 32;
// one packed operand row of the pair kernels in LDS (transform.hip pt_kernel, y2.hip y2_fused_kernel)
constexpr int kPtRawMax = (kPairTransformMaxN * (kPairTransformMaxN + 1) / 2 + 1 + 127) / 128;   // double2 per lane: 5
constexpr int kPtRowLen = kPtRawMax * 128 + 4;   // + two zero slots (padding fragments), 16-byte multiple
int launch_pair_transform(const PairTransformArgs &a, int count, hipStream_t st);
// pair_dma.hip: the fully symmetric dense (pair, pair) -> dense (pair, pair) step, 16 < n <= 30, operand rows by LDS-DMA
bool pair_transform_dma_applicable(const PairTransformArgs &a, int count);
int launch_pair_transform_dma(const PairTransformArgs &a, int count, hipStream_t st);
// ... and the fused Y2 contraction with its two operand rows by LDS-DMA (same slabs as launch_y2_fused)
bool y2_dma_applicable(int n);
int launch_y2_dma(const double *SB, const double *M1, const double *X, int64_t sX, int n, double *partial, int64_t sws,
                  int count, int slabs, int tiles_per_wg, int ppt, hipStream_t st);
int launch_pack(const double *h2, int64_t sh2, int n, double diag_mult, double *out, int64_t sout, int64_t out_len,
                int count, hipStream_t st);
// 8-fold compressed vector of a tensor with the index symmetries of real two-electron integrals:
//   out[tri(u,v)] = h2[i,j,k,l] * (u == v ? diag_mult : 1) * (i != j ? 2 : 1) * (k != l ? 2 : 1),
//   u = tri(i,j) (i >= j), v = tri(k,l) (k >= l), u >= v
int launch_pack_sym8(const double *h2, int64_t sh2, int n, double diag_mult, double *out, int64_t sout,
                     int64_t out_len, int count, hipStream_t st);
int launch_unpack(const double *packed, int64_t sp, int n, double *out, int64_t sout, int count, hipStream_t st);
// Gs^T[jkl][i] = G[i,j,k,l] + G[j,i,k,l] + G[l,k,j,i] + G[k,l,i,j]   (gradients_loewdin.py:213-215)
int launch_sym_oao_t(const double *G, int64_t sG, int n, double *out, int64_t sout, int count, hipStream_t st);
// Packed fast path: from the packed predicted 2-RDM p (pair symmetric by construction) write in one pass
//   GsT[jkl][i] = 2 p(ij,kl) + p(ji,kl) + p(ji,lk)        (= the OAO symmetrisation above, transposed)
//   SB[i,j,k,l] = 2 (p(ij,kl) + p(ji,lk))                  (= AO-type symmetrisation, gradients_loewdin.py:238-240,
//                                                             applied BEFORE the OAO->AO rotation, with which it commutes)
//   G[i,j,k,l]  = p(ij,kl)                                 (optional: the unpacked 2-RDM, eiu:69-88)
int launch_unpack_sym(const double *packed, int64_t sp, int n, double *GsT, double *SB, int64_t sws, double *G,
                      int64_t sG, int count, hipStream_t st);
// EVC_LAYOUT_SYM8: `packed` is the 8-fold compressed vector p8 of a fully symmetric 2-RDM:
//   SB[i,j,k,l] = 4 p8(ijkl) (only for i >= j, l <= k when lead_half = 1; lead_half = 2: the dense (pair, pair)
//   matrix SB[tri(i,j)][tri(k,l)] instead), G[i,j,k,l] = p8(ijkl) (optional)
int launch_unpack8(const double *packed, int64_t sp, int n, double *SB, int64_t sws, double *G, int64_t sG, int count,
                   int lead_half, hipStream_t st);
// partial[b][i][a] = sum_{k in slab b} SB[i][k] * K3[k][a]: the Y2 contraction with the row-major operand
int launch_y2_sb(const double *SB, const double *K3, int n, double *partial, int64_t sws, int count, hipStream_t st);
// Y2 without K3 (symmetric pipeline): the half-transformed integrals are recomputed from the dense (pair, pair)
// intermediate `M1` of the first pair step (kept by the energy phase) inside the contraction,
//   Y2[i][a] = sum_v mult(v) sum_j SB[v][tri(i,j)] (M1_v X)[a][j];
// writes y2_fused_slabs(n, count) partial (n, n) matrices per geometry (slabs as launch_y2's); SB and M1 at the pitch
// pair_ld(n)
int launch_y2_fused(const double *SB, const double *M1, const double *X, int64_t sX, int n, double *partial,
                    int64_t sws, int count, hipStream_t st);
int y2_fused_slabs(int n, int count);
bool y2_fused_available(int n);
// partial[b][i][a] = sum_{k in slab b} GsT[k][i] * K3[k][a]   (k = jkl)
int y2_slabs(int n);
int y2_slab_capacity(int n);   // slabs the pipeline's partial buffer holds (>= y2_slabs)
int launch_y2(const double *GsT, const double *K3, int n, double *partial, int64_t sws, int count, hipStream_t st);
// ip1 contraction with on-the-fly AO symmetrisation (gradients_loewdin.py:234-252), dhcore:P_ao dots
// and the fixed-order sum of the Y2 slabs (three block families of one launch)
int ip1_chunks(int n);
struct Ip1Args {
    const double *ip1;     // (3,n^4)      + g*sip1
    const double *Gao;     // (n^4)        + g*sws
    double *t2part;        // (n,3,nchunk) + g*sws
    const double *dh;      // (A,3,n,n)    + g*sdh      (may be NULL with natm = 0)
    const double *Pao;     // (n,n)        + g*sws
    double *term3;         // (A*3)        + g*sws
    const double *y2part;  // (nslab,n,n)  + g*sws
    double *y2;            // (n,n)        + g*sws
    int64_t sip1, sdh, sws;
    int n, natm, nslab, nchunk;
    int presym;            // Gao already carries the 4-fold AO symmetrisation (packed fast path)
    int fold_cd;           // (with presym) Gao[m,b,c,d] is symmetric in c <-> d and in m <-> b and only valid for
                           // d <= c, b <= m
    int ip1_s2kl;          // (with fold_cd) ip1 is (3,n,n,n(n+1)/2): packed in its last two indices, c >= d
                           // (EVC_FLAG_IP1_S2KL); sip1 is the packed size and Gao the dense (pair, pair) matrix
                           // Gao[tri(m,b)][tri(c,d)] at the pitch pair_ld(n), WITHOUT the multiplicity of (c,d): the
                           // dot weighs it (2 for c != d)
};
int launch_ip1_dh(const Ip1Args &a, int count, hipStream_t st);

// ---- dense_small.hip ---------------------------------------------------------------
// ---- pair64.hip: the symmetric pipeline's pair step and Y2 for 32 < n <= 64 ----------------------------
bool pair64_applicable(const PairTransformArgs &a);
int launch_pair_transform64(const PairTransformArgs &a, int count, hipStream_t st);
bool y2_64_applicable(int n);
int y2_64_slabs(int n, int count);
int launch_y2_64(const double *SB, const double *M1, const double *X, int64_t sX, int n, double *partial, int64_t sws,
                 int count, hipStream_t st);
struct LoewdinArgs {
    const double *S, *h;   // + g*sS, + g*sh   (h may be NULL)
    double *X, *U, *s, *h1;  // + g*sws        (h1 may be NULL)
    int64_t sS, sh, sws;
    int n;
    int warm;  // U holds the eigenvectors of a previous, nearby S: start the Jacobi sweeps from them
    int fast;  // set by launch_loewdin: FP32 Jacobi + FP64 refinement for n <= 32 (EVC_EIGH_F32=0: FP64 Jacobi)
    double *scratch;   // n > 64: 2 Tp^2 doubles (Tp = n rounded up to 16) + g*sscratch for two of the three work matrices
    int64_t sscratch;  // (NULL: the LDS-only kernels, n <= 80)
    double *flag;      // + g*sws: one word per geometry (32 < n <= 64, part != 0): 1 = the Newton-Schulz launch wrote X, h1
    int part;          // 0: X, h1, U, s.  loewdin_split_available(n) only: 1 = X and h1 alone (Newton-Schulz on the matrix
                       // cores, no eigensolver; U and s are not touched), 2 = U and s alone (X, h1 are not touched) -- the
                       // two halves of a step whose gradient tail alone needs the eigendecomposition (pipeline.hip);
                       // 3 (internal, 32 < n <= 64): X and h1 by the eigensolver for the geometries whose flag is 0
};
int launch_loewdin(const LoewdinArgs &a, int count, hipStream_t st);
bool loewdin_split_available(int n);
int launch_loewdin_big(const LoewdinArgs &a, int count, hipStream_t st);   // subspace_big.hip: 32 < n <= 64
struct SolveArgs {
    const double *h1part;  // (nsp1, T*T) partial sums of the one-body rows      + g*sh1
    int nsp1;
    double alpha1;
    const double *h2part;  // (nsp2, rows2_total) partial sums of two-body rows  + g*sh2
    int nsp2;
    double alpha2;
    const double *S;  // (T,T) + g*sS  (sS = 0: one overlap matrix shared by the batch)
    int64_t sS;
    int T, layout, nroots;
    double e_shift;             // used when e_shift_dev == NULL
    const double *e_shift_dev;  // [count] or NULL
    double *evals, *evecs, *Hout;  // + g*sev, + g*svec, + g*sH  (Hout may be NULL)
    double *w2, *w1;               // + g*sw
    double *w2t;                   // (w2_count, kMaxBatchG) or NULL: w2 of the geometries of a group of kMaxBatchG,
                                   // transposed, in the workspace of the group's first geometry (+ (g - g%32)*sw)
    double *w1t;                   // (T*T, kMaxBatchG) or NULL: the same for w1
    int64_t sh1, sh2, sev, svec, sH, sw;
    int64_t w2_offset, w2_count;  // slice of the global weight vector to write (multi-GPU)
    double *vstd;                 // (m,m), m = T rounded up to even: standard-form eigenvectors, kept in the
                                  // workspace from call to call (+ g*sw); may be NULL
    int warm;                     // start the Jacobi sweeps from vstd
    int fast;                     // set by launch_subspace_solve: FP32 Jacobi + FP64 refinement for T <= 32
    double *bcache;               // (2, T, T) + g*sw or NULL: the lower triangle of the overlap matrix the cached
                                  // B = L^-1 (second block) was computed from -- S_train does not depend on the
                                  // geometry, so every call after the first finds its factorisation here
                                  // (T > kSubspaceSmallT: two blocks of Tp^2, Tp = T rounded up to 16, B at pitch Tp)
    int few;                      // set by launch_subspace_big: nroots <= 4 through the tridiagonal route (EVC_SUBSPACE_FEW)
    double *scratch;              // T > kSubspaceSmallT: subspace_big_scratch_doubles(T) doubles + g*sscratch
    int64_t sscratch;
};
constexpr int kSubspaceSmallT = 32;   // up to here: the register / 32 x 32-tile kernel of dense_small.hip
constexpr int kSubspaceMaxT = 512;    // beyond kSubspaceSmallT: subspace_big.hip (LDS-resident up to 128, then global)
// subspace solve + the eigendecomposition half of the Loewdin step (part 2) in one launch (dense_small.hip)
int launch_subspace_loewdin(const SolveArgs &s, const LoewdinArgs &l, int count, hipStream_t st);
int launch_subspace_solve(const SolveArgs &a, int count, hipStream_t st);
// subspace_big.hip: T > kSubspaceSmallT (vstd, when given, holds Tp^2 doubles: the eigenvectors as ROWS at pitch Tp)
size_t subspace_big_scratch_doubles(int T);
int launch_subspace_big(const SolveArgs &a, int count, hipStream_t st);
// Weights of the t-RDM rows for the predicted RDMs of a GIVEN coefficient vector c[T] (gradients_loewdin.py:343-356):
// w1[a*T+b] = c_a c_b; w2 = the slice [w2_offset, +w2_count) of the two-body row weights (pairs: 2 c_a c_b, c_a^2 on
// the diagonal; otherwise c_a c_b).
int launch_pair_weights(const double *c, int T, int layout, double *w1, double *w2, int64_t w2_offset,
                        int64_t w2_count, hipStream_t st);
struct GradPrepArgs {
    int n;
    const double *X;      // + g*sws
    const double *hcore;  // + g*sh
    const double *D;      // predicted 1-RDM (n,n)  + g*sD
    double *Pao;          // X D X^T                + g*sws
    double *Y1;           // hcore X (D + D^T)      + g*sws
    int64_t sws, sh, sD;
    double scale1;        // 1 on the rank that owns the one-body part, else 0
};
int launch_grad_prep(const GradPrepArgs &a, int count, hipStream_t st);
// grad_prep and the unpack of the packed predicted 2-RDM into the dense (pair, pair) SB in ONE launch (dense_small.hip)
int launch_unpack8_prep(const GradPrepArgs &a, const double *packed, int64_t sp, double *SB, int64_t sws, int count,
                        hipStream_t st);
struct GradFinalArgs {
    int n, natm;
    const double *U, *s;          // eigen-decomposition of S_AO   + g*sws
    const double *Y1;             // (n,n) [a][i]                  + g*sws
    const double *y2;             // (n,n) [i][a]                  + g*sws
    const double *ipovlp;         // (3,n,n)                       + g*sip
    const int64_t *aoslices;      // (A,2) shared
    const double *t2part;         // (n,3,nchunk)                  + g*sws
    int nchunk;
    const double *term3;          // (A*3)                         + g*sws
    const double *gnuc;           // (A,3) or NULL                 + g*sgn
    double scale1;                // 1: include term3 + gnuc
    double *grad;                 // (A,3)                         + g*sgrad
    int64_t sws, sip, sgn, sgrad;
};
int launch_grad_final(const GradFinalArgs &a, int count, hipStream_t st);

}  // namespace evc
