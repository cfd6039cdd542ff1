/*
 * evcont_hip.h -- C ABI of libevcont_hip.so: the MI355X (gfx950) kernels behind the
 * eigenvector-continuation energy/force hot path of BoothGroup/evcont.
 *
 * The reference has no FFI layer (it is pure numpy/scipy/PySCF); its boundary for this
 * path is the Python API of three modules.  Every entry point below replaces one dense
 * operation of those modules and cites it as file:line under /root/reference/evcont.
 * The host-side mirror of the Python API (the evcont_amd Python modules) binds these symbols with
 * ctypes; INTEGRATION.md shows the stub a maintainer of the reference would add.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer to float64 (int64 for aoslices) unless noted;
 *   - `stream` is a hipStream_t passed as void*; all work is enqueued on it, nothing
 *     synchronises, nothing allocates.  State kept by the library: a thread-local error string; per
 *     kernel one bit per device "dynamic LDS limit raised" (hipFuncSetAttribute is per device: one
 *     process may drive several devices through this library); the opt-in evc_profile_* measurement
 *     hook (process-wide, mutex-protected); tuning knobs read from the environment once per process;
 *   - return value 0 = success; <0 = argument error (nothing enqueued);
 *     >0 = hipError_t of a failed launch;
 *   - matrices are row-major (C order), exactly as numpy hands them to the reference.
 */
#ifndef EVCONT_HIP_H
#define EVCONT_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define EVC_ABI_VERSION 9 /* 2: batched phases, evc_subspace_solve_batch, evc_integrals_oao_batch, EVC_FLAG_WARM_START;
                             3: EVC_LAYOUT_SYM8; 4: evc_profile_stage/_select, EVC_FLAG_IP1_S2KL, EVC_FLAG_ERI_S4;
                             5: evc_phase_set_coeffs; 6: evc_phase_loewdin_batch, EVC_FLAG_LOEWDIN_DONE;
                             7: training sets of up to 512 states (evc_subspace_solve[_batch] take a workspace,
                                evc_subspace_solve_ws_bytes), `flags` argument of the phase A / B entry points;
                             8: evc_profile_kernel; the workspace of the compressed layout's pipeline holds its dense
                                (pair, pair) intermediates at the pitch N(N+1)/2 rounded up to 16 doubles;
                             9: evc_release_workspace (a workspace may own a side stream) */

/* t-RDM storage layouts = ndim of the reference's two_RDM argument
 * (ab_initio_eigenvector_continuation.py:41-68). */
#define EVC_LAYOUT_FULL6 6 /* (T,T,N,N,N,N)  rows=T*T        cols=N^4          */
#define EVC_LAYOUT_PAIR5 5 /* (P,N,N,N,N)    rows=T(T+1)/2   cols=N^4          */
#define EVC_LAYOUT_ELEC3 3 /* (T,T,M)        rows=T*T        cols=N^2(N^2+1)/2 */
#define EVC_LAYOUT_PACK2 2 /* (P,M)          rows=T(T+1)/2   cols=M            */
/* Device-side compressed layout (no counterpart in the reference): per training pair (a >= b) the
 * 8-fold symmetrised t-RDM
 *   Gs[i,j,k,l] = mean of Gamma over (i<->j), (k<->l), (ij<->kl)
 * stored for i >= j, k >= l, u = i(i+1)/2+j >= v = k(k+1)/2+l at column u(u+1)/2+v:
 *   rows = T(T+1)/2, cols = Ms(Ms+1)/2 with Ms = N(N+1)/2      (3.7x fewer bytes than PACK2 at N = 30).
 * H_ab and the nuclear gradient only see this part of Gamma WHEN THE AO INTEGRALS HAVE THE INDEX
 * SYMMETRIES OF REAL TWO-ELECTRON INTEGRALS: eri[p,q,r,s] 8-fold symmetric and eri_ip1[x,p,q,r,s] =
 * eri_ip1[x,p,q,s,r] (true for every int2e / int2e_ip1 of real AOs).  Energies, coefficients and gradients are
 * then those of the reference layouts to rounding; g_pred is the symmetrised predicted 2-RDM. */
#define EVC_LAYOUT_SYM8 8

int evc_abi_version(void);
/* Message of the last failing call on this host thread ("" if none). */
const char *evc_last_error(void);

/* ---------------------------------------------------------------------------------
 * K5/K4  H2 = Gamma . h2   -- streaming row GEMV (HBM bound)
 *   y[r] = alpha * sum_{c<cols} A[r*ld + c] * v[c]            r < rows
 * replaces np.tensordot(two_RDM, h2, axes=4) / two_RDM.dot(h2_compressed) and
 * np.tensordot(one_RDM, h1, axes=2)   (ab_initio_eigenvector_continuation.py:38,43,47,57,64)
 * Requirements: A and v 16-byte aligned, ld even, ld >= cols.  `ws` holds the
 * per-span partial sums (deterministic two-stage reduction, no atomics).
 * --------------------------------------------------------------------------------- */
size_t evc_gemv_rows_ws_bytes(int64_t rows, int64_t cols);
int evc_gemv_rows(const double *A, int64_t rows, int64_t cols, int64_t ld, const double *v,
                  double alpha, double *y, void *ws, size_t ws_bytes, void *stream);

/* ---------------------------------------------------------------------------------
 * K8/K7  Gamma_pred = w . Gamma  -- streaming column GEMV (GEMV-T, HBM bound)
 *   out[c] = sum_{r<rows} w[r] * A[r*ld + c]                  c < cols
 * replaces np.tensordot(weights, two_RDM) (ab_initio_gradients_loewdin.py:343,351-356).
 * Requirements: A and out 16-byte aligned, ld even.
 * --------------------------------------------------------------------------------- */
int evc_gemv_cols(const double *A, int64_t rows, int64_t cols, int64_t ld, const double *w,
                  double *out, void *stream);

/* ---------------------------------------------------------------------------------
 * a4/a5  electron-exchange-symmetry codecs (electron_integral_utils.py:38-66, 69-88)
 *   pack  : (n,n,n,n) -> row-major lower triangle of the (n^2,n^2) matrix, diagonal * diag_mult;
 *           elements [M, out_len) of `out` are zero-filled (padding for the streaming GEMV).
 *   unpack: (M,) -> (n,n,n,n), both triangles.
 * Out of place (the reference scales the caller's diagonal in place and restores it).
 * --------------------------------------------------------------------------------- */
int evc_pack_pair_sym(const double *h2, int n, double diag_mult, double *out, int64_t out_len,
                      void *stream);
int evc_unpack_pair_sym(const double *packed, int n, double *out, void *stream);

/* ---------------------------------------------------------------------------------
 * K3/K14  four-index basis rotation, FP64 MFMA (v_mfma_f64_16x16x4_f64)
 *   quarter step: out[q, a,b,c] = sum_d in[a,b,c,d] * C[d,q]     (c_transposed == 0)
 *                                 sum_d in[a,b,c,d] * C[q,d]     (c_transposed != 0)
 *   i.e. the LAST index is rotated and moved to the FRONT; four steps rotate all four
 *   indices and restore their order.
 *   full: out[i,j,k,l] = sum_abcd in[a,b,c,d] C[a,i] C[b,j] C[c,k] C[d,l]
 *   replaces pyscf.ao2mo.kernel + ao2mo.restore(1,..) (electron_integral_utils.py:136,
 *   ab_initio_gradients_loewdin.py:339) and the OAO->AO back-rotation of the 2-RDM (:224-232,
 *   with c_transposed=1).  If `three_quarter` is non-NULL it receives the tensor after three
 *   steps, K[j,k,l,a] = sum_bcd in[a,b,c,d] C[b,j] C[c,k] C[d,l]  (used by K13, :210-222).
 *   `tmp` is scratch of n^4 doubles; in/out/tmp/three_quarter must not alias.  n <= 96.
 * --------------------------------------------------------------------------------- */
int evc_quarter_transform(const double *in, const double *C, int c_transposed, int n,
                          double *out, void *stream);
int evc_four_index_transform(const double *in, const double *C, int c_transposed, int n,
                             double *out, double *tmp, double *three_quarter, void *stream);

/* ---------------------------------------------------------------------------------
 * K1/K2  Loewdin orthogonalisation on one workgroup (parallel cyclic Jacobi in LDS)
 *   S = U diag(s) U^T ; X = U diag(s>1e-15 ? s^-1/2 : 0) U^T ; h1 = X^T hcore X
 *   replaces get_loewdin_trafo (electron_integral_utils.py:6-18) and the h1 rotation
 *   (:135, ab_initio_gradients_loewdin.py:338).  hcore/h1 may be NULL.  n <= 80 (96 inside the fused pipeline, which lends
 *   the kernel scratch from its workspace).
 * --------------------------------------------------------------------------------- */
int evc_loewdin(const double *S, const double *hcore, int n, double *X, double *U, double *s,
                double *h1, void *stream);

/* ---------------------------------------------------------------------------------
 * K6  subspace generalised eigenproblem H c = E S c on one workgroup
 *   H is assembled from the one-body part h1rows (T*T) and the two-body rows
 *   (rows2 = T*T or T(T+1)/2 values, already scaled) exactly as
 *   ab_initio_eigenvector_continuation.py:38-68 does, then solved like scipy.linalg.eigh(H,S)
 *   (LAPACK dsygvd: lower triangles, Cholesky of S, c^T S c = 1) (:73-88, :157-173).
 *   Outputs: evals[nroots] ascending (+ e_shift), evecs[nroots*T] (row k = k-th vector),
 *   w2[rows2] / w1[T*T]: weights of the two-/one-body t-RDM rows for the predicted RDMs of
 *   root 0 (ab_initio_gradients_loewdin.py:343-356), Hout[T*T] (may be NULL): the matrix handed
 *   to the eigensolver.  Hermitian branch only.
 *   T <= 512 (the reference has no bound: its Zundel learning curve uses 80 and 100 training states,
 *   scripts/MD/Zundel_thermodynamics/continuation/05_Zundel_test_potential_energy.py:182-210).  T <= 32 runs in the
 *   registers / LDS of one 256-thread workgroup and needs no workspace (ws may be NULL); larger T run on a 1024-thread
 *   workgroup with one T x T matrix in LDS (T <= 128; in global memory beyond) and need `ws` of
 *   evc_subspace_solve_ws_bytes(T, count) bytes, 16-byte aligned.
 * --------------------------------------------------------------------------------- */
size_t evc_subspace_solve_ws_bytes(int T, int count);
int evc_subspace_solve(const double *h1rows, const double *h2rows, const double *S_train, int T,
                       int layout, int nroots, double e_shift, double *evals, double *evecs,
                       double *w2, double *w1, double *Hout, void *ws, size_t ws_bytes, void *stream);
/* `count` independent problems H[g] c = E S[g] c of one size (one workgroup each): H (count,T,T) and
 * S (T,T) shared (s_stride = 0) or (count,T,T) (s_stride = T*T), lower triangles read.  Writes
 * evals (count,T) (first nroots of each row, + e_shift[g] if e_shift != NULL) and evecs (count,T,T).
 * This is what the active-learning loop needs: the continuation energies of a trajectory for SUBSETS
 * of the training set (MD_utils.py:264-299,448-483) are eigenproblems of sub-matrices of the one
 * H(R) of the full set, so the t-RDM is contracted once per geometry, not once per subset. */
int evc_subspace_solve_batch(const double *H, const double *S, int64_t s_stride, int T, int count,
                             int nroots, const double *e_shift, double *evals, double *evecs,
                             void *ws, size_t ws_bytes, void *stream);

/* OAO integrals of `count` geometries (get_basis + get_integrals, electron_integral_utils.py:91-138):
 * X = S^-1/2, h1 = X^T hcore X (count,N,N), h2 = four-index rotation of eri (count,N,N,N,N); `trafo`
 * (count,N,N) receives X if non-NULL.  Inputs stacked along the leading axis.  Used by the
 * "farthest_point_ham" selection metric (MD_utils.py:363-405).  n <= 96. */
size_t evc_integrals_oao_ws_bytes(int n, int count);
int evc_integrals_oao_batch(int n, int count, const double *S, const double *hcore, const double *eri,
                            double *h1, double *h2, double *trafo, void *ws, size_t ws_bytes,
                            void *stream);

/* ---------------------------------------------------------------------------------
 * Fused per-geometry pipeline (ab_initio_gradients_loewdin.py:308-379 get_energy_with_grad,
 * ab_initio_eigenvector_continuation.py:178-250 *_OAO)
 * --------------------------------------------------------------------------------- */
typedef struct evc_trdm_set {
    int32_t n;           /* orbitals N (<= 96; the reference's largest orbital space is 58) */
    int32_t ntrain;      /* training states T (<= 512) */
    int32_t layout;      /* EVC_LAYOUT_* */
    int32_t reserved;
    int64_t rows2;       /* rows of the two-body matrix view held by THIS rank */
    int64_t row_offset;  /* index of the first local row in the global row numbering */
    int64_t rows2_total; /* global number of rows (T*T or T(T+1)/2) */
    int64_t cols2;       /* N^4 or M */
    int64_t ld2;         /* leading dimension of two_rdm (even, >= cols2) */
    int64_t ld1;         /* leading dimension of one_rdm (even, >= N*N) */
    const double *two_rdm; /* (rows2, ld2) */
    const double *one_rdm; /* (T*T, ld1), replicated on every rank */
    const double *s_train; /* (T,T) */
} evc_trdm_set;

typedef struct evc_geometry {
    int32_t natm;
    int32_t reserved;
    double enuc;             /* mol.energy_nuc() */
    const double *S;         /* (N,N)       int1e_ovlp */
    const double *hcore;     /* (N,N)       scf.hf.get_hcore */
    const double *eri;       /* (N,N,N,N)   int2e; 16-byte aligned (as eri_ip1) */
    const double *ipovlp;    /* (3,N,N)     int1e_ipovlp             (NULL: energy only) */
    const double *dhcore;    /* (A,3,N,N)   hcore_generator()(atom)  (NULL: energy only) */
    const double *eri_ip1;   /* (3,N,N,N,N) int2e_ip1                (NULL: energy only) */
    const double *gnuc;      /* (A,3)       grad_nuc()               (NULL: energy only) */
    const int64_t *aoslices; /* (A,2) [start,stop)                   (NULL: energy only) */
} evc_geometry;

typedef struct evc_outputs {
    double *energy;  /* [nroots]  total energies (electronic + enuc) */
    double *coeffs;  /* [nroots*T] */
    double *grad;    /* [A*3] or NULL */
    double *d_pred;  /* [N*N]  predicted 1-RDM (always written by the gradient phase) */
    double *g_pred;  /* [N^4]  predicted 2-RDM, unpacked; may be NULL (for the packed layouts the gradient
                        phase then never materialises it) */
    double *hmat;    /* [T*T] or NULL: matrix handed to the eigensolver */
} evc_outputs;

/* flags */
#define EVC_FLAG_ENERGY_ONLY 1  /* stop after the eigensolve */
#define EVC_FLAG_PARTIAL_RANK 2 /* multi-GPU: this rank is not rank 0 -> its partial gradient carries
                                   only the two-body contribution of its rows */
#define EVC_FLAG_WARM_START 4   /* the workspace holds the results of a previous evaluation at a NEARBY geometry
                                   (an MD step): the two Jacobi eigensolvers (overlap matrix, subspace problem) start
                                   from its eigenvectors and converge in 2-3 sweeps instead of 7-8.  Results agree
                                   with a cold start to solver tolerance (~1e-14), not bit for bit; stale or
                                   never-written eigenvectors are detected and ignored.  Fused entry points only. */

#define EVC_FLAG_IP1_S2KL 8     /* geometry.eri_ip1 is (3,N,N,N(N+1)/2) [per geometry of a batch]: int2e_ip1 packed in its
                                   last two AO indices, element (x,p,q,k(k+1)/2+l), k >= l -- what PySCF returns for
                                   mol.intor("int2e_ip1", aosym="s2kl").  Half the bytes of the largest input; the
                                   contraction then streams dense rows.  Only with EVC_LAYOUT_SYM8 and N <= 32 (the
                                   path that uses the r <-> s symmetry of int2e_ip1 anyway). */
#define EVC_FLAG_ERI_S4 16      /* geometry.eri is the dense (Ms,Ms) matrix, Ms = N(N+1)/2 [per geometry of a batch]: int2e
                                   packed in both index pairs, element (p(p+1)/2+q, r(r+1)/2+s), p >= q, r >= s -- what PySCF
                                   returns for mol.intor("int2e", aosym="s4").  A quarter of the bytes.  Fused entry points,
                                   EVC_LAYOUT_SYM8 and N <= 32 only. */

#define EVC_FLAG_LOEWDIN_DONE 32 /* the workspace already holds X, U, s and h1 of THESE geometries (evc_phase_loewdin_batch
                                   ran on it): phase A skips the Loewdin kernel.  Lets a host overlap that latency-bound
                                   kernel -- it reads only S and hcore -- with the upload of the large integral arrays
                                   (hosted MD) or with the previous batch's streaming kernels (throughput).  Batch entry
                                   points only. */

/* The workspace also caches the inverse Cholesky factor of t->s_train next to the matrix it was computed from; a call
 * whose s_train is bit-identical to that copy reuses the factor.  ZERO-FILL a workspace once after allocating it (the
 * Python layer does): a recycled allocation must not look like a hit. */
size_t evc_workspace_bytes(const evc_trdm_set *t, int natm);

/* Phase A: Loewdin + integrals + H rows.  Writes h2rows_local[rows2] (scaled two-body rows of this
 * rank) and h1rows[T*T] into the workspace; pointers are returned through the out arguments so a
 * multi-GPU host can all-gather the rows between the phases.
 * flags: EVC_FLAG_ERI_S4 (g->eri packed), EVC_FLAG_WARM_START. */
int evc_phase_hamiltonian(const evc_trdm_set *t, const evc_geometry *g, int flags, void *ws, size_t ws_bytes,
                          double **h2rows_local, double **h1rows, void *stream);
/* Phase B: eigensolve from the complete row vector h2rows_all[rows2_total].  flags: EVC_FLAG_WARM_START. */
int evc_phase_solve(const evc_trdm_set *t, const evc_geometry *g, const double *h2rows_all,
                    const evc_outputs *out, int nroots, int flags, void *ws, size_t ws_bytes, void *stream);
/* Phase C: predicted RDMs of root 0 + Loewdin-response nuclear gradient. */
int evc_phase_gradient(const evc_trdm_set *t, const evc_geometry *g, const evc_outputs *out,
                       int flags, void *ws, size_t ws_bytes, void *stream);
/* Between B and C, optional: replace the row weights phase B left in the workspace by those of a coefficient vector
 * the CALLER supplies (coeffs[T], device): w1 = c c^T, two-body rows 2 c_a c_b / c_a^2 (pair layouts) or c_a c_b
 * (ab_initio_gradients_loewdin.py:343-356).  This is how hermitian=False is served: the T x T pencil (outputs.hmat of
 * phase B) is solved by scipy.linalg.eig on the host, exactly as the reference does (:76-88), and phase C then builds
 * the predicted RDMs and the gradient from ITS eigenvector.  `natm` as passed to evc_workspace_bytes. */
int evc_phase_set_coeffs(const evc_trdm_set *t, const double *coeffs, int natm, void *ws, size_t ws_bytes,
                         void *stream);
/* A+B(+C) back to back on one device. */
int evc_energy_with_grad(const evc_trdm_set *t, const evc_geometry *g, const evc_outputs *out,
                         int nroots, int flags, void *ws, size_t ws_bytes, void *stream);

/* ---------------------------------------------------------------------------------
 * Batched form: `count` independent geometries of the SAME molecule (same N, A, aoslices) per call.
 * Every launch of the pipeline covers the whole batch, and the two streaming kernels read the
 * t-RDM ONCE for up to 32 geometries (K5 becomes (rows,cols)x(cols,G), K8 (G,rows)x(rows,cols)),
 * so the HBM cost of the t-RDM per evaluation drops by the batch size.  This is the throughput
 * form for PES scans / batched re-evaluations; an MD run is count = 1.
 * All arrays are stacked along a leading batch axis, C order.
 * --------------------------------------------------------------------------------- */
typedef struct evc_geometry_batch {
    int32_t natm;
    int32_t count;
    const double *enuc;      /* (count)           mol.energy_nuc() per geometry, DEVICE array */
    const double *S;         /* (count,N,N) */
    const double *hcore;     /* (count,N,N) */
    const double *eri;       /* (count,N,N,N,N) */
    const double *ipovlp;    /* (count,3,N,N)       (NULL: energy only) */
    const double *dhcore;    /* (count,A,3,N,N)     (NULL: energy only) */
    const double *eri_ip1;   /* (count,3,N,N,N,N)   (NULL: energy only) */
    const double *gnuc;      /* (count,A,3)         (NULL: energy only) */
    const int64_t *aoslices; /* (A,2) shared by the batch */
} evc_geometry_batch;

typedef struct evc_outputs_batch {
    double *energy;  /* (count,T)    first nroots entries of each row are written */
    double *coeffs;  /* (count,T,T)  first nroots rows of each block are written */
    double *grad;    /* (count,A,3) or NULL with EVC_FLAG_ENERGY_ONLY */
    double *d_pred;  /* (count,N,N) or NULL (kept in the workspace) */
    double *g_pred;  /* (count,N,N,N,N) or NULL (kept in the workspace) */
    double *hmat;    /* (count,T,T) or NULL */
} evc_outputs_batch;

size_t evc_workspace_bytes_batch(const evc_trdm_set *t, int natm, int count);
/* The full calls (evc_energy_with_grad, evc_energy_with_grad_batch) compute the Loewdin transformation X = S^-1/2
 * (electron_integral_utils.py:6-18) by a Newton-Schulz iteration and keep the eigendecomposition of S, which only the
 * response term of the gradient needs (ab_initio_gradients_loewdin.py:41-134,300-303), off the critical path: for
 * N <= 32 and T <= 32 it rides in the launch of the subspace solve; for 33 ... 64 orbitals (or larger training sets)
 * calls of fewer than 12 geometries run it on a side stream the library creates per device, forked from and joined into
 * `stream` inside the call (two events per workspace, created at its first such call) -- the stream semantics of the
 * call are unchanged.  Before freeing a workspace that may have been used that way, hand its pointer to
 * evc_release_workspace: it waits for the last such launch into the workspace and destroys the events (a pointer that
 * owns none: no-op, returns 0).  EVC_LOEWDIN_SPLIT=0: one Loewdin kernel always. */
int evc_release_workspace(void *ws);
int evc_energy_with_grad_batch(const evc_trdm_set *t, const evc_geometry_batch *gb,
                               const evc_outputs_batch *ob, int nroots, int flags, void *ws,
                               size_t ws_bytes, void *stream);

/* Phase L: Loewdin orthogonalisation of the batch alone (electron_integral_utils.py:6-18,135): reads gb->S and
 * gb->hcore only, leaves X, U, s, h1 in the workspace for a following call with EVC_FLAG_LOEWDIN_DONE.
 * flags: EVC_FLAG_WARM_START. */
int evc_phase_loewdin_batch(const evc_trdm_set *t, const evc_geometry_batch *gb, int flags, void *ws, size_t ws_bytes,
                            void *stream);

/* The three phases of the pair-sharded evaluation for a batch (t may hold a row slice).
 * A: writes the scaled two-body rows of this rank to rows_out[g*ld_rows_out + r], r < t->rows2
 *    (caller's buffer: the send buffer of the all-gather).  flags: EVC_FLAG_ERI_S4 (gb->eri packed: the phases then
 *    run the same symmetric pipeline as the fused entry point), EVC_FLAG_LOEWDIN_DONE, EVC_FLAG_WARM_START.
 * B: h2rows_all[g*ld_rows_all + p], p < t->rows2_total, is the gathered row vector of geometry g.
 *    flags: EVC_FLAG_WARM_START.
 * C: as evc_phase_gradient; with EVC_FLAG_PARTIAL_RANK ob->grad receives only the two-body share;
 *    EVC_FLAG_IP1_S2KL as for the fused entry point. */
int evc_phase_hamiltonian_batch(const evc_trdm_set *t, const evc_geometry_batch *gb, int flags, double *rows_out,
                                int64_t ld_rows_out, void *ws, size_t ws_bytes, void *stream);
int evc_phase_solve_batch(const evc_trdm_set *t, const evc_geometry_batch *gb, const double *h2rows_all,
                          int64_t ld_rows_all, const evc_outputs_batch *ob, int nroots, int flags, void *ws,
                          size_t ws_bytes, void *stream);
int evc_phase_gradient_batch(const evc_trdm_set *t, const evc_geometry_batch *gb,
                             const evc_outputs_batch *ob, int flags, void *ws, size_t ws_bytes,
                             void *stream);

/* Gradient of given (not predicted) RDMs: get_grad_elec_OAO (ab_initio_gradients_loewdin.py:255-305).
 * `trafo` = caller's ao_mo_trafo (N,N) or NULL for the Loewdin trafo of g->S; the trafo derivative is
 * the Loewdin response of g->S in both cases (:274-277).  Writes the ELECTRONIC gradient (no grad_nuc)
 * to grad[A*3]. */
int evc_grad_elec_oao(int n, const evc_geometry *g, const double *trafo, const double *one_rdm,
                      const double *two_rdm, double *grad, void *ws, size_t ws_bytes, void *stream);
size_t evc_grad_elec_ws_bytes(int n, int natm);

/* ---------------------------------------------------------------------------------
 * Tensor-valued building blocks of ab_initio_gradients_loewdin.py (off the fused path, which uses
 * the adjoint form and never materialises them).  Layouts are the reference's: (N,N,N,N) and
 * (N,N,A,3), C order.  n <= 64 (evc_one_el_grad: n <= 60).
 *   evc_loewdin_trafo_grad      loewdin_trafo_grad(S)            (:41-112)  LG[p,q,a,b] = dX_pq/dS_ab(sym)
 *                               in Daleckii-Krein closed form; ws = 2n^2+n doubles
 *   evc_derivative_ao_mo_trafo  get_derivative_ao_mo_trafo(mol)  (:115-134) dX[k,l,A,x]; ws as above
 *   evc_one_el_grad             get_one_el_grad(mol, X, dX)      (:155-187) h1_jac[j,n,A,x]
 *   evc_two_el_grad             two_el_grad(h2_ao, G, X, dX, ip1, slices) (:190-252) -> (A,3)
 *   evc_contract_nnA3           out[A,x] = sum_ij T[i,j,A,x] * (transposed ? M[j,i] : M[i,j])  (:300)
 * --------------------------------------------------------------------------------- */
int evc_loewdin_trafo_grad(const double *S, int n, double *LG, double *ws, void *stream);
int evc_derivative_ao_mo_trafo(const double *S, const double *ipovlp, const int64_t *aoslices, int n,
                               int natm, double *dX, double *ws, void *stream);
int evc_one_el_grad(const double *X, const double *hcore, const double *dhcore, const double *dX, int n,
                    int natm, double *out, void *stream);
size_t evc_two_el_grad_ws_bytes(int n);
int evc_two_el_grad(const double *h2_ao, const double *two_rdm, const double *X, const double *dX,
                    const double *ip1, const int64_t *aoslices, int n, int natm, double *out, void *ws,
                    size_t ws_bytes, void *stream);
int evc_contract_nnA3(const double *T, const double *M, int transposed, int n, int natm, double *out,
                      void *stream);

/* ---------------------------------------------------------------------------------
 * Measurement hook (bench.py): while enabled, the fused pipeline records hipEvents on the launch
 * stream immediately before and after the launches of the stages below, for up to max_samples
 * evaluations.  evc_profile_end synchronises those events, returns the summed durations in
 * milliseconds with the number of launches for K5 (rows GEMV) and K8 (cols GEMV), and frees them;
 * afterwards evc_profile_stage reports the same two numbers for any stage selected for that session.
 * Process-wide state.
 * --------------------------------------------------------------------------------- */
#define EVC_PROF_ROWS 0           /* K5  H2 = Gamma . h2 (+ one-body) */
#define EVC_PROF_COLS 1           /* K8  predicted RDMs */
#define EVC_PROF_PAIR_TRANSFORM 2 /* fused pair steps of the two four-index rotations (4 launches per call) */
#define EVC_PROF_IP1 3            /* int2e_ip1 contraction (+ dhcore dots, Y2 slab sums) */
#define EVC_PROF_Y2 4             /* Y2 = K3 . Gamma_sym */
#define EVC_PROF_UNPACK 5         /* unpack/symmetrise the predicted 2-RDM */
#define EVC_PROF_LOEWDIN 6        /* Loewdin orthogonalisation */
#define EVC_PROF_SUBSPACE 7       /* subspace generalised eigenproblem */
int evc_profile_begin(int max_samples);
int evc_profile_end(double *rows_ms, int *rows_n, double *cols_ms, int *cols_n);
int evc_profile_stage(int stage, double *ms, int *launches);
/* Stages timed by the next sessions: bit s = stage s (default: EVC_PROF_ROWS and EVC_PROF_COLS only -- every timed
 * launch costs two event records, which is visible in the one-geometry-at-a-time regime). */
int evc_profile_select(unsigned stage_mask);
/* Name (with template arguments) of the kernel the library most recently launched for a stage, e.g.
 * "gemv_rows_lds_kernel<2,7,2> G=32" (K5 / EVC_PROF_ROWS: followed by the geometries that launch contracted): what a
 * measurement of that stage timed.  "" before the first launch.  Process-wide. */
const char *evc_profile_kernel(int stage);

#ifdef __cplusplus
}
#endif
#endif /* EVCONT_HIP_H */
