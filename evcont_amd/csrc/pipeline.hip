// Per-geometry orchestration: enqueues the whole energy(+force) DAG on one HIP stream.
// Mirrors get_energy_with_grad (gradients_loewdin.py:308-379) and approximate_*_OAO
// (evcont.py:178-250).  Two generalisations over the reference:
//   * BATCH: `count` independent geometries go through every launch together (batch index =
//     blockIdx.y / blockIdx.x) and share ONE pass over the t-RDM in the two streaming kernels;
//   * PHASES: split in three so a pair-sharded multi-GPU host can put its two small collectives
//     (all-gather of the H rows, all-reduce of the gradient) between them.
#include <stdlib.h>
#include <string.h>

#include <atomic>
#include <mutex>
#include <unordered_map>

#include "common.hpp"
#include "kernels.hpp"

namespace evc {

static thread_local char g_err[512] = "";
void set_error(const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

// Optional in-stream timing of the two streaming kernels (bench.py's roofline leg): hipEvents are
// recorded on the launch stream right before/after the kernel, so the figure is the kernel's own
// duration inside the real per-geometry DAG.  Process-wide, off by default.
// Stages: EVC_PROF_* of include/evcont_hip.h.
constexpr int kProfStages = 8;
static char g_kernel_ran[kProfStages][96];
void note_kernel(int stage, const char *fmt, ...) {
    if (stage < 0 || stage >= kProfStages) return;
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_kernel_ran[stage], sizeof(g_kernel_ran[stage]), fmt, ap);
    va_end(ap);
}
constexpr int kProfPerSample = 16;   // event pairs one evaluation can record
struct Prof {
    std::atomic<bool> on{false};
    int cap = 0, n = 0;          // records: capacity, used
    hipEvent_t *ev = nullptr;    // [cap][2]: start, stop
    int *stage = nullptr;        // [cap]
    double ms[kProfStages] = {0};   // results of the last evc_profile_end
    int cnt[kProfStages] = {0};
    unsigned mask = (1u << EVC_PROF_ROWS) | (1u << EVC_PROF_COLS);   // stages that are timed (evc_profile_select)
};
static Prof g_prof;
static std::mutex g_prof_mu;   // record allocation and begin/end/select: host threads may share the hook
// start of a timed launch: returns the record index or -1
static int prof_start(int stage, hipStream_t st) {
    if (!g_prof.on) return -1;   // the common case: no lock taken
    std::lock_guard<std::mutex> lk(g_prof_mu);
    if (!g_prof.on || !(g_prof.mask >> stage & 1u) || g_prof.n >= g_prof.cap) return -1;
    const int i = g_prof.n++;
    g_prof.stage[i] = stage;
    (void)hipEventRecord(g_prof.ev[2 * i], st);
    return i;
}
static void prof_stop(int i, hipStream_t st) {
    if (i >= 0) (void)hipEventRecord(g_prof.ev[2 * i + 1], st);
}

// Internal batch view of the geometry inputs / outputs (strides in doubles; 0 for a single geometry).
struct Geo {
    int natm, count;
    const double *S, *hcore, *eri, *ipovlp, *dhcore, *eri_ip1, *gnuc;
    const int64_t *aoslices;
    int64_t sS, sh, seri, sip, sdh, sip1, sgn;
    double enuc;             // used when enuc_dev == NULL
    const double *enuc_dev;  // [count]
    int eri_s4;              // eri is the dense (pair, pair) matrix (EVC_FLAG_ERI_S4); set from the call's flags
};
struct Out {
    double *energy, *coeffs, *grad, *d_pred, *g_pred, *hmat;
    int64_t se, sc, sg, sd, sG, sH;
};

struct Ws {
    // N^2-sized
    double *X, *U, *s, *h1, *Dpred, *Pao, *Y1;
    double *lflag;   // one word: the Newton-Schulz launch of a split Loewdin step delivered (32 < n <= 64)
    // N^4-sized
    double *B1, *B2, *K3, *G;
    double *vec2;  // ld2-long vector: packed h2 (phase A) / packed predicted 2-RDM (phase C)
    // t-RDM contraction
    double *h2part, *h1part, *h2rows, *w2, *w1, *w2t, *w1t;
    double *d1part;   // row-slab partials of the predicted 1-RDM (large training sets: gemv_cols_slab_kernel)
    // gradient partials
    double *y2part, *y2, *t2part, *term3;
    // scratch outputs when the caller passes NULL
    double *evals, *evecs;
    // eigenvectors kept from call to call for EVC_FLAG_WARM_START (U above serves the Loewdin step)
    double *vstd;
    double *bcache;   // (2, T, T): overlap matrix (lower triangle) and the inverse Cholesky factor computed from it
    double *sbig;     // T > kSubspaceSmallT: scratch of the large-T subspace kernel (subspace_big.hip)
    bool warm;
    bool loewdin_done;   // X, U, s, h1 are already in the workspace (EVC_FLAG_LOEWDIN_DONE)
    void *base;          // the caller's workspace pointer (key of its side stream, side_of)
    int split;           // Loewdin step of this call: 0 = one kernel; 1 = X, h1 by Newton-Schulz on the call's stream and
                         // U, s by the eigensolver on the device's side stream, joined in front of launch_grad_final;
                         // 3 = the same with the eigensolver riding in the launch of the subspace solve (la_ride)
    LoewdinArgs la_ride; // split == 3: the eigensolver launch phase_solve still owes
    size_t bytes;    // of ONE geometry
    int64_t stride;  // the same in doubles
    RowProblem rp2, rp1;
};

// n <= 32: the four-index rotations run as two fused pair steps (transform.hip / pair_dma.hip); larger n as four
// quarter steps.
static bool use_pair_transform(int n) { return n <= kPairTransformMaxN; }
// 32 < n <= 64 on the compressed layout with BOTH large arrays handed over packed (EVC_FLAG_ERI_S4 with the energy
// phase, EVC_FLAG_IP1_S2KL with the gradient phase): the symmetric pipeline on 64 x 64 operand matrices (pair64.hip).
// The two phases of one evaluation must agree (the gradient phase finds the first pair step's intermediate, not the
// three-quarter-transformed integrals, in the K3 buffer): the fused entry points check it, callers of the phase entry
// points pass both flags or neither.  (Full arrays take the quarter-step route.)
static bool use_pair64(int layout, int n, bool packed_input) {
    return layout == EVC_LAYOUT_SYM8 && n > kPairTransformMaxN && n <= 64 && packed_input;
}

// Geometries per pass of the multi-kernel stages of a batched call (integral rotation, gradient tail): one pass over
// the whole batch (chunks of 16, which keep the intermediates closer to the Infinity Cache, measured within noise).
static int stage_chunk(int count) { return count; }

// Many spans: sum the partials in a multi-workgroup launch instead of inside the eigensolver kernel.
static bool reduce_in_own_launch(const Ws &w) { return w.rp2.nspans > 64; }

static bool is_sym8(int layout) { return layout == EVC_LAYOUT_SYM8; }
// Y2 with the half-transformed integrals recomputed (y2.hip y2_fused_kernel): the energy phase then keeps the
// dense (pair, pair) intermediate of its first pair step in the K3 buffer instead of writing K3 (EVC_Y2_FUSED=0: K3)
static bool use_fused_y2(bool sym8, int n) {
    return sym8 && y2_fused_available(n);   // (callers have decided for the pair-step route: n <= 32 or use_pair64)
}
static bool is_packed(int layout) { return layout == EVC_LAYOUT_ELEC3 || layout == EVC_LAYOUT_PACK2 || is_sym8(layout); }
static bool is_pairs(int layout) { return layout == EVC_LAYOUT_PAIR5 || layout == EVC_LAYOUT_PACK2 || is_sym8(layout); }

static int check_set(const evc_trdm_set *t) {
    EVC_REQUIRE(t != nullptr, "trdm_set is NULL");
    EVC_REQUIRE(t->n >= 1 && t->n <= kMaxOrbitals, "trdm_set: n=%d out of range 1..%d", t->n, kMaxOrbitals);
    EVC_REQUIRE(t->ntrain >= 1 && t->ntrain <= kSubspaceMaxT, "trdm_set: ntrain=%d out of range 1..%d", t->ntrain,
                kSubspaceMaxT);
    EVC_REQUIRE(t->layout == 6 || t->layout == 5 || t->layout == 3 || t->layout == 2 || t->layout == EVC_LAYOUT_SYM8,
                "trdm_set: layout=%d (must be the ndim of two_RDM: 6, 5, 3 or 2, or EVC_LAYOUT_SYM8)", t->layout);
    const int64_t n2 = (int64_t)t->n * t->n, ns = (int64_t)t->n * (t->n + 1) / 2;
    const int64_t cols = is_sym8(t->layout) ? ns * (ns + 1) / 2 : is_packed(t->layout) ? n2 * (n2 + 1) / 2 : n2 * n2;
    const int64_t rows = is_pairs(t->layout) ? (int64_t)t->ntrain * (t->ntrain + 1) / 2
                                              : (int64_t)t->ntrain * t->ntrain;
    EVC_REQUIRE(t->cols2 == cols, "trdm_set: cols2=%lld, expected %lld", (long long)t->cols2, (long long)cols);
    EVC_REQUIRE(t->rows2_total == rows, "trdm_set: rows2_total=%lld, expected %lld", (long long)t->rows2_total,
                (long long)rows);
    EVC_REQUIRE(t->rows2 >= 0 && t->row_offset >= 0 && t->row_offset + t->rows2 <= rows,
                "trdm_set: local rows [%lld,+%lld) outside 0..%lld", (long long)t->row_offset,
                (long long)t->rows2, (long long)rows);
    EVC_REQUIRE(t->ld2 >= cols && t->ld2 % 2 == 0, "trdm_set: ld2=%lld must be even and >= cols2",
                (long long)t->ld2);
    EVC_REQUIRE(t->ld1 >= n2 && t->ld1 % 2 == 0, "trdm_set: ld1=%lld must be even and >= N*N", (long long)t->ld1);
    EVC_REQUIRE(t->rows2 == 0 || (t->two_rdm && aligned16(t->two_rdm)), "trdm_set: two_rdm NULL or misaligned");
    EVC_REQUIRE(t->one_rdm && aligned16(t->one_rdm) && t->s_train, "trdm_set: one_rdm/s_train NULL or misaligned");
    return 0;
}

static void carve(const evc_trdm_set *t, int natm, char *base, Ws &w) {
    const size_t n = t->n, n2 = n * n, n4 = n2 * n2, T = t->ntrain;
    size_t off = 0;
    auto take = [&](size_t doubles) {
        double *p = base ? reinterpret_cast<double *>(base + off) : nullptr;
        off += align_up(doubles * sizeof(double), 256);
        return p;
    };
    w.X = take(n2);
    w.U = take(n2);
    w.s = take(n);
    w.lflag = take(1);
    w.h1 = take(n2);
    w.Dpred = take(n2);
    w.Pao = take(n2);
    w.Y1 = take(n2);
    // (the symmetric pipeline keeps dense (pair, pair) matrices in these: pair_ld(n) rows -- whole 16-row groups are
    //  written -- at the pitch pair_ld(n), which exceeds n^4 doubles for n <= 3)
    const size_t ldp = (size_t)pair_ld((int)n), nbig = n4 > ldp * ldp ? n4 : ldp * ldp;
    w.B1 = take(nbig);
    w.B2 = take(nbig);
    w.K3 = take(nbig);
    w.G = take(n4);
    w.vec2 = take((size_t)t->ld2 + 2);
    memset(&w.rp2, 0, sizeof(w.rp2));
    memset(&w.rp1, 0, sizeof(w.rp1));
    w.rp2.rows = t->rows2;
    w.rp2.cols = t->cols2;
    w.rp2.ld = t->ld2;
    // buffers are sized for the finer (batched) span plan; replan() picks the plan of the actual call
    if (t->rows2 > 0) plan_rows(w.rp2, true);
    w.rp1.rows = (int64_t)T * T;
    w.rp1.cols = (int64_t)n2;
    w.rp1.ld = t->ld1;
    plan_rows(w.rp1, true);
    // (carved for whichever span plan makes more spans: plan_rows or the LDS-staged kernel's, gemv_lds.hip)
    w.h2part = take((size_t)t->rows2 * (t->rows2 > 0 ? rows_max_spans(w.rp2, false) : 1) + 1);
    w.h1part = take((size_t)T * T * rows_max_spans(w.rp1, true));
    w.h2rows = take((size_t)t->rows2_total);
    w.w2 = take((size_t)t->rows2 + 1);
    w.w2t = take((size_t)t->rows2 * kMaxBatchG + 1);
    w.w1 = take(T * T);
    w.w1t = take((size_t)T * T * kMaxBatchG);
    w.d1part = take(T * T >= 1024 ? (size_t)kColSlabs * t->ld1 : 0);
    w.y2part = take((size_t)y2_slab_capacity((int)n) * n2);
    w.y2 = take(n2);
    w.t2part = take((size_t)n * 3 * ip1_chunks((int)n));
    w.term3 = take((size_t)(natm > 0 ? natm : 1) * 3);
    w.evals = take(T);
    w.evecs = take(T * T);
    const size_t Tp = (T + 15) & ~(size_t)15;   // T > kSubspaceSmallT: matrices at pitch Tp (subspace_big.hip)
    const bool bigT = T > (size_t)kSubspaceSmallT;
    w.vstd = take(bigT ? Tp * Tp : ((T + 1) & ~(size_t)1) * ((T + 1) & ~(size_t)1));
    w.bcache = take(bigT ? 2 * Tp * Tp : 2 * T * T);
    w.sbig = take(bigT ? subspace_big_scratch_doubles((int)T) : 0);
    w.warm = false;
    w.loewdin_done = false;
    w.base = base;
    w.split = 0;
    w.bytes = off;
    w.stride = (int64_t)(off / sizeof(double));
}

// y[g][r] = alpha * sum_k partial[g][k][r]: the fixed-order sum of the span partials, done here (many
// workgroups) rather than inside the single-workgroup eigensolver when there are many spans.
// Block = 64 rows x 4 span groups; blockIdx.y = geometry.
__global__ __launch_bounds__(256) void rows_reduce_kernel(const double *partial, int64_t spart, int64_t rows,
                                                          int nspans, double alpha, double *y, int64_t sy) {
    __shared__ double part[16][17];
    partial += (int64_t)blockIdx.y * spart;
    y += (int64_t)blockIdx.y * sy;
    const int rl = threadIdx.x & 15, grp = threadIdx.x >> 4;   // 16 rows x 16 span lanes
    const int64_t r = (int64_t)blockIdx.x * 16 + rl;
    double s0 = 0.0, s1 = 0.0;
    if (r < rows) {
        int k = grp;
        for (; k + 16 < nspans; k += 32) {
            s0 += partial[(int64_t)k * rows + r];
            s1 += partial[(int64_t)(k + 16) * rows + r];
        }
        if (k < nspans) s0 += partial[(int64_t)k * rows + r];
    }
    part[grp][rl] = s0 + s1;
    __syncthreads();
    if (grp == 0 && r < rows) {
        double s = 0.0;
#pragma unroll
        for (int k = 0; k < 16; ++k) s += part[k][rl];   // fixed order
        y[r] = alpha * s;
    }
}

// ---- the Loewdin step in two halves (n <= 64) -------------------------------------------------------------
// The energy phase needs X = S^-1/2 and h1 only; the eigenvectors and eigenvalues of S enter at the very end of the
// gradient (the response term, launch_grad_final).  A full call therefore computes X and h1 by Newton-Schulz on the
// matrix cores (dense_small.hip loewdin_ns / loewdin_ns64_kernel) and keeps the eigensolver off the critical path
// (loewdin_split_mode below): either in the launch of the subspace solve (Ws.split = 3) or on a side stream, forked at
// the start of the call and joined by whichever call reads U and s next (Ws.split = 1):
// one side stream per device and two events per workspace, created at the workspace's first such call; the events live
// until evc_release_workspace, the stream until the last workspace that used it is released.
struct Side {
    hipStream_t s;           // the device's side stream (shared by all workspaces on it: one more hardware queue in use,
                             // not one per workspace -- the runtime multiplexes all streams onto four of them, and a
                             // process whose streams outnumber them sees unrelated streams serialised)
    hipEvent_t fork, join;   // of this workspace
    int dev;
    bool pending;            // an eigensolver launch into this workspace has not been joined yet
};
struct SideStream {
    hipStream_t s;
    int users;               // workspaces holding events on it; destroyed with the last one
};
static std::mutex g_side_mu;
static std::unordered_map<void *, Side> g_side;
static std::unordered_map<int, SideStream> g_side_stream;   // by device

static Side *side_of(void *ws) {
    std::lock_guard<std::mutex> lk(g_side_mu);
    auto it = g_side.find(ws);
    if (it != g_side.end()) return &it->second;
    int dev = 0;
    Side sd{};
    if (hipGetDevice(&dev) != hipSuccess) return nullptr;
    auto ds = g_side_stream.find(dev);
    if (ds == g_side_stream.end()) {
        hipStream_t ns;
        if (hipStreamCreateWithFlags(&ns, hipStreamNonBlocking) != hipSuccess) {
            set_error("side stream: %s", hipGetErrorString(hipGetLastError()));
            return nullptr;
        }
        ds = g_side_stream.emplace(dev, SideStream{ns, 0}).first;
    }
    sd.s = ds->second.s;
    sd.dev = dev;
    if (hipEventCreateWithFlags(&sd.fork, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&sd.join, hipEventDisableTiming) != hipSuccess) {
        set_error("side stream events: %s", hipGetErrorString(hipGetLastError()));
        return nullptr;
    }
    ++ds->second.users;
    return &g_side.emplace(ws, sd).first->second;
}

// The join: whoever reads U and s of a workspace next (launch_grad_final -- in the same call or, after an energy-only
// call, in a later evc_phase_gradient on the same workspace) waits for the eigensolver launch that writes them.
static int side_join(void *ws, hipStream_t st) {
    std::lock_guard<std::mutex> lk(g_side_mu);
    auto it = g_side.find(ws);
    if (it == g_side.end() || !it->second.pending) return 0;
    EVC_HIP(hipStreamWaitEvent(st, it->second.join, 0));
    it->second.pending = false;
    return 0;
}

extern "C" int evc_release_workspace(void *ws) {
    std::lock_guard<std::mutex> lk(g_side_mu);
    auto it = g_side.find(ws);
    if (it == g_side.end()) return 0;
    (void)hipEventSynchronize(it->second.join);   // (the last eigensolver launch that writes into this workspace)
    (void)hipEventDestroy(it->second.fork);
    (void)hipEventDestroy(it->second.join);
    auto ds = g_side_stream.find(it->second.dev);
    if (ds != g_side_stream.end() && --ds->second.users == 0) {
        (void)hipStreamDestroy(ds->second.s);   // (idle: every launch on it was followed by a join event, all waited for)
        g_side_stream.erase(ds);
    }
    g_side.erase(it);
    return 0;
}

// which form the Loewdin step of a FULL call (evc_energy_with_grad[_batch]) takes
static int loewdin_split_mode(int n, int ntrain, int count, bool loewdin_done, bool energy_only, bool warm,
                              hipStream_t st) {
    // EVC_LOEWDIN_SPLIT=0: the one-kernel Loewdin step always.
    static const int knob = getenv("EVC_LOEWDIN_SPLIT") ? atoi(getenv("EVC_LOEWDIN_SPLIT")) : 12;
    if (knob == 0 || loewdin_done || !loewdin_split_available(n)) return 0;
    // Small kernels on both sides (n <= 32 orbitals, T <= 32 states), any number of geometries, cold or warm: the
    // eigensolver half rides in the launch of the subspace solve, one workgroup per geometry beside one workgroup per
    // geometry (dense_small.hip subspace_loewdin_kernel) -- no second stream.  One geometry per call it performs like the
    // side stream below (H30: 4 480 against 4 500 steps/s, H10: 11 170 against 11 210) without costing the process a
    // hardware queue; 32 geometries per call on one stream: 60 700 -> 64 500 geometries/s, three streams unchanged.
    if (n <= kPairTransformMaxN && ntrain <= kSubspaceSmallT) return 3;
    // Otherwise (33 ... 64 orbitals: the 1024-thread eigensolver, 550 us at n = 58, has no launch to ride in; or a large
    // training set) the side stream, for calls of fewer than `knob` geometries (default 12: the latency regime).  Not the
    // large batches: with several of them in flight on different streams the chip is full anyway and a fifth stream
    // shares a hardware queue with one of them (measured at H30, 32 geometries per call, three streams: 87 000 -> 73 700).
    if (count >= knob) return 0;
    if (warm && n <= kPairTransformMaxN) return 0;
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(st, &cs) != hipSuccess || cs != hipStreamCaptureStatusNone) return 0;
    // (energy-only calls as well: a later evc_phase_gradient on the same workspace reads U and s -- hosted.py uploads the
    //  gradient's inputs in between)
    (void)energy_only;
    return 1;
}

// Span plan of this call (never more spans than the buffers were carved for).
static void replan(const evc_trdm_set *t, Ws &w, int count) {
    const bool batched = count > 1;
    if (t->rows2 > 0 && rows_groups_all_mfma(count) && rows_lds_applicable(w.rp2, w.rp1)) {
        plan_rows_lds(w.rp2, w.rp1);
        return;
    }
    if (t->rows2 > 0) plan_rows(w.rp2, batched);
    plan_rows(w.rp1, batched);
}

// rows_out != NULL: the scaled two-body rows go to rows_out[g*srows_out + r] (r local) instead of the workspace.
static int phase_hamiltonian(const evc_trdm_set *t, const Geo &g_in, Ws &w, bool reduce_rows, hipStream_t st,
                             double *rows_out = nullptr, int64_t srows_out = 0) {
    const int n = t->n, cnt = g_in.count;
    Geo g = g_in;
    const bool p64 = use_pair64(t->layout, n, g.eri_s4 != 0);
    if (g.eri_s4) {
        EVC_REQUIRE(is_sym8(t->layout) && (use_pair_transform(n) || p64),
                    "EVC_FLAG_ERI_S4 needs the compressed layout (EVC_LAYOUT_SYM8) and N <= 64");
        const int64_t npr = (int64_t)n * (n + 1) / 2;
        if (cnt > 1) g.seri = npr * npr;
    }
    const int64_t sw = w.stride;
    int rc;
    replan(t, w, cnt);
    LoewdinArgs la{};
    la.S = g.S;
    la.h = g.hcore;
    la.X = w.X;
    la.U = w.U;
    la.s = w.s;
    la.h1 = w.h1;
    la.sS = g.sS;
    la.sh = g.sh;
    la.sws = sw;
    la.n = n;
    la.warm = w.warm ? 1 : 0;
    la.scratch = w.B1;   // (free until the integral rotation; n > 64 only)
    la.sscratch = sw;
    la.flag = w.lflag;
    if (!w.loewdin_done) {
        if (w.split == 1) {
            // the eigendecomposition of S (U, s: read by launch_grad_final alone) on the side stream, forked here: the
            // inputs are ready, U of the previous call has been consumed
            Side *sd = side_of(w.base);
            if (!sd) return -1;
            EVC_HIP(hipEventRecord(sd->fork, st));
            EVC_HIP(hipStreamWaitEvent(sd->s, sd->fork, 0));
            la.part = 2;
            if ((rc = launch_loewdin(la, cnt, sd->s))) return rc;
            EVC_HIP(hipEventRecord(sd->join, sd->s));
            sd->pending = true;
        }
        la.part = w.split ? 1 : 0;
        const int pr = prof_start(EVC_PROF_LOEWDIN, st);
        if ((rc = launch_loewdin(la, cnt, st))) return rc;
        prof_stop(pr, st);
        if (w.split == 3) w.la_ride = la;
    }
    // (ab|cd) -> K3[jkl][a] -> h2[ijkl]
    const double *v2;
    if (use_pair_transform(n) || p64) {
        // two fused pair steps; the second one emits K3 and writes h2 straight into the form the
        // streaming kernel consumes (packed with diag x 1/2, or full)
        // in chunks of geometries, so that the intermediate of a chunk (6.5 MB per geometry) is still in the
        // 256 MB Infinity Cache when the second step reads it
        v2 = is_packed(t->layout) ? w.vec2 : w.B2;
        const int chunk = stage_chunk(cnt);
        for (int c0 = 0; c0 < cnt; c0 += chunk) {
            const int cc = cnt - c0 < chunk ? cnt - c0 : chunk;
            const int64_t o = (int64_t)c0 * sw;
            PairTransformArgs pa;
            memset(&pa, 0, sizeof(pa));
            pa.in = g.eri + (int64_t)c0 * g.seri;
            pa.sin = g.seri;
            pa.C = w.X + o;
            pa.sC = sw;
            pa.n = n;
            const bool fused_y2 = use_fused_y2(is_sym8(t->layout), n);
            double *mid = (fused_y2 ? w.K3 : w.B1) + o;   // the intermediate; kept for the gradient phase when fused
            pa.out = mid;
            pa.sout = sw;
            // compressed layout: the AO integrals are 8-fold symmetric by contract, the first step only
            // produces the q <= p half of its output and the second one reads the lower triangles
            // (in_lower: eri[p,q,r,s] = eri[p,q,s,r]; rs_lower: the next step's leading pairs are (r',s'), s' <= r')
            pa.lead_sym = pa.in_lower = pa.rs_lower = is_sym8(t->layout) ? 1 : 0;
            pa.in_pairs = g.eri_s4;       // int2e handed over as the dense (pair, pair) matrix (EVC_FLAG_ERI_S4)
            pa.in_ld = 0;                 // ... at the caller's pitch n(n+1)/2
            pa.out_pairs = pa.lead_sym;   // the intermediate as a dense (pair, pair) matrix
            pa.out_ld = pair_ld(n);
            int pr = prof_start(EVC_PROF_PAIR_TRANSFORM, st);
            if ((rc = launch_pair_transform(pa, cc, st))) return rc;
            prof_stop(pr, st);
            pa.in_pairs = pa.out_pairs;
            pa.in_ld = pa.out_ld;
            pa.out_pairs = 0;
            pa.out_ld = 0;
            // ... and the second step again only needs the q <= p half of ITS leading pair
            pa.in = mid;
            pa.sin = sw;
            pa.k3 = fused_y2 ? nullptr : w.K3 + o;
            pa.sk3 = sw;
            if (is_packed(t->layout)) {
                pa.out = nullptr;
                pa.packed = w.vec2 + o;
                pa.spacked = sw;
                pa.packed_len = t->ld2;
                pa.diag_mult = 0.5;
                pa.sym8 = is_sym8(t->layout) ? 1 : 0;
            } else {
                pa.out = w.B2 + o;
            }
            pr = prof_start(EVC_PROF_PAIR_TRANSFORM, st);
            if ((rc = launch_pair_transform(pa, cc, st))) return rc;
            prof_stop(pr, st);
        }
    } else {
        if ((rc = launch_quarter_transform(g.eri, g.seri, w.X, sw, 0, n, w.B1, sw, cnt, st))) return rc;
        if ((rc = launch_quarter_transform(w.B1, sw, w.X, sw, 0, n, w.B2, sw, cnt, st))) return rc;
        if ((rc = launch_quarter_transform(w.B2, sw, w.X, sw, 0, n, w.K3, sw, cnt, st))) return rc;
        if ((rc = launch_quarter_transform(w.K3, sw, w.X, sw, 0, n, w.B1, sw, cnt, st))) return rc;
        v2 = w.B1;
        if (is_packed(t->layout)) {
            rc = is_sym8(t->layout) ? launch_pack_sym8(w.B1, sw, n, 0.5, w.vec2, sw, t->ld2, cnt, st)
                                    : launch_pack(w.B1, sw, n, 0.5, w.vec2, sw, t->ld2, cnt, st);
            if (rc) return rc;
            v2 = w.vec2;
        }
    }
    RowProblem p2 = w.rp2, p1 = w.rp1;
    p2.A = t->two_rdm;
    p2.v = v2;
    p2.partial = w.h2part;
    p2.vstride = p2.pstride = sw;
    if (t->rows2 == 0) p2.nblocks = 0;
    p1.A = t->one_rdm;
    p1.v = w.h1;
    p1.partial = w.h1part;
    p1.vstride = p1.pstride = sw;
    const int pr = prof_start(EVC_PROF_ROWS, st);
    if ((rc = launch_gemv_rows(p2, p1, cnt, st))) return rc;
    prof_stop(pr, st);
    if ((reduce_rows || reduce_in_own_launch(w)) && t->rows2 > 0) {
        const double alpha2 = is_packed(t->layout) ? 1.0 : 0.5;
        hipLaunchKernelGGL(rows_reduce_kernel, dim3((unsigned)ceil_div(t->rows2, 16), (unsigned)cnt), dim3(256), 0,
                           st, w.h2part, sw, t->rows2, w.rp2.nspans, alpha2,
                           rows_out ? rows_out : w.h2rows + t->row_offset, rows_out ? srows_out : sw);
        EVC_LAUNCH_CHECK("rows_reduce");
    }
    return 0;
}

static int phase_solve(const evc_trdm_set *t, const Geo &g, const double *h2rows_all, int64_t sh2_all, const Out &out,
                       int nroots, Ws &w, hipStream_t st) {
    SolveArgs a;
    memset(&a, 0, sizeof(a));
    const int64_t sw = w.stride;
    a.h1part = w.h1part;
    a.nsp1 = w.rp1.nspans;
    a.alpha1 = 1.0;
    a.sh1 = sw;
    if (h2rows_all) {
        a.h2part = h2rows_all;
        a.nsp2 = 1;
        a.alpha2 = 1.0;
        a.sh2 = sh2_all;
    } else if (reduce_in_own_launch(w)) {
        a.h2part = w.h2rows;  // written by rows_reduce_kernel in phase A (complete t-RDM on this device)
        a.nsp2 = 1;
        a.alpha2 = 1.0;
        a.sh2 = sw;
    } else {
        a.h2part = w.h2part;
        a.nsp2 = w.rp2.nspans;
        a.alpha2 = is_packed(t->layout) ? 1.0 : 0.5;
        a.sh2 = sw;
    }
    a.S = t->s_train;
    a.T = t->ntrain;
    a.layout = t->layout;
    a.nroots = nroots;
    a.e_shift = g.enuc;
    a.e_shift_dev = g.enuc_dev;
    if (out.energy) {
        a.evals = out.energy;
        a.sev = out.se;
    } else {
        a.evals = w.evals;
        a.sev = sw;
    }
    if (out.coeffs) {
        a.evecs = out.coeffs;
        a.svec = out.sc;
    } else {
        a.evecs = w.evecs;
        a.svec = sw;
    }
    a.Hout = out.hmat;
    a.sH = out.sH;
    a.w2 = w.w2;
    a.w2t = g.count > 1 ? w.w2t : nullptr;
    a.w1t = g.count > 1 ? w.w1t : nullptr;
    a.w1 = w.w1;
    a.sw = sw;
    a.w2_offset = t->row_offset;
    a.w2_count = t->rows2;
    a.vstd = w.vstd;
    a.bcache = w.bcache;
    a.scratch = w.sbig;
    a.sscratch = sw;
    a.warm = w.warm ? 1 : 0;
    const int pr = prof_start(EVC_PROF_SUBSPACE, st);
    const int rc = w.split == 3 ? launch_subspace_loewdin(a, w.la_ride, g.count, st) : launch_subspace_solve(a, g.count, st);
    prof_stop(pr, st);
    return rc;
}

// Gradient of the energy functional defined by (D, G) [G unpacked, N^4] given X,U,s,K3 in the
// workspace.  scale1 = 0 drops everything that is not linear in G (multi-GPU partial ranks).
// `packed` != NULL selects the fast path for pair-symmetric (packed) predicted 2-RDMs: both symmetrisations
// are taken straight from the packed vector (G is then only written when the caller wants it, G may be NULL);
// sym8: the packed vector is the 8-fold compressed one (EVC_LAYOUT_SYM8).
static int gradient_from_rdms(int n, const Geo &g, const double *D, int64_t sD, double *G, int64_t sG,
                              const double *packed, int64_t spacked, int sym8, int ip1_s2kl, double scale1,
                              bool add_gnuc, double *grad, int64_t sgrad, Ws &w, hipStream_t st) {
    const int cnt = g.count;
    const int64_t sw = w.stride;
    int rc;
    double *gao;  // symmetrised or plain 2-RDM in the AO basis
    // G^AO = (X x X x X x X) G, contraction over the SECOND index of X (gradients_loewdin.py:224-232):
    // src -> ... -> dst, through `other`, as two fused pair steps (n <= 32) or four quarter steps
    auto rotate_to_ao = [&](const double *src, int64_t ssrc, double *other, double *dst) -> int {
        int r;
        if (use_pair_transform(n)) {
            PairTransformArgs pa;
            memset(&pa, 0, sizeof(pa));
            pa.C = w.X;
            pa.sC = sw;
            pa.ct = 1;
            pa.n = n;
            pa.in = src;
            pa.sin = ssrc;
            pa.out = other;
            pa.sout = sw;
            if ((r = launch_pair_transform(pa, cnt, st))) return r;
            pa.in = other;
            pa.sin = sw;
            pa.out = dst;
            return launch_pair_transform(pa, cnt, st);
        }
        if ((r = launch_quarter_transform(src, ssrc, w.X, sw, 1, n, dst, sw, cnt, st))) return r;
        if ((r = launch_quarter_transform(dst, sw, w.X, sw, 1, n, other, sw, cnt, st))) return r;
        if ((r = launch_quarter_transform(other, sw, w.X, sw, 1, n, dst, sw, cnt, st))) return r;
        if ((r = launch_quarter_transform(dst, sw, w.X, sw, 1, n, other, sw, cnt, st))) return r;
        // result sits in `other`: one more hop would cost a launch, so report where it is
        return 1 << 30;
    };
    GradPrepArgs p;
    p.n = n;
    p.X = w.X;
    p.hcore = g.hcore;
    p.D = D;
    p.Pao = w.Pao;
    p.Y1 = w.Y1;
    p.sws = sw;
    p.sh = g.sh;
    p.sD = sD;
    p.scale1 = scale1;
    // (symmetric pipeline without a request for the unpacked 2-RDM: grad_prep rides in the unpack launch below)
    // (n <= 32 only: every block of the shared launch reserves grad_prep's LDS -- 52 KB there, 108 KB at n = 58, where
    //  the unpack blocks would run one per CU)
    const bool prep_with_unpack = packed && sym8 && !G && use_pair_transform(n);
    if (!prep_with_unpack && (rc = launch_grad_prep(p, cnt, st))) return rc;
    // 32 < n <= 64: the symmetric pipeline on 64 x 64 matrices when int2e_ip1 came packed (use_pair64)
    const bool p64 = packed && use_pair64(sym8 ? EVC_LAYOUT_SYM8 : 0, n, ip1_s2kl != 0);
    const bool pairs_route = use_pair_transform(n) || p64;
    const bool fused_y2 = packed && pairs_route && use_fused_y2(sym8 != 0, n);
    int y2_slabs_used = y2_slabs(n);   // (the fused kernel: per chunk of geometries, set where it is launched)
    auto ip1_stage = [&](const double *gao_, int c0, int cc) -> int {
        const int64_t o = (int64_t)c0 * sw;
        Ip1Args ia;
        ia.ip1 = g.eri_ip1 + (int64_t)c0 * g.sip1;
        ia.Gao = gao_ + o;
        ia.presym = packed ? 1 : 0;
        ia.fold_cd = (packed && sym8 && pairs_route) ? 1 : 0;
        ia.ip1_s2kl = ip1_s2kl;
        ia.t2part = w.t2part + o;
        ia.dh = g.dhcore ? g.dhcore + (int64_t)c0 * g.sdh : nullptr;
        ia.Pao = w.Pao + o;
        ia.term3 = w.term3 + o;
        ia.y2part = w.y2part + o;
        ia.y2 = w.y2 + o;
        ia.sip1 = g.sip1;
        ia.sdh = g.sdh;
        ia.sws = sw;
        ia.n = n;
        ia.natm = g.natm;
        ia.nslab = y2_slabs_used;
        ia.nchunk = ip1_chunks(n);
        return launch_ip1_dh(ia, cc, st);
    };
    bool ip1_done = false;
    if (packed) {
        if (pairs_route) {
            // unpack+symmetrise -> Y2 -> B1 (symmetrised, OAO) -> B2 -> B1 (AO) -> ip1 contraction, in chunks of
            // geometries so that each kernel finds its predecessor's output in the Infinity Cache
            const int chunk = stage_chunk(cnt);
            for (int c0 = 0; c0 < cnt; c0 += chunk) {
                const int cc = cnt - c0 < chunk ? cnt - c0 : chunk;
                const int64_t o = (int64_t)c0 * sw;
                int pr = prof_start(EVC_PROF_UNPACK, st);
                if (sym8) {
                    // (K3 was written for l <= k only by the symmetric second step of phase A)
                    // (without a request for the unpacked 2-RDM, SB is the dense (pair, pair) matrix)
                    // (p64 with the unpacked 2-RDM requested: the N^4-addressed SB that comes with it is not used -- it
                    //  goes to B2, which the next step overwrites -- and B1 gets the dense form every step of this route reads)
                    if (prep_with_unpack && c0 == 0 && cc == cnt) {
                        if ((rc = launch_unpack8_prep(p, packed, spacked, w.B1, sw, cnt, st))) return rc;
                    } else {
                        if (prep_with_unpack && c0 == 0 && (rc = launch_grad_prep(p, cnt, st))) return rc;
                        if ((rc = launch_unpack8(packed + (int64_t)c0 * spacked, spacked, n, (G && p64 ? w.B2 : w.B1) + o, sw,
                                                 G ? G + (int64_t)c0 * sG : nullptr, sG, cc, G ? 1 : 2, st)))
                            return rc;
                    }
                    if (G && p64 &&
                        (rc = launch_unpack8(packed + (int64_t)c0 * spacked, spacked, n, w.B1 + o, sw, nullptr, 0, cc, 2, st)))
                        return rc;
                    prof_stop(pr, st);
                    pr = prof_start(EVC_PROF_Y2, st);
                    if (fused_y2) {
                        // (the K3 buffer holds the first pair step's intermediate; with the unpacked 2-RDM requested
                        //  SB above is N^4-addressed: the dense (pair, pair) form goes to B2, free until the next step)
                        const double *sbp = w.B1 + o;
                        if (G && !p64) {
                            if ((rc = launch_unpack8(packed + (int64_t)c0 * spacked, spacked, n, w.B2 + o, sw, nullptr, 0,
                                                     cc, 2, st)))
                                return rc;
                            sbp = w.B2 + o;
                        }
                        if ((rc = launch_y2_fused(sbp, w.K3 + o, w.X + o, sw, n, w.y2part + o, sw, cc, st))) return rc;
                        y2_slabs_used = y2_fused_slabs(n, cc);
                    } else {
                        set_error("gradient: the symmetric pipeline needs the fused Y2 contraction (n <= 32)");
                        return -1;
                    }
                } else {
                    if ((rc = launch_unpack_sym(packed + (int64_t)c0 * spacked, spacked, n, w.B2 + o, w.B1 + o, sw,
                                                G ? G + (int64_t)c0 * sG : nullptr, sG, cc, st)))
                        return rc;
                    prof_stop(pr, st);
                    pr = prof_start(EVC_PROF_Y2, st);
                    if ((rc = launch_y2(w.B2 + o, w.K3 + o, n, w.y2part + o, sw, cc, st))) return rc;
                }
                prof_stop(pr, st);
                PairTransformArgs pa;
                memset(&pa, 0, sizeof(pa));
                pa.C = w.X + o;
                pa.sC = sw;
                pa.ct = 1;
                pa.n = n;
                pa.in = w.B1 + o;
                pa.sin = sw;
                pa.out = w.B2 + o;
                pa.sout = sw;
                pa.lead_sym = pa.in_lower = pa.rs_lower = sym8;   // SB is fully symmetric
                pa.in_pairs = (sym8 && (!G || p64)) ? 1 : 0;
                pa.out_pairs = sym8;
                pa.in_ld = pa.out_ld = pair_ld(n);   // (pitch of every dense (pair, pair) form of the pipeline)
                pr = prof_start(EVC_PROF_PAIR_TRANSFORM, st);
                if ((rc = launch_pair_transform(pa, cc, st))) return rc;
                prof_stop(pr, st);
                // (the second step keeps rs_lower as well: G^AO[m,b,c,d] = G^AO[b,m,c,d], fold_cd reads b <= m)
                // the result is only valid for d <= c of G^AO[m,b,c,d] (fold_cd below)
                pa.in = w.B2 + o;
                pa.out = w.B1 + o;
                pa.in_pairs = pa.out_pairs;
                pa.out_pairs = ip1_s2kl ? 1 : 0;   // the packed-ip1 dot wants the dense (pair, pair) form (it weighs it itself)
                pr = prof_start(EVC_PROF_PAIR_TRANSFORM, st);
                if ((rc = launch_pair_transform(pa, cc, st))) return rc;
                prof_stop(pr, st);
                pr = prof_start(EVC_PROF_IP1, st);
                if ((rc = ip1_stage(w.B1, c0, cc))) return rc;
                prof_stop(pr, st);
            }
            ip1_done = true;
        } else {
            if (sym8) {
                if ((rc = launch_unpack8(packed, spacked, n, w.B1, sw, G, sG, cnt, 0, st))) return rc;
                if ((rc = launch_y2_sb(w.B1, w.K3, n, w.y2part, sw, cnt, st))) return rc;
            } else {
                if ((rc = launch_unpack_sym(packed, spacked, n, w.B2, w.B1, sw, G, sG, cnt, st))) return rc;
                if ((rc = launch_y2(w.B2, w.K3, n, w.y2part, sw, cnt, st))) return rc;
            }
            if ((rc = launch_quarter_transform(w.B1, sw, w.X, sw, 1, n, w.B2, sw, cnt, st))) return rc;
            if ((rc = launch_quarter_transform(w.B2, sw, w.X, sw, 1, n, w.B1, sw, cnt, st))) return rc;
            if ((rc = launch_quarter_transform(w.B1, sw, w.X, sw, 1, n, w.B2, sw, cnt, st))) return rc;
            if ((rc = launch_quarter_transform(w.B2, sw, w.X, sw, 1, n, w.B1, sw, cnt, st))) return rc;
        }
        gao = w.B1;
    } else {
        if ((rc = launch_sym_oao_t(G, sG, n, w.B2, sw, cnt, st))) return rc;
        if ((rc = launch_y2(w.B2, w.K3, n, w.y2part, sw, cnt, st))) return rc;
        rc = rotate_to_ao(G, sG, w.B1, w.B2);
        if (rc == (1 << 30)) gao = w.B1;       // quarter-step route ends in `other`
        else if (rc) return rc;
        else gao = w.B2;
    }
    if (!ip1_done && (rc = ip1_stage(gao, 0, cnt))) return rc;
    if ((rc = side_join(w.base, st))) return rc;   // U and s may come from the side stream (phase_hamiltonian)
    GradFinalArgs f;
    f.n = n;
    f.natm = g.natm;
    f.U = w.U;
    f.s = w.s;
    f.Y1 = w.Y1;
    f.y2 = w.y2;
    f.ipovlp = g.ipovlp;
    f.aoslices = g.aoslices;
    f.t2part = w.t2part;
    f.nchunk = ip1_chunks(n);
    f.term3 = w.term3;
    f.gnuc = add_gnuc ? g.gnuc : nullptr;
    f.scale1 = scale1;
    f.grad = grad;
    f.sws = sw;
    f.sip = g.sip;
    f.sgn = g.sgn;
    f.sgrad = sgrad;
    return launch_grad_final(f, cnt, st);
}

static int phase_gradient(const evc_trdm_set *t, const Geo &g_in, const Out &out, int flags, Ws &w, hipStream_t st) {
    const int n = t->n, cnt = g_in.count;
    Geo g = g_in;
    const int s2kl = (flags & EVC_FLAG_IP1_S2KL) ? 1 : 0;
    if (s2kl) {
        EVC_REQUIRE(is_sym8(t->layout) && (use_pair_transform(n) || use_pair64(t->layout, n, true)),
                    "EVC_FLAG_IP1_S2KL needs the compressed layout (EVC_LAYOUT_SYM8) and N <= 64");
        if (cnt > 1) g.sip1 = (int64_t)3 * n * n * (n * (n + 1) / 2);
    }
    const int64_t sw = w.stride;
    int rc;
    double *D = out.d_pred ? out.d_pred : w.Dpred;
    const int64_t sD = out.d_pred ? out.sd : sw;
    double *G = out.g_pred ? out.g_pred : w.G;
    const int64_t sG = out.g_pred ? out.sG : sw;
    ColProblem c2{}, c1{};
    c2.A = t->two_rdm;
    c2.w = w.w2;
    c2.wt = cnt > 1 ? w.w2t : nullptr;
    c2.wstride = sw;
    c2.rows = t->rows2;
    c2.cols = t->cols2;
    c2.ld = t->ld2;
    if (is_packed(t->layout)) {
        c2.out = w.vec2;
        c2.ostride = sw;
    } else {
        c2.out = G;
        c2.ostride = sG;
    }
    c1.A = t->one_rdm;
    c1.w = w.w1;
    c1.wt = cnt > 1 ? w.w1t : nullptr;
    c1.wstride = sw;
    c1.rows = (int64_t)t->ntrain * t->ntrain;
    c1.cols = (int64_t)n * n;
    c1.ld = t->ld1;
    c1.out = D;
    c1.ostride = sD;
    c1.part = (int64_t)t->ntrain * t->ntrain >= 1024 ? w.d1part : nullptr;
    c1.pstride = sw;
    const int pr = prof_start(EVC_PROF_COLS, st);
    if ((rc = launch_gemv_cols(c2, c1, cnt, st))) return rc;
    prof_stop(pr, st);
    const bool partial = (flags & EVC_FLAG_PARTIAL_RANK) != 0;
    if (is_packed(t->layout))
        // the unpacked 2-RDM is only materialised when the caller asked for it
        return gradient_from_rdms(n, g, D, sD, out.g_pred, out.sG, w.vec2, sw, is_sym8(t->layout) ? 1 : 0, s2kl,
                                  partial ? 0.0 : 1.0, !partial, out.grad, out.sg, w, st);
    return gradient_from_rdms(n, g, D, sD, G, sG, nullptr, 0, 0, 0, partial ? 0.0 : 1.0, !partial, out.grad, out.sg,
                              w, st);
}

static int check_geometry(const evc_geometry *g, bool need_grad) {
    EVC_REQUIRE(g != nullptr, "geometry is NULL");
    EVC_REQUIRE(g->S && g->hcore && g->eri, "geometry: S/hcore/eri must be given");
    // (the pair kernels fetch the rows of the two large arrays through 16-byte windows)
    EVC_REQUIRE(aligned16(g->eri) && (!g->eri_ip1 || aligned16(g->eri_ip1)), "geometry: eri / eri_ip1 must be 16-byte aligned");
    if (need_grad) {
        EVC_REQUIRE(g->natm >= 1, "geometry: natm=%d", g->natm);
        EVC_REQUIRE(g->ipovlp && g->dhcore && g->eri_ip1 && g->aoslices,
                    "geometry: ipovlp/dhcore/eri_ip1/aoslices are required for the gradient");
    }
    return 0;
}

static Geo geo_single(const evc_geometry *g) {
    Geo o;
    memset(&o, 0, sizeof(o));
    o.natm = g->natm;
    o.count = 1;
    o.S = g->S;
    o.hcore = g->hcore;
    o.eri = g->eri;
    o.ipovlp = g->ipovlp;
    o.dhcore = g->dhcore;
    o.eri_ip1 = g->eri_ip1;
    o.gnuc = g->gnuc;
    o.aoslices = g->aoslices;
    o.enuc = g->enuc;
    return o;
}

static Out out_single(const evc_outputs *o) {
    Out r;
    memset(&r, 0, sizeof(r));
    if (o) {
        r.energy = o->energy;
        r.coeffs = o->coeffs;
        r.grad = o->grad;
        r.d_pred = o->d_pred;
        r.g_pred = o->g_pred;
        r.hmat = o->hmat;
    }
    return r;
}

}  // namespace evc

using namespace evc;

extern "C" int evc_abi_version(void) { return EVC_ABI_VERSION; }
extern "C" const char *evc_last_error(void) { return g_err; }

extern "C" int evc_profile_begin(int max_samples) {
    std::lock_guard<std::mutex> lk(g_prof_mu);
    EVC_REQUIRE(!g_prof.on, "evc_profile_begin: already profiling");
    EVC_REQUIRE(max_samples > 0 && max_samples <= 1 << 16, "evc_profile_begin: max_samples=%d", max_samples);
    const int cap = max_samples * kProfPerSample;
    g_prof.ev = new hipEvent_t[2 * (size_t)cap];
    g_prof.stage = new int[cap];
    for (int i = 0; i < 2 * cap; ++i) {
        hipError_t e = hipEventCreate(&g_prof.ev[i]);
        if (e != hipSuccess) {
            set_error("evc_profile_begin: hipEventCreate: %s", hipGetErrorString(e));
            return (int)e;
        }
    }
    g_prof.cap = cap;
    g_prof.n = 0;
    g_prof.on = true;
    return 0;
}

extern "C" int evc_profile_end(double *rows_ms, int *rows_n, double *cols_ms, int *cols_n) {
    std::lock_guard<std::mutex> lk(g_prof_mu);
    EVC_REQUIRE(g_prof.on, "evc_profile_end: not profiling");
    for (int k = 0; k < kProfStages; ++k) {
        g_prof.ms[k] = 0.0;
        g_prof.cnt[k] = 0;
    }
    for (int i = 0; i < g_prof.n; ++i) {
        float ms = 0.f;
        (void)hipEventSynchronize(g_prof.ev[2 * i + 1]);
        (void)hipEventElapsedTime(&ms, g_prof.ev[2 * i], g_prof.ev[2 * i + 1]);
        const int k = g_prof.stage[i];
        g_prof.ms[k] += ms;
        g_prof.cnt[k] += 1;
    }
    if (rows_ms) *rows_ms = g_prof.ms[EVC_PROF_ROWS];
    if (rows_n) *rows_n = g_prof.cnt[EVC_PROF_ROWS];
    if (cols_ms) *cols_ms = g_prof.ms[EVC_PROF_COLS];
    if (cols_n) *cols_n = g_prof.cnt[EVC_PROF_COLS];
    for (int i = 0; i < 2 * g_prof.cap; ++i) (void)hipEventDestroy(g_prof.ev[i]);
    delete[] g_prof.ev;
    delete[] g_prof.stage;
    g_prof.ev = nullptr;
    g_prof.stage = nullptr;
    g_prof.cap = g_prof.n = 0;
    g_prof.on = false;
    return 0;
}

extern "C" const char *evc_profile_kernel(int stage) {
    return (stage >= 0 && stage < kProfStages) ? g_kernel_ran[stage] : "";
}

extern "C" int evc_profile_select(unsigned stage_mask) {
    std::lock_guard<std::mutex> lk(g_prof_mu);
    EVC_REQUIRE(!g_prof.on, "evc_profile_select: not while profiling");
    g_prof.mask = stage_mask & ((1u << kProfStages) - 1u);
    return 0;
}

extern "C" int evc_profile_stage(int stage, double *ms, int *launches) {
    std::lock_guard<std::mutex> lk(g_prof_mu);
    EVC_REQUIRE(stage >= 0 && stage < kProfStages, "evc_profile_stage: stage=%d", stage);
    EVC_REQUIRE(!g_prof.on, "evc_profile_stage: call evc_profile_end first");
    if (ms) *ms = g_prof.ms[stage];
    if (launches) *launches = g_prof.cnt[stage];
    return 0;
}

extern "C" size_t evc_workspace_bytes(const evc_trdm_set *t, int natm) {
    if (check_set(t)) return 0;
    Ws w;
    carve(t, natm, nullptr, w);
    return w.bytes;
}

extern "C" size_t evc_workspace_bytes_batch(const evc_trdm_set *t, int natm, int count) {
    if (check_set(t) || count < 1) return 0;
    Ws w;
    carve(t, natm, nullptr, w);
    return w.bytes * (size_t)count;
}

#define EVC_SETUP(need_grad)                                                                      \
    if (check_set(t)) return -1;                                                                  \
    if (check_geometry(g, need_grad)) return -1;                                                  \
    EVC_REQUIRE(ws && aligned16(ws), "workspace NULL or misaligned");                             \
    Ws w;                                                                                         \
    carve(t, g->natm, static_cast<char *>(ws), w);                                                \
    EVC_REQUIRE(ws_bytes >= w.bytes, "workspace too small: %zu < %zu", ws_bytes, w.bytes);        \
    replan(t, w, 1);                                                                              \
    hipStream_t st = as_stream(stream);                                                           \
    const Geo geo = geo_single(g)

extern "C" int evc_phase_hamiltonian(const evc_trdm_set *t, const evc_geometry *g, int flags, void *ws, size_t ws_bytes,
                                     double **h2rows_local, double **h1rows, void *stream) {
    EVC_SETUP(false);
    w.warm = (flags & EVC_FLAG_WARM_START) != 0;
    Geo gg = geo;
    gg.eri_s4 = (flags & EVC_FLAG_ERI_S4) ? 1 : 0;
    int rc = phase_hamiltonian(t, gg, w, true, st);
    if (rc) return rc;
    if (h2rows_local) *h2rows_local = w.h2rows + t->row_offset;
    if (h1rows) *h1rows = w.h1part;
    return 0;
}

extern "C" int evc_phase_solve(const evc_trdm_set *t, const evc_geometry *g, const double *h2rows_all,
                               const evc_outputs *out, int nroots, int flags, void *ws, size_t ws_bytes, void *stream) {
    EVC_SETUP(false);
    EVC_REQUIRE(nroots >= 1 && nroots <= t->ntrain, "nroots=%d out of range 1..%d", nroots, t->ntrain);
    w.warm = (flags & EVC_FLAG_WARM_START) != 0;
    return phase_solve(t, geo, h2rows_all ? h2rows_all : w.h2rows, 0, out_single(out), nroots, w, st);
}

extern "C" int evc_phase_set_coeffs(const evc_trdm_set *t, const double *coeffs, int natm, void *ws, size_t ws_bytes,
                                    void *stream) {
    if (check_set(t)) return -1;
    EVC_REQUIRE(coeffs, "evc_phase_set_coeffs: coeffs is NULL");
    EVC_REQUIRE(ws && aligned16(ws), "workspace NULL or misaligned");
    Ws w;
    carve(t, natm, static_cast<char *>(ws), w);
    EVC_REQUIRE(ws_bytes >= w.bytes, "workspace too small: %zu < %zu", ws_bytes, w.bytes);
    return launch_pair_weights(coeffs, t->ntrain, t->layout, w.w1, w.w2, t->row_offset, t->rows2, as_stream(stream));
}

extern "C" int evc_phase_gradient(const evc_trdm_set *t, const evc_geometry *g, const evc_outputs *out,
                                  int flags, void *ws, size_t ws_bytes, void *stream) {
    EVC_SETUP(true);
    EVC_REQUIRE(out && out->grad, "outputs.grad is required");
    return phase_gradient(t, geo, out_single(out), flags, w, st);
}

extern "C" int evc_energy_with_grad(const evc_trdm_set *t, const evc_geometry *g, const evc_outputs *out,
                                    int nroots, int flags, void *ws, size_t ws_bytes, void *stream) {
    const bool energy_only = (flags & EVC_FLAG_ENERGY_ONLY) != 0;
    EVC_SETUP(!energy_only);
    EVC_REQUIRE(out != nullptr, "outputs is NULL");
    EVC_REQUIRE(nroots >= 1 && nroots <= t->ntrain, "nroots=%d out of range 1..%d", nroots, t->ntrain);
    EVC_REQUIRE(t->rows2 == t->rows2_total && t->row_offset == 0,
                "evc_energy_with_grad needs the complete t-RDM on this device (use the phase calls when sharded)");
    EVC_REQUIRE(energy_only || out->grad, "outputs.grad is required unless EVC_FLAG_ENERGY_ONLY");
    const Out o = out_single(out);
    int rc;
    EVC_REQUIRE(energy_only || t->n <= kPairTransformMaxN || !(flags & EVC_FLAG_ERI_S4) == !(flags & EVC_FLAG_IP1_S2KL),
                "N > 32: EVC_FLAG_ERI_S4 and EVC_FLAG_IP1_S2KL go together (both packed inputs, or neither)");
    w.warm = (flags & EVC_FLAG_WARM_START) != 0;
    w.split = loewdin_split_mode(t->n, t->ntrain, 1, false, energy_only, w.warm, st);
    Geo gg = geo;
    gg.eri_s4 = (flags & EVC_FLAG_ERI_S4) ? 1 : 0;
    if ((rc = phase_hamiltonian(t, gg, w, false, st))) return rc;
    if ((rc = phase_solve(t, gg, nullptr, 0, o, nroots, w, st))) return rc;
    if (energy_only) return 0;
    return phase_gradient(t, gg, o, flags & ~EVC_FLAG_PARTIAL_RANK, w, st);
}

// Shared argument checking / descriptor set-up of the batch entry points.
static int setup_batch(const char *who, const evc_trdm_set *t, const evc_geometry_batch *gb,
                       const evc_outputs_batch *ob, bool need_grad, void *ws, size_t ws_bytes, Ws &w, Geo &g, Out &o) {
    if (check_set(t)) return -1;
    EVC_REQUIRE(gb, "%s: null batch descriptor", who);
    EVC_REQUIRE(gb->count >= 1 && gb->count <= 4096, "%s: batch count=%d out of range", who, gb->count);
    EVC_REQUIRE(gb->S && gb->hcore && gb->eri && gb->enuc, "%s: batch geometry: S/hcore/eri/enuc must be given", who);
    EVC_REQUIRE(aligned16(gb->eri) && (!gb->eri_ip1 || aligned16(gb->eri_ip1)),
                "%s: batch geometry: eri / eri_ip1 must be 16-byte aligned", who);
    if (need_grad) {
        EVC_REQUIRE(gb->natm >= 1 && gb->ipovlp && gb->dhcore && gb->eri_ip1 && gb->aoslices && gb->gnuc,
                    "%s: batch geometry: ipovlp/dhcore/eri_ip1/gnuc/aoslices are required for the gradient", who);
        EVC_REQUIRE(ob && ob->grad, "%s: batch outputs.grad is required", who);
    }
    EVC_REQUIRE(ws && aligned16(ws), "%s: workspace NULL or misaligned", who);
    carve(t, gb->natm, static_cast<char *>(ws), w);
    EVC_REQUIRE(ws_bytes >= w.bytes * (size_t)gb->count, "%s: workspace too small: %zu < %zu", who, ws_bytes,
                w.bytes * (size_t)gb->count);
    const int64_t n = t->n, n2 = n * n, n4 = n2 * n2, T = t->ntrain, A = gb->natm;
    memset(&g, 0, sizeof(g));
    g.natm = gb->natm;
    g.count = gb->count;
    g.S = gb->S;
    g.sS = n2;
    g.hcore = gb->hcore;
    g.sh = n2;
    g.eri = gb->eri;
    g.seri = n4;
    g.ipovlp = gb->ipovlp;
    g.sip = 3 * n2;
    g.dhcore = gb->dhcore;
    g.sdh = A * 3 * n2;
    g.eri_ip1 = gb->eri_ip1;
    g.sip1 = 3 * n4;
    g.gnuc = gb->gnuc;
    g.sgn = A * 3;
    g.aoslices = gb->aoslices;
    g.enuc_dev = gb->enuc;
    memset(&o, 0, sizeof(o));
    if (ob) {
        o.energy = ob->energy;
        o.se = T;
        o.coeffs = ob->coeffs;
        o.sc = T * T;
        o.grad = ob->grad;
        o.sg = A * 3;
        o.d_pred = ob->d_pred;
        o.sd = n2;
        o.g_pred = ob->g_pred;
        o.sG = n4;
        o.hmat = ob->hmat;
        o.sH = T * T;
    }
    return 0;
}

extern "C" int evc_energy_with_grad_batch(const evc_trdm_set *t, const evc_geometry_batch *gb,
                                          const evc_outputs_batch *ob, int nroots, int flags, void *ws,
                                          size_t ws_bytes, void *stream) {
    const bool energy_only = (flags & EVC_FLAG_ENERGY_ONLY) != 0;
    Ws w;
    Geo g;
    Out o;
    if (setup_batch("evc_energy_with_grad_batch", t, gb, ob, !energy_only, ws, ws_bytes, w, g, o)) return -1;
    EVC_REQUIRE(ob && ob->energy && ob->coeffs, "batch outputs.energy/coeffs are required");
    EVC_REQUIRE(nroots >= 1 && nroots <= t->ntrain, "nroots=%d out of range 1..%d", nroots, t->ntrain);
    EVC_REQUIRE(t->rows2 == t->rows2_total && t->row_offset == 0,
                "evc_energy_with_grad_batch needs the complete t-RDM on this device (use the phase calls when sharded)");
    hipStream_t st = as_stream(stream);
    int rc;
    w.warm = (flags & EVC_FLAG_WARM_START) != 0;
    EVC_REQUIRE(energy_only || t->n <= kPairTransformMaxN || !(flags & EVC_FLAG_ERI_S4) == !(flags & EVC_FLAG_IP1_S2KL),
                "N > 32: EVC_FLAG_ERI_S4 and EVC_FLAG_IP1_S2KL go together (both packed inputs, or neither)");
    w.loewdin_done = (flags & EVC_FLAG_LOEWDIN_DONE) != 0;
    w.split = loewdin_split_mode(t->n, t->ntrain, g.count, w.loewdin_done, energy_only, w.warm, st);
    g.eri_s4 = (flags & EVC_FLAG_ERI_S4) ? 1 : 0;
    if ((rc = phase_hamiltonian(t, g, w, false, st))) return rc;
    if ((rc = phase_solve(t, g, nullptr, 0, o, nroots, w, st))) return rc;
    if (energy_only) return 0;
    return phase_gradient(t, g, o, flags & EVC_FLAG_IP1_S2KL, w, st);
}

extern "C" int evc_phase_loewdin_batch(const evc_trdm_set *t, const evc_geometry_batch *gb, int flags, void *ws,
                                       size_t ws_bytes, void *stream) {
    if (check_set(t)) return -1;
    EVC_REQUIRE(gb && gb->count >= 1 && gb->count <= 4096 && gb->S && gb->hcore,
                "evc_phase_loewdin_batch: batch descriptor / S / hcore missing");
    EVC_REQUIRE(ws && aligned16(ws), "evc_phase_loewdin_batch: workspace NULL or misaligned");
    Ws w;
    carve(t, gb->natm, static_cast<char *>(ws), w);
    EVC_REQUIRE(ws_bytes >= w.bytes * (size_t)gb->count, "evc_phase_loewdin_batch: workspace too small: %zu < %zu",
                ws_bytes, w.bytes * (size_t)gb->count);
    const int64_t n2 = (int64_t)t->n * t->n;
    LoewdinArgs la{};
    la.S = gb->S;
    la.h = gb->hcore;
    la.X = w.X;
    la.U = w.U;
    la.s = w.s;
    la.h1 = w.h1;
    la.sS = n2;
    la.sh = n2;
    la.sws = w.stride;
    la.n = t->n;
    la.warm = (flags & EVC_FLAG_WARM_START) ? 1 : 0;
    la.scratch = w.B1;
    la.sscratch = w.stride;
    return launch_loewdin(la, gb->count, as_stream(stream));
}

extern "C" int evc_phase_hamiltonian_batch(const evc_trdm_set *t, const evc_geometry_batch *gb, int flags,
                                           double *rows_out, int64_t ld_rows_out, void *ws, size_t ws_bytes,
                                           void *stream) {
    Ws w;
    Geo g;
    Out o;
    if (setup_batch("evc_phase_hamiltonian_batch", t, gb, nullptr, false, ws, ws_bytes, w, g, o)) return -1;
    EVC_REQUIRE(t->rows2 == 0 || (rows_out && ld_rows_out >= t->rows2),
                "evc_phase_hamiltonian_batch: rows_out NULL or ld_rows_out=%lld < rows2=%lld", (long long)ld_rows_out,
                (long long)t->rows2);
    w.warm = (flags & EVC_FLAG_WARM_START) != 0;
    w.loewdin_done = (flags & EVC_FLAG_LOEWDIN_DONE) != 0;
    g.eri_s4 = (flags & EVC_FLAG_ERI_S4) ? 1 : 0;
    return phase_hamiltonian(t, g, w, true, as_stream(stream), rows_out, ld_rows_out);
}

extern "C" int evc_phase_solve_batch(const evc_trdm_set *t, const evc_geometry_batch *gb, const double *h2rows_all,
                                     int64_t ld_rows_all, const evc_outputs_batch *ob, int nroots, int flags,
                                     void *ws, size_t ws_bytes, void *stream) {
    Ws w;
    Geo g;
    Out o;
    if (setup_batch("evc_phase_solve_batch", t, gb, ob, false, ws, ws_bytes, w, g, o)) return -1;
    EVC_REQUIRE(ob && ob->energy && ob->coeffs, "batch outputs.energy/coeffs are required");
    EVC_REQUIRE(nroots >= 1 && nroots <= t->ntrain, "nroots=%d out of range 1..%d", nroots, t->ntrain);
    EVC_REQUIRE(h2rows_all && ld_rows_all >= t->rows2_total,
                "evc_phase_solve_batch: h2rows_all NULL or ld_rows_all=%lld < rows2_total=%lld",
                (long long)ld_rows_all, (long long)t->rows2_total);
    replan(t, w, g.count);
    w.warm = (flags & EVC_FLAG_WARM_START) != 0;
    return phase_solve(t, g, h2rows_all, ld_rows_all, o, nroots, w, as_stream(stream));
}

extern "C" int evc_phase_gradient_batch(const evc_trdm_set *t, const evc_geometry_batch *gb,
                                        const evc_outputs_batch *ob, int flags, void *ws, size_t ws_bytes,
                                        void *stream) {
    Ws w;
    Geo g;
    Out o;
    if (setup_batch("evc_phase_gradient_batch", t, gb, ob, true, ws, ws_bytes, w, g, o)) return -1;
    replan(t, w, g.count);
    return phase_gradient(t, g, o, flags, w, as_stream(stream));
}

extern "C" size_t evc_subspace_solve_ws_bytes(int T, int count) {
    if (T <= kSubspaceSmallT || T > kSubspaceMaxT || count < 1) return 0;
    return sizeof(double) * subspace_big_scratch_doubles(T) * (size_t)count;
}

extern "C" int evc_subspace_solve(const double *h1rows, const double *h2rows, const double *S_train, int T,
                                  int layout, int nroots, double e_shift, double *evals, double *evecs,
                                  double *w2, double *w1, double *Hout, void *ws, size_t ws_bytes, void *stream) {
    EVC_REQUIRE(h1rows && h2rows && S_train && evals && evecs, "evc_subspace_solve: null pointer");
    EVC_REQUIRE(T >= 1 && T <= kSubspaceMaxT, "evc_subspace_solve: T=%d out of range 1..%d", T, kSubspaceMaxT);
    EVC_REQUIRE(T <= kSubspaceSmallT || (ws && aligned16(ws) && ws_bytes >= evc_subspace_solve_ws_bytes(T, 1)),
                "evc_subspace_solve: T=%d needs a workspace of evc_subspace_solve_ws_bytes(T, 1) bytes", T);
    EVC_REQUIRE(layout == 6 || layout == 5 || layout == 3 || layout == 2 || layout == EVC_LAYOUT_SYM8,
                "evc_subspace_solve: layout=%d", layout);
    EVC_REQUIRE(nroots >= 1 && nroots <= T, "evc_subspace_solve: nroots=%d out of range", nroots);
    SolveArgs a;
    memset(&a, 0, sizeof(a));
    a.h1part = h1rows;
    a.nsp1 = 1;
    a.alpha1 = 1.0;
    a.h2part = h2rows;
    a.nsp2 = 1;
    a.alpha2 = 1.0;
    a.S = S_train;
    a.T = T;
    a.layout = layout;
    a.nroots = nroots;
    a.e_shift = e_shift;
    a.evals = evals;
    a.evecs = evecs;
    a.w2 = w2;
    a.w1 = w1;
    a.Hout = Hout;
    a.w2_offset = 0;
    a.w2_count = is_pairs(layout) ? (int64_t)T * (T + 1) / 2 : (int64_t)T * T;
    a.scratch = static_cast<double *>(ws);
    return launch_subspace_solve(a, 1, as_stream(stream));
}

extern "C" int evc_subspace_solve_batch(const double *H, const double *S, int64_t s_stride, int T, int count,
                                        int nroots, const double *e_shift, double *evals, double *evecs,
                                        void *ws, size_t ws_bytes, void *stream) {
    EVC_REQUIRE(H && S && evals && evecs, "evc_subspace_solve_batch: null pointer");
    EVC_REQUIRE(T >= 1 && T <= kSubspaceMaxT, "evc_subspace_solve_batch: T=%d out of range 1..%d", T, kSubspaceMaxT);
    EVC_REQUIRE(T <= kSubspaceSmallT || (ws && aligned16(ws) && ws_bytes >= evc_subspace_solve_ws_bytes(T, count)),
                "evc_subspace_solve_batch: T=%d needs a workspace of evc_subspace_solve_ws_bytes(T, count) bytes", T);
    EVC_REQUIRE(count >= 1 && count <= (1 << 24), "evc_subspace_solve_batch: count=%d out of range", count);
    EVC_REQUIRE(nroots >= 1 && nroots <= T, "evc_subspace_solve_batch: nroots=%d out of range", nroots);
    EVC_REQUIRE(s_stride == 0 || s_stride >= (int64_t)T * T, "evc_subspace_solve_batch: s_stride=%lld",
                (long long)s_stride);
    SolveArgs a;
    memset(&a, 0, sizeof(a));
    a.h1part = H;  // the assembled matrix plays the role of the (single) one-body partial
    a.nsp1 = 1;
    a.alpha1 = 1.0;
    a.sh1 = (int64_t)T * T;
    a.h2part = nullptr;
    a.nsp2 = 0;
    a.S = S;
    a.sS = s_stride;
    a.T = T;
    a.layout = EVC_LAYOUT_FULL6;
    a.nroots = nroots;
    a.e_shift_dev = e_shift;
    a.evals = evals;
    a.sev = T;
    a.evecs = evecs;
    a.svec = (int64_t)T * T;
    a.scratch = static_cast<double *>(ws);
    a.sscratch = T > kSubspaceSmallT ? (int64_t)subspace_big_scratch_doubles(T) : 0;
    return launch_subspace_solve(a, count, as_stream(stream));
}

// Workspace of evc_integrals_oao_batch per geometry: X, U, s, h1 (Loewdin outputs) + one N^4 buffer.
// Layout inside one stride (every piece starts on a 16-byte boundary): [X | U | s | h1 | B1 (n^4)].
static int64_t even_up(int64_t x) { return (x + 1) & ~(int64_t)1; }
static int64_t integrals_ws_stride(int n) {
    const int64_t n2 = (int64_t)n * n;
    return 3 * even_up(n2) + even_up(n) + even_up(n2 * n2);
}

extern "C" size_t evc_integrals_oao_ws_bytes(int n, int count) {
    if (n < 1 || n > kMaxOrbitals || count < 1) return 0;
    return sizeof(double) * (size_t)integrals_ws_stride(n) * (size_t)count;
}

extern "C" int evc_integrals_oao_batch(int n, int count, const double *S, const double *hcore, const double *eri,
                                       double *h1, double *h2, double *trafo, void *ws, size_t ws_bytes,
                                       void *stream) {
    EVC_REQUIRE(S && hcore && eri && h1 && h2 && ws, "evc_integrals_oao_batch: null pointer");
    EVC_REQUIRE(n >= 1 && n <= kMaxOrbitals, "evc_integrals_oao_batch: n=%d out of range 1..%d", n, kMaxOrbitals);
    EVC_REQUIRE(count >= 1 && count <= 65535, "evc_integrals_oao_batch: count=%d out of range", count);
    EVC_REQUIRE(aligned16(ws) && ws_bytes >= evc_integrals_oao_ws_bytes(n, count),
                "evc_integrals_oao_batch: workspace misaligned or too small");
    hipStream_t st = as_stream(stream);
    const int64_t n2 = (int64_t)n * n, n4 = n2 * n2, sw = integrals_ws_stride(n);
    double *base = static_cast<double *>(ws);
    double *X = base, *U = X + even_up(n2), *s = U + even_up(n2), *h1w = s + even_up(n), *B1 = h1w + even_up(n2);
    int rc;
    LoewdinArgs la{};
    la.S = S;
    la.h = hcore;
    la.X = X;
    la.U = U;
    la.s = s;
    la.h1 = h1w;
    la.sS = n2;
    la.sh = n2;
    la.sws = sw;
    la.n = n;
    la.scratch = B1;   // (the N^4 buffer is free until the rotation)
    la.sscratch = sw;
    if ((rc = launch_loewdin(la, count, st))) return rc;
    if (use_pair_transform(n)) {
        PairTransformArgs pa;
        memset(&pa, 0, sizeof(pa));
        pa.in = eri;
        pa.sin = n4;
        pa.C = X;
        pa.sC = sw;
        pa.n = n;
        pa.out = B1;
        pa.sout = sw;
        if ((rc = launch_pair_transform(pa, count, st))) return rc;
        pa.in = B1;
        pa.sin = sw;
        pa.out = h2;
        pa.sout = n4;
        if ((rc = launch_pair_transform(pa, count, st))) return rc;
    } else {
        if ((rc = launch_quarter_transform(eri, n4, X, sw, 0, n, B1, sw, count, st))) return rc;
        if ((rc = launch_quarter_transform(B1, sw, X, sw, 0, n, h2, n4, count, st))) return rc;
        if ((rc = launch_quarter_transform(h2, n4, X, sw, 0, n, B1, sw, count, st))) return rc;
        if ((rc = launch_quarter_transform(B1, sw, X, sw, 0, n, h2, n4, count, st))) return rc;
    }
    hipError_t e = hipMemcpy2DAsync(h1, sizeof(double) * n2, h1w, sizeof(double) * sw, sizeof(double) * n2, count,
                                    hipMemcpyDeviceToDevice, st);
    if (e == hipSuccess && trafo)
        e = hipMemcpy2DAsync(trafo, sizeof(double) * n2, X, sizeof(double) * sw, sizeof(double) * n2, count,
                             hipMemcpyDeviceToDevice, st);
    if (e != hipSuccess) {
        set_error("evc_integrals_oao_batch: copy failed: %s", hipGetErrorString(e));
        return (int)e;
    }
    return 0;
}

static void fake_set(evc_trdm_set &t, int n) {
    memset(&t, 0, sizeof(t));
    t.n = n;
    t.ntrain = 1;
    t.layout = EVC_LAYOUT_FULL6;
    t.rows2_total = 1;
    t.cols2 = (int64_t)n * n * n * n;
    t.ld2 = t.cols2 + (t.cols2 & 1);
    t.ld1 = (int64_t)n * n + ((n * n) & 1);
}

extern "C" size_t evc_grad_elec_ws_bytes(int n, int natm) {
    if (n < 1 || n > kMaxOrbitals) return 0;
    evc_trdm_set t;
    fake_set(t, n);
    Ws w;
    carve(&t, natm, nullptr, w);
    return w.bytes;
}

extern "C" int evc_grad_elec_oao(int n, const evc_geometry *g, const double *trafo, const double *one_rdm,
                                 const double *two_rdm, double *grad, void *ws, size_t ws_bytes, void *stream) {
    EVC_REQUIRE(n >= 1 && n <= kMaxOrbitals, "evc_grad_elec_oao: n=%d out of range 1..%d", n, kMaxOrbitals);
    if (check_geometry(g, true)) return -1;
    EVC_REQUIRE(one_rdm && two_rdm && grad && ws && aligned16(ws), "evc_grad_elec_oao: null/misaligned pointer");
    evc_trdm_set t;
    fake_set(t, n);
    Ws w;
    carve(&t, g->natm, static_cast<char *>(ws), w);
    EVC_REQUIRE(ws_bytes >= w.bytes, "evc_grad_elec_oao: workspace too small: %zu < %zu", ws_bytes, w.bytes);
    hipStream_t st = as_stream(stream);
    const Geo geo = geo_single(g);
    int rc;
    LoewdinArgs la{};
    la.S = g->S;
    la.h = g->hcore;
    la.X = w.X;
    la.U = w.U;
    la.s = w.s;
    la.h1 = w.h1;
    la.n = n;
    la.scratch = w.B1;
    if ((rc = launch_loewdin(la, 1, st))) return rc;
    if (trafo) {
        // caller-supplied ao_mo_trafo (gradients_loewdin.py:271-272); its derivative is still the
        // Loewdin response of g->S, exactly as the reference computes it when none is passed (:274-277)
        hipError_t e = hipMemcpyAsync(w.X, trafo, sizeof(double) * (size_t)n * n, hipMemcpyDeviceToDevice, st);
        if (e != hipSuccess) {
            set_error("evc_grad_elec_oao: copy failed: %s", hipGetErrorString(e));
            return (int)e;
        }
    }
    if ((rc = launch_quarter_transform(g->eri, 0, w.X, 0, 0, n, w.B1, 0, 1, st))) return rc;
    if ((rc = launch_quarter_transform(w.B1, 0, w.X, 0, 0, n, w.B2, 0, 1, st))) return rc;
    if ((rc = launch_quarter_transform(w.B2, 0, w.X, 0, 0, n, w.K3, 0, 1, st))) return rc;
    return gradient_from_rdms(n, geo, one_rdm, 0, const_cast<double *>(two_rdm), 0, nullptr, 0, 0, 0, 1.0, false, grad, 0,
                              w, st);
}
