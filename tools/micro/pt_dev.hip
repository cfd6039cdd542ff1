// Development harness of the symmetric pair step: pt_pipe_kernel (transform.hip) against ptd_kernel (pair_dma.hip) on
// the same operands -- results compared element by element and against a host reference, both timed with HIP events.
//   build:  hipcc --offload-arch=gfx950 -O2 -std=c++17 -Ievcont_amd/csrc tools/micro/pt_dev.hip -Levcont_amd \
//           -levcont_hip -Wl,-rpath,'$ORIGIN/../evcont_amd' -o build_micro/pt_dev
//   run:    build_micro/pt_dev [n=30] [G=32] [reps=50] [in_pitch: 0 = n(n+1)/2, 1 = pair_ld(n)] [ct=0]
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <vector>

#include "kernels.hpp"
#ifdef PT_DEV_INLINE_KERNEL   // the kernel under development compiled into the harness (variants by -D flags)
#include "pair_dma.hip"
#endif

#define CK(x)                                                                             \
    do {                                                                                  \
        hipError_t e_ = (x);                                                              \
        if (e_ != hipSuccess) {                                                           \
            fprintf(stderr, "%s:%d %s: %s\n", __FILE__, __LINE__, #x, hipGetErrorString(e_)); \
            exit(2);                                                                      \
        }                                                                                 \
    } while (0)

static unsigned long long rs = 88172645463325252ull;
static double rnd() {
    rs ^= rs << 13;
    rs ^= rs >> 7;
    rs ^= rs << 17;
    return (double)(rs >> 11) / 9007199254740992.0 - 0.5;
}

int main(int argc, char **argv) {
    const int n = argc > 1 ? atoi(argv[1]) : 30;
    const int G = argc > 2 ? atoi(argv[2]) : 32;
    const int reps = argc > 3 ? atoi(argv[3]) : 50;
    const int pitch_mode = argc > 4 ? atoi(argv[4]) : 1;
    const int ct = argc > 5 ? atoi(argv[5]) : 0;
    const int np = n * (n + 1) / 2, ldp = evc::pair_ld(n);
    const int in_ld = pitch_mode ? ldp : np;
    const size_t in_old_sz = (size_t)np * np, in_new_sz = (size_t)np * in_ld + 64, out_old_sz = (size_t)np * np + 64,
                 out_new_sz = (size_t)ldp * ldp;
    std::vector<double> h_in_old(G * in_old_sz), h_in_new(G * in_new_sz, 0.0), h_C((size_t)G * n * n);
    for (int g = 0; g < G; ++g)
        for (int e = 0; e < np; ++e)
            for (int u = 0; u < np; ++u) {
                const double v = rnd();
                h_in_old[g * in_old_sz + (size_t)e * np + u] = v;
                h_in_new[g * in_new_sz + (size_t)e * in_ld + u] = v;
            }
    for (auto &v : h_C) v = rnd();
    double *d_in_old, *d_in_new, *d_C, *d_out_old, *d_out_new;
    CK(hipMalloc(&d_in_old, sizeof(double) * h_in_old.size()));
    CK(hipMalloc(&d_in_new, sizeof(double) * h_in_new.size()));
    CK(hipMalloc(&d_C, sizeof(double) * h_C.size()));
    CK(hipMalloc(&d_out_old, sizeof(double) * G * out_old_sz));
    CK(hipMalloc(&d_out_new, sizeof(double) * G * out_new_sz));
    CK(hipMemcpy(d_in_old, h_in_old.data(), sizeof(double) * h_in_old.size(), hipMemcpyHostToDevice));
    CK(hipMemcpy(d_in_new, h_in_new.data(), sizeof(double) * h_in_new.size(), hipMemcpyHostToDevice));
    CK(hipMemcpy(d_C, h_C.data(), sizeof(double) * h_C.size(), hipMemcpyHostToDevice));
    CK(hipMemset(d_out_old, 0, sizeof(double) * G * out_old_sz));
    CK(hipMemset(d_out_new, 0xff, sizeof(double) * G * out_new_sz));   // NaNs: every element that is read must be written

    evc::PairTransformArgs pa;
    memset(&pa, 0, sizeof(pa));
    pa.C = d_C;
    pa.sC = (int64_t)n * n;
    pa.ct = ct;
    pa.n = n;
    pa.lead_sym = pa.in_lower = pa.rs_lower = pa.in_pairs = pa.out_pairs = 1;
    evc::PairTransformArgs po = pa, pn = pa;
    po.in = d_in_old;
    po.sin = (int64_t)in_old_sz;
    po.out = d_out_old;
    po.sout = (int64_t)out_old_sz;
    pn.in = d_in_new;
    pn.sin = (int64_t)in_new_sz;
    pn.in_ld = in_ld;
    pn.out = d_out_new;
    pn.sout = (int64_t)out_new_sz;
    pn.out_ld = ldp;
    pn.tiles_per_wg = getenv("PT_TILES") ? atoi(getenv("PT_TILES")) : 4;
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    auto time_it = [&](auto &&launch, const char *name) {
        for (int r = 0; r < 3; ++r)
            if (launch()) {
                fprintf(stderr, "%s: launch failed: %s\n", name, evc_last_error());
                exit(3);
            }
        CK(hipDeviceSynchronize());
        float best = 1e30f, sum = 0.f;
        for (int rr = 0; rr < 5; ++rr) {
            CK(hipEventRecord(e0, 0));
            for (int r = 0; r < reps; ++r) launch();
            CK(hipEventRecord(e1, 0));
            CK(hipEventSynchronize(e1));
            float ms;
            CK(hipEventElapsedTime(&ms, e0, e1));
            best = ms < best ? ms : best;
            sum += ms;
        }
        printf("%-28s best %.2f us   mean %.2f us per launch\n", name, 1e3 * best / reps, 1e3 * sum / 5 / reps);
    };
    time_it([&] { return evc::launch_pair_transform(po, G, 0); }, "pt_pipe_kernel (old)");
    time_it([&] { return evc::launch_pair_transform_dma(pn, G, 0); }, "ptd_kernel (new)");
    time_it([&] { return evc::launch_pair_transform(po, G, 0); }, "pt_pipe_kernel (old)");
    time_it([&] { return evc::launch_pair_transform_dma(pn, G, 0); }, "ptd_kernel (new)");

    std::vector<double> o_old(G * out_old_sz), o_new(G * out_new_sz);
    CK(hipMemcpy(o_old.data(), d_out_old, sizeof(double) * o_old.size(), hipMemcpyDeviceToHost));
    CK(hipMemcpy(o_new.data(), d_out_new, sizeof(double) * o_new.size(), hipMemcpyDeviceToHost));
    double dmax = 0.0;
    long long nbad = 0;
    for (int g = 0; g < G; ++g)
        for (int u = 0; u < np; ++u)
            for (int e = 0; e < np; ++e) {
                const double a = o_old[g * out_old_sz + (size_t)u * np + e], b = o_new[g * out_new_sz + (size_t)u * ldp + e];
                const double d = fabs(a - b);
                if (!(d <= 1e-13)) ++nbad;
                if (d > dmax || d != d) dmax = d;
            }
    printf("old vs new: max |diff| = %.3e, elements off by more than 1e-13: %lld\n", dmax, nbad);
    // host reference, geometries 0 and G-1: N_e = X^T M_e X (ct: X := C^T)
    double rmax = 0.0;
    for (int g : {0, G - 1}) {
        const double *Cg = h_C.data() + (size_t)g * n * n;
        std::vector<double> M(n * n), H(n * n);
        for (int e = 0; e < np; ++e) {
            const double *row = h_in_old.data() + g * in_old_sz + (size_t)e * np;
            for (int r = 0; r < n; ++r)
                for (int s = 0; s < n; ++s) M[r * n + s] = row[r >= s ? r * (r + 1) / 2 + s : s * (s + 1) / 2 + r];
            auto X = [&](int d, int c) { return ct ? Cg[c * n + d] : Cg[d * n + c]; };
            for (int r = 0; r < n; ++r)
                for (int s2 = 0; s2 < n; ++s2) {
                    double acc = 0.0;
                    for (int s = 0; s < n; ++s) acc += M[r * n + s] * X(s, s2);
                    H[r * n + s2] = acc;
                }
            for (int r2 = 0; r2 < n; ++r2)
                for (int s2 = 0; s2 <= r2; ++s2) {
                    double acc = 0.0;
                    for (int r = 0; r < n; ++r) acc += X(r, r2) * H[r * n + s2];
                    const double got = o_new[g * out_new_sz + (size_t)(r2 * (r2 + 1) / 2 + s2) * ldp + e];
                    const double d = fabs(acc - got);
                    if (d > rmax || d != d) rmax = d;
                }
        }
    }
    printf("new vs host reference (geometries 0, %d): max |diff| = %.3e\n", G - 1, rmax);
    return (nbad == 0 && rmax < 1e-11) ? 0 : 1;
}
