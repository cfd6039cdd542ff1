// Does anything issue beside an FP64 MFMA on gfx950?  One workgroup of 8 waves per CU: waves 0-3 (one per SIMD) run a
// v_mfma_f64_16x16x4_f64 loop, waves 4-7 (their SIMD mates) run another instruction mix.  Each half records its own
// duration (100 MHz wall clock); the mix is run alone, the MFMA loop alone, and both together.  together = max(alone)
// means the two co-issue, together = sum means the SIMD serialises them.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef double d4 __attribute__((ext_vector_type(4)));

typedef float f4 __attribute__((ext_vector_type(4)));
// MIX 6 / 7: a streaming read (16-byte loads, 8 per iteration and lane, scalar base + constant lane offset: no vector
// instruction per load) with 16 loads in flight per lane; 7 also waits for everything once per iteration
template <int MIX>
__device__ __forceinline__ void stream_work(int iters, double *sink, const char *buf) {
    const char *vb = buf + ((size_t)(blockIdx.x * 4 + (threadIdx.x >> 6) - 4) * iters) * 8192;
    const unsigned long long ub = (unsigned long long)vb;
    unsigned long long sb = ((unsigned long long)__builtin_amdgcn_readfirstlane((unsigned)(ub >> 32)) << 32) |
                            (unsigned)__builtin_amdgcn_readfirstlane((unsigned)ub);
    const unsigned voff = (threadIdx.x & 63) * 16;
    f4 r0, r1, r2, r3, r4, r5, r6, r7;
    for (int it = 0; it < iters; ++it) {
        asm volatile("global_load_dwordx4 %0, %8, %9\n\t"
                     "global_load_dwordx4 %1, %8, %9 offset:1024\n\t"
                     "global_load_dwordx4 %2, %8, %9 offset:2048\n\t"
                     "global_load_dwordx4 %3, %8, %9 offset:3072\n\t"
                     "s_waitcnt vmcnt(12)\n\t"
                     "global_load_dwordx4 %4, %8, %10\n\t"
                     "global_load_dwordx4 %5, %8, %10 offset:1024\n\t"
                     "global_load_dwordx4 %6, %8, %10 offset:2048\n\t"
                     "global_load_dwordx4 %7, %8, %10 offset:3072\n\t"
                     "s_waitcnt vmcnt(12)"
                     : "=&v"(r0), "=&v"(r1), "=&v"(r2), "=&v"(r3), "=&v"(r4), "=&v"(r5), "=&v"(r6), "=&v"(r7)
                     : "v"(voff), "s"(sb), "s"(sb + 4096)
                     : "memory");
        if (MIX == 7) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        sb += 8192;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    sink[blockIdx.x * 512 + threadIdx.x] = r0[0] + r1[0] + r2[0] + r3[0] + r4[0] + r5[0] + r6[0] + r7[0];
}

template <int MIX>   // 0 int VALU, 1 f32 FMA, 2 f64 FMA, 3 LDS read, 4 v_mov, 5 SALU
__device__ __forceinline__ void other_work(int iters, double *sink, double *lds) {
    unsigned x = threadIdx.x, y = 12345u;
    float ff = threadIdx.x * 1e-3f;
    double fd = threadIdx.x * 1e-3;
    double acc = 0;
    int s = 0;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 64; ++r) {
            if (MIX == 0) { x = x * 3u + y; y ^= x; }
            if (MIX == 1) ff = __builtin_fmaf(ff, 0.999f, 1e-3f);
            if (MIX == 2) fd = __builtin_fma(fd, 0.999, 1e-3);
            if (MIX == 3) acc += lds[(threadIdx.x + r * 67 + it) & 2047];
            if (MIX == 4) asm volatile("v_mov_b32 %0, %0" : "+v"(x));
            if (MIX == 5) asm volatile("s_add_u32 %0, %0, 1" : "+s"(s));
        }
    }
    sink[blockIdx.x * 512 + threadIdx.x] = x + y + ff + fd + acc + s;
}

template <int MIX>
__global__ __launch_bounds__(512) void k(double *out, long long *clk, int mf_iters, int ot_iters, double a0,
                                         const char *buf, int prio) {
    __shared__ double lds[2048];
    for (int i = threadIdx.x; i < 2048; i += 512) lds[i] = i;
    __syncthreads();
    const int wave = threadIdx.x >> 6;
    const long long t0 = wall_clock64();
    if (wave < 4) {
        d4 acc[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[i] = (d4){0, 0, 0, 0};
        const double a = a0 + threadIdx.x * 1e-9, b = 0.999999;
        for (int it = 0; it < mf_iters; ++it) {
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
        }
        out[blockIdx.x * 512 + threadIdx.x] = acc[0][0] + acc[1][1] + acc[2][2] + acc[3][3];
    } else {
        if (prio) __builtin_amdgcn_s_setprio(3);
        if constexpr (MIX >= 6) stream_work<MIX>(ot_iters, out, buf);
        else other_work<MIX>(ot_iters, out, lds);
    }
    const long long t1 = wall_clock64();
    if ((threadIdx.x & 63) == 0) clk[blockIdx.x * 8 + wave] = t1 - t0;
}

template <int MIX>
static void run(const char *name, int ot_iters, int prio = 0) {
    const int blocks = 256, mf_iters = 2000;   // 8000 MFMAs per wave = 512K cycles
    double *out;
    long long *clk, h[256 * 8];
    static char *buf = nullptr;
    if (!buf) {
        (void)hipMalloc(&buf, (size_t)1024 * 300 * 8192 + 8192);   // 2.5 GB: every streaming wave reads its own range
        (void)hipMemset(buf, 0, (size_t)1024 * 300 * 8192 + 8192);
    }
    (void)hipMalloc(&out, sizeof(double) * blocks * 512);
    (void)hipMalloc(&clk, sizeof(long long) * blocks * 8);
    double res[3][2];
    for (int cfg = 0; cfg < 3; ++cfg) {   // 0: MFMA alone, 1: mix alone, 2: together
        const int mi = cfg == 1 ? 0 : mf_iters, oi = cfg == 0 ? 0 : ot_iters;
        for (int rep = 0; rep < 2; ++rep) {
            hipLaunchKernelGGL(k<MIX>, dim3(blocks), dim3(512), 0, 0, out, clk, mi, oi, 1.0, buf, prio);
            (void)hipDeviceSynchronize();
        }
        (void)hipMemcpy(h, clk, sizeof(h), hipMemcpyDeviceToHost);
        double m = 0, o = 0;
        for (int b = 0; b < blocks; ++b)
            for (int w = 0; w < 8; ++w) (w < 4 ? m : o) += h[b * 8 + w];
        res[cfg][0] = m / (blocks * 4) * 0.01;   // us
        res[cfg][1] = o / (blocks * 4) * 0.01;
    }
    printf("%-12s MFMA alone %7.1f us | mix alone %7.1f us | together: MFMA waves %7.1f us, mix waves %7.1f us  (sum %7.1f)\n",
           name, res[0][0], res[1][1], res[2][0], res[2][1], res[0][0] + res[1][1]);
    (void)hipFree(out);
    (void)hipFree(clk);
}

int main() {
    run<0>("int VALU", 500);
    run<1>("f32 FMA", 1000);
    run<2>("f64 FMA", 500);
    run<3>("LDS read", 300);
    run<4>("v_mov", 1000);
    run<5>("SALU", 1000);
    run<6>("stream 16", 300);   // 300 x 8 KB per wave: 2.5 GB in all
    run<7>("stream 8", 300);
    // the same with s_setprio 3 in the second wave of every SIMD
    run<0>("int VALU p3", 500, 1);
    run<2>("f64 FMA p3", 500, 1);
    run<5>("SALU p3", 1000, 1);
    run<6>("stream16 p3", 300, 1);
    return 0;
}
