// Shared device helpers and host-side error plumbing for libevcont_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>

#include <atomic>

#include "../../include/evcont_hip.h"

namespace evc {

constexpr int kWave = 64;  // CDNA wavefront

// ---- host side -------------------------------------------------------------------
void set_error(const char *fmt, ...);
// Name of the kernel a launcher just enqueued for a stage (EVC_PROF_* of include/evcont_hip.h): what a measurement of
// that stage names as the kernel it timed (evc_profile_kernel).  printf-style; the last launch of a stage wins.
void note_kernel(int stage, const char *fmt, ...);
inline hipStream_t as_stream(void *s) { return reinterpret_cast<hipStream_t>(s); }

#define EVC_REQUIRE(cond, ...)            \
    do {                                  \
        if (!(cond)) {                    \
            evc::set_error(__VA_ARGS__);  \
            return -1;                    \
        }                                 \
    } while (0)

#define EVC_HIP(call)                                                               \
    do {                                                                            \
        hipError_t e_ = (call);                                                     \
        if (e_ != hipSuccess) {                                                     \
            evc::set_error("%s: %s", #call, hipGetErrorString(e_));                 \
            return (int)e_;                                                         \
        }                                                                           \
    } while (0)

#define EVC_LAUNCH_CHECK(name)                                                      \
    do {                                                                            \
        hipError_t e_ = hipGetLastError();                                          \
        if (e_ != hipSuccess) {                                                     \
            evc::set_error("%s: launch failed: %s", name, hipGetErrorString(e_));   \
            return (int)e_;                                                         \
        }                                                                           \
    } while (0)

// hipFuncAttributeMaxDynamicSharedMemorySize is a PER-DEVICE attribute of a kernel: one process may drive several
// devices through this library, so the "already raised" flag is kept per device (bit d of `done`; devices >= 64 set
// the attribute on every call).  Returns 0 or the hipError_t of the failed call (error string set).
struct LdsAttr {
    std::atomic<uint64_t> done{0};
};
template <typename K>
inline int allow_dynamic_lds(K kernel, LdsAttr &st, int bytes, const char *name) {
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e == hipSuccess) {
        const bool tracked = dev >= 0 && dev < 64;
        if (tracked && (st.done.load(std::memory_order_acquire) >> dev & 1u)) return 0;
        e = hipFuncSetAttribute(reinterpret_cast<const void *>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
        if (e == hipSuccess) {
            if (tracked) st.done.fetch_or((uint64_t)1 << dev, std::memory_order_release);
            return 0;
        }
    }
    set_error("%s: cannot raise the dynamic LDS limit to %d bytes: %s", name, bytes, hipGetErrorString(e));
    return (int)e;
}

inline bool aligned16(const void *p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }
inline int64_t ceil_div(int64_t a, int64_t b) { return (a + b - 1) / b; }
inline size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

// ---- device side -----------------------------------------------------------------
// Workgroup barrier that orders LDS traffic only.  __syncthreads() carries a workgroup-scope fence
// and therefore waits for vmcnt(0): every global load AND store of the wave must have completed
// before it reaches the barrier.  Where the barrier only hands an LDS buffer from one set of lanes to
// another, waiting for the wave's own LDS operations is enough, and the global stores (or prefetches)
// issued before it keep draining behind the work that follows.
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// Cross-lane moves on the DPP path (one v_mov_b32_dpp per 32-bit half).  __shfl_xor lowers to ds_bpermute_b32: an
// LDS-pipeline round trip per half and level, ~10x the latency.  CTRL: quad_perm [1,0,3,2] = 0xB1 (xor 1),
// [2,3,0,1] = 0x4E (xor 2), row_half_mirror = 0x141, row_mirror = 0x140.
template <int CTRL>
__device__ __forceinline__ double dpp_move(double v) {
    const unsigned long long b = __builtin_bit_cast(unsigned long long, v);
    const int lo = __builtin_amdgcn_update_dpp(0, (int)(unsigned)b, CTRL, 0xF, 0xF, true);
    const int hi = __builtin_amdgcn_update_dpp(0, (int)(unsigned)(b >> 32), CTRL, 0xF, 0xF, true);
    return __builtin_bit_cast(double, ((unsigned long long)(unsigned)hi << 32) | (unsigned)lo);
}
template <int CTRL>
__device__ __forceinline__ float dpp_move(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, true));
}
__device__ __forceinline__ double readlane_f64(double v, int l) {
    const unsigned long long b = __builtin_bit_cast(unsigned long long, v);
    const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)b, l);
    const unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(b >> 32), l);
    return __builtin_bit_cast(double, ((unsigned long long)hi << 32) | lo);
}

// Sum over the 64 lanes of a wave; every lane gets the total.  Fixed order (quads, half rows, rows on DPP moves, then
// the four row sums through readlane), so results are reproducible run to run.
__device__ __forceinline__ double wave_sum(double v) {
    v += dpp_move<0xB1>(v);
    v += dpp_move<0x4E>(v);
    v += dpp_move<0x141>(v);
    v += dpp_move<0x140>(v);
    return (readlane_f64(v, 0) + readlane_f64(v, 16)) + (readlane_f64(v, 32) + readlane_f64(v, 48));
}

// Sum over a workgroup of NW waves; result valid in thread 0 (and all threads of wave 0).
// `scratch` must hold NW doubles of LDS per concurrently reduced value.
template <int NW>
__device__ __forceinline__ double block_sum(double v, double *scratch) {
    v = wave_sum(v);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) scratch[wave] = v;
    __syncthreads();
    double t = 0.0;
#pragma unroll
    for (int w = 0; w < NW; ++w) t += scratch[w];
    __syncthreads();
    return t;
}

// Index of element (R,C), R>=C, in the row-major lower triangle (np.tril_indices order).
__device__ __host__ __forceinline__ int64_t tri_index(int64_t R, int64_t C) { return R * (R + 1) / 2 + C; }

// Inverse: packed index m -> row R (largest R with R(R+1)/2 <= m).
__device__ __forceinline__ int64_t tri_row(int64_t m) {
    int64_t R = (int64_t)((sqrt(8.0 * (double)m + 1.0) - 1.0) * 0.5);
    while (R * (R + 1) / 2 > m) --R;
    while ((R + 1) * (R + 2) / 2 <= m) ++R;
    return R;
}

// The same for small indices (m < 2^20) in 32-bit / single-precision arithmetic (a handful of instructions).
__device__ __forceinline__ int tri_row_small(int m) {
    int R = (int)((__builtin_sqrtf(8.0f * (float)m + 1.0f) - 1.0f) * 0.5f);
    if (R * (R + 1) / 2 > m) --R;
    if ((R + 1) * (R + 2) / 2 <= m) ++R;
    return R;
}

}  // namespace evc
