// Shared helpers for the REFID HIP kernels (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>
#include <atomic>
#include "../../include/refid_hip.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

void refid_set_error(const char* fmt, ...);

#define REFID_CHECK(cond, ...)                    \
    do {                                          \
        if (!(cond)) {                            \
            refid_set_error(__VA_ARGS__);         \
            return 1;                             \
        }                                         \
    } while (0)

#define REFID_LAUNCH_CHECK(what)                                                   \
    do {                                                                           \
        hipError_t e__ = hipGetLastError();                                        \
        if (e__ != hipSuccess) {                                                   \
            refid_set_error("%s: launch failed: %s", what, hipGetErrorString(e__)); \
            return 2;                                                              \
        }                                                                          \
    } while (0)

__device__ __forceinline__ float lrelu(float v, float slope) { return v > 0.f ? v : v * slope; }
// four values at once; `on` is workgroup-uniform (slope != 1): a scalar branch skips the 3 VALU per value of an identity
// activation (every input-gradient conv, half of the forward convs) -- the tiles are VALU + MFMA issue bound
__device__ __forceinline__ void lrelu4(f32x4& v, float slope, bool on) {
    if (on) {
#pragma unroll
        for (int k = 0; k < 4; ++k) v[k] = lrelu(v[k], slope);
        asm volatile("" ::: "memory");                      // (keeps the branch: the compiler would turn it into selects)
    }
}

// Deterministic sum over the threads of a wave that share q = lane % LPP (LPP a power of two): fixed xor-shuffle tree;
// the total ends up in every lane.  (Reductions of parameter gradients never use floating-point atomics: a step's
// bits do not depend on arrival order.)
template <int LPP>
__device__ __forceinline__ float refid_wave_rows_sum(float v) {
#pragma unroll
    for (int o = LPP; o < 64; o <<= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// out[r] = sum of the n doubles of row r of `part`, in a fixed order (train.hip): second stage of every scalar reduction
int refid_launch_sum_rows_f64(const double* part, int rows, int n, double* out, hipStream_t st);

static inline int cdiv(int a, int b) { return (a + b - 1) / b; }
static inline int round_up(int a, int b) { return cdiv(a, b) * b; }

// One-time per-DEVICE set-up of a kernel's dynamic-LDS limit (the attribute belongs to the device's copy of the
// code object).  hipFuncSetAttribute is idempotent, so a benign race between host threads costs one extra call.
template <class K>
static inline int refid_lds_attr_once(std::atomic<unsigned long long>& done, K kernel, int bytes, const char* what) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev > 63) { refid_set_error("%s: hipGetDevice failed", what); return 2; }
    if ((done.load(std::memory_order_relaxed) >> dev) & 1ull) return 0;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    if (e != hipSuccess) { refid_set_error("%s: LDS attribute: %s", what, hipGetErrorString(e)); return 2; }
    done.fetch_or(1ull << dev, std::memory_order_relaxed);
    return 0;
}
