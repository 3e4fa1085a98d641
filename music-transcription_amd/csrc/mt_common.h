// Shared helpers for libmt_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>
#include <atomic>
#include "../../include/mt_hip.h"

namespace mt {

void set_error(const char* fmt, ...);

#define MT_CHECK_HIP(expr)                                                              \
    do {                                                                                \
        hipError_t _e = (expr);                                                         \
        if (_e != hipSuccess) {                                                         \
            mt::set_error("%s:%d: %s -> %s", __FILE__, __LINE__, #expr, hipGetErrorString(_e)); \
            return MT_EHIP;                                                             \
        }                                                                               \
    } while (0)

#define MT_REQUIRE(cond, code, ...)                                                     \
    do {                                                                                \
        if (!(cond)) { mt::set_error(__VA_ARGS__); return (code); }                     \
    } while (0)

#define MT_CHECK_LAUNCH() MT_CHECK_HIP(hipGetLastError())

// hipFuncSetAttribute(MaxDynamicSharedMemorySize) belongs to the DEVICE's copy of the code object: once per (kernel, device), not
// once per process (a process that drives a second GPU would otherwise launch there with > 64 KB of dynamic LDS and no attribute).
// Two host threads may both find the bit clear: the call is idempotent.  `kernel` in parentheses when it has template commas.
#define MT_SET_MAX_LDS(kernel, bytes)                                                                                        \
    do {                                                                                                                    \
        static std::atomic<unsigned long long> done_{0};                                                                    \
        int dev_ = 0;                                                                                                       \
        MT_CHECK_HIP(hipGetDevice(&dev_));                                                                                  \
        const unsigned long long bit_ = 1ull << (dev_ & 63);                                                                \
        if (!(done_.load(std::memory_order_acquire) & bit_)) {                                                              \
            MT_CHECK_HIP(hipFuncSetAttribute((const void*)(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)(bytes))); \
            done_.fetch_or(bit_, std::memory_order_release);                                                                \
        }                                                                                                                   \
    } while (0)

typedef unsigned short bf16_t;   // raw bf16 bits

__device__ __forceinline__ float bf16_to_f32(bf16_t v) { return __uint_as_float(((unsigned)v) << 16); }

// round-to-nearest-even; finite inputs only on every path that uses it
// Inter-layer dropout mask of the training path: counter-based hash of (seed, layer, element index), regenerated wherever
// the mask is needed (forward re-layout, backward dh) instead of being stored.
__device__ __forceinline__ bool dropout_keep(unsigned seed, unsigned layer, unsigned long long idx, float p) {
    unsigned long long z = idx + ((((unsigned long long)seed) << 8) ^ layer) * 0x9E3779B97F4A7C15ull + 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    z ^= z >> 31;
    return (float)(z >> 40) * (1.0f / 16777216.0f) >= p;
}

__device__ __forceinline__ bf16_t f32_to_bf16(float f) {
    unsigned u = __float_as_uint(f);
    u += 0x7FFFu + ((u >> 16) & 1u);
    return (bf16_t)(u >> 16);
}
__device__ __forceinline__ unsigned pack_bf16x2(float lo, float hi) {
    return (unsigned)f32_to_bf16(lo) | ((unsigned)f32_to_bf16(hi) << 16);
}

static inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }
static inline int cdiv(int a, int b) { return (a + b - 1) / b; }

typedef __attribute__((ext_vector_type(8))) short bf16x8;    // MFMA A/B fragment (4 VGPRs)
typedef _Float16 f16_t;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;  // f16 MFMA A/B fragment (4 VGPRs)
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;

// ---- 16-bit GEMM / conv operand type (MT_DT_BF16 | MT_DT_F16, include/mt_hip.h).  Both run on the same MFMA rate;
// f16 carries 11 significand bits against bf16's 8 (inference activations and weights are far inside its range: the
// conversion saturates at +-65504 instead of producing an infinity), bf16 keeps f32's exponent range (training:
// gradients).  Storage is always raw 16-bit words (bf16_t / bf16x8): the type only selects the conversion and the
// MFMA opcode.
template <int DT> __device__ __forceinline__ bf16_t f32_to_h16(float f) {
    if (DT == MT_DT_F16) {
        const f16_t h = (f16_t)__builtin_amdgcn_fmed3f(f, -65504.0f, 65504.0f);     // NaN stays NaN through v_med3 + v_cvt
        return __builtin_bit_cast(unsigned short, h);
    }
    return f32_to_bf16(f);
}
template <int DT> __device__ __forceinline__ float h16_to_f32(bf16_t v) {
    if (DT == MT_DT_F16) return (float)__builtin_bit_cast(f16_t, v);
    return bf16_to_f32(v);
}
template <int DT> __device__ __forceinline__ f32x16 mfma_32x32x16(bf16x8 a, bf16x8 b, f32x16 c) {
    if (DT == MT_DT_F16) return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}
template <int DT> __device__ __forceinline__ f32x4 mfma_16x16x32(bf16x8 a, bf16x8 b, f32x4 c) {
    if (DT == MT_DT_F16) return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
}
#define MT_REQUIRE_DT(dt, who) MT_REQUIRE((dt) == MT_DT_BF16 || (dt) == MT_DT_F16, MT_EINVAL, who ": operand dtype must be MT_DT_BF16 or MT_DT_F16")

// ---- position index -> (b, f, t) without integer division.  x / d for 0 <= x < 2^31 as one multiply-high and a shift: m = floor(2^(31 + l) / d) + 1,
// l = ceil(log2 d) (Granlund-Montgomery; exact on 31 bits).  The elementwise training kernels turned a linear position index into its three
// coordinates with 64-bit divisions -- a few hundred instructions per position in front of a handful of loads.
struct Div3 { unsigned T, F, mT, sT, mF, sF; };
static inline void div_magic(unsigned d, unsigned* m, unsigned* s) {
    if (d <= 1) { *m = 0; *s = 0; return; }
    unsigned l = 0;
    while ((1ull << l) < d) ++l;
    *m = (unsigned)(((1ull << (31 + l)) / d) + 1);
    *s = l - 1;
}
static inline Div3 make_div3(int T, int F) {
    Div3 d{(unsigned)T, (unsigned)F, 0, 0, 0, 0};
    div_magic(d.T, &d.mT, &d.sT);
    div_magic(d.F, &d.mF, &d.sF);
    return d;
}
__device__ __forceinline__ unsigned fast_div(unsigned x, unsigned d, unsigned m, unsigned s) { return d <= 1 ? x : (__umulhi(x, m) >> s); }
// i < 2^31 -> t = i % T, f = (i / T) % F, b = i / (T * F)
__device__ __forceinline__ void div3(unsigned i, const Div3& d, int& t, int& f, int& b) {
    const unsigned q = fast_div(i, d.T, d.mT, d.sT);
    t = (int)(i - q * d.T);
    const unsigned q2 = fast_div(q, d.F, d.mF, d.sF);
    f = (int)(q - q2 * d.F);
    b = (int)q2;
}

}  // namespace mt
