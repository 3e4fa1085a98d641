// Shared helpers for libmt_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>
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

}  // namespace mt
