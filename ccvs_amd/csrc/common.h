// Shared helpers for the gfx950 kernels of libccvs_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include "ccvs_hip.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

#define WAVE 64

void ccvs_set_error(const char* fmt, ...);

#define CCVS_REQUIRE(cond, ...)          \
    do {                                 \
        if (!(cond)) {                   \
            ccvs_set_error(__VA_ARGS__); \
            return CCVS_ERR_ARG;         \
        }                                \
    } while (0)

#define CCVS_CHECK_LAUNCH(name)                                                   \
    do {                                                                          \
        hipError_t e_ = hipGetLastError();                                        \
        if (e_ != hipSuccess) {                                                   \
            ccvs_set_error("%s: launch failed: %s", name, hipGetErrorString(e_)); \
            return CCVS_ERR_LAUNCH;                                               \
        }                                                                         \
    } while (0)

static inline int cdiv(int a, int b) { return (a + b - 1) / b; }
static inline int64_t cdiv64(int64_t a, int64_t b) { return (a + b - 1) / b; }

__device__ __forceinline__ float lrelu01(float v) { return v > 0.f ? v : 0.1f * v; }

__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
