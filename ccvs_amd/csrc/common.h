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

// CU budget of a stream (ccvs_stream_cu_limit, misc.hip): 0 = the whole chip
int ccvs_cu_limit_of(void* stream);

// A 3-D grid of workgroups walked by a 1-D launch: workgroup w of the walk stands for block (w % nx, (w / nx) % ny,
// w / (nx ny)) of the grid it replaces.  With a CU budget the launch is `cap` workgroups that stride over the `total`
// blocks (a persistent grid); without one it is `total` workgroups that each take one block.
struct GridWalk {
    long total;
    int nx, ny;
    int tiled;   // the four-pixel gather kernels: t > 0 -- block bx is a tile of 4t x 256/t pixels instead of 1024 consecutive pixels (quad_pixel)
};
#define GRID_WALK_BEGIN(gw, bx, by, bz)                                              \
    for (long w_ = blockIdx.x; w_ < (gw).total; w_ += gridDim.x) {                   \
        const int bx = (int)(w_ % (gw).nx);                                          \
        const long r_ = w_ / (gw).nx;                                                \
        const int by = (int)(r_ % (gw).ny), bz = (int)(r_ / (gw).ny);
#define GRID_WALK_END }

static inline GridWalk grid_walk(long nx, long ny, long nz) {
    GridWalk g;
    g.total = nx * ny * nz; g.nx = (int)nx; g.ny = (int)ny; g.tiled = 0;
    return g;
}
// workgroups to launch for `blocks` blocks of work on `stream`: all of them, or cu_limit x per_cu persistent ones
static inline unsigned limited_grid(long blocks, void* stream, int per_cu) {
    const int lim = ccvs_cu_limit_of(stream);
    const long cap = lim > 0 ? (long)lim * per_cu : blocks;
    const long n = blocks < cap ? blocks : cap;
    return (unsigned)(n < 1 ? 1 : (n > 0x7fffffffL ? 0x7fffffffL : n));
}

// 8 / 16 bytes of floats at a dword-aligned address: one global_load/store_dwordx2 / dwordx4 (gfx950 takes dword-aligned
// wide accesses).  The HBM-bound kernels write 16 bytes per lane wherever a row allows it: with 4-byte stores per lane they
// ran at 0.6 of what a plain copy reaches on this chip (tools/mem_bench.py).
struct __attribute__((packed, aligned(4))) F32Pair { float x, y; };
struct __attribute__((packed, aligned(4))) F32Quad { float v[4]; };

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
