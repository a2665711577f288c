// Error reporting and the uint8 output pack.
#include <stdarg.h>
#include <mutex>
#include "common.h"

static thread_local char g_err[512] = "";

void ccvs_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" const char* ccvs_last_error(void) { return g_err; }
extern "C" int ccvs_abi_version(void) { return 6; }

// Per-stream CU budgets: a handful of (stream, limit) pairs.  Written by the thread that drives the decode stream, read by
// every launching thread (the token worker reads its own stream's entry): one mutex around the table.  A budget of 0
// erases the entry (a destroyed stream's handle may be handed out again; nothing stale is left behind).
static struct { void* stream; int limit; } g_cu_limits[16];
static int g_n_cu_limits = 0;
static std::mutex g_cu_mutex;

extern "C" int ccvs_stream_cu_limit(void* stream, int32_t cu_limit) {
    CCVS_REQUIRE(cu_limit >= 0, "ccvs_stream_cu_limit: negative limit");
    std::lock_guard<std::mutex> lock(g_cu_mutex);
    for (int i = 0; i < g_n_cu_limits; ++i)
        if (g_cu_limits[i].stream == stream) {
            if (cu_limit == 0) g_cu_limits[i] = g_cu_limits[--g_n_cu_limits];
            else g_cu_limits[i].limit = cu_limit;
            return CCVS_OK;
        }
    if (cu_limit == 0) return CCVS_OK;
    CCVS_REQUIRE(g_n_cu_limits < 16, "ccvs_stream_cu_limit: more than 16 streams with a budget");
    g_cu_limits[g_n_cu_limits].stream = stream;
    g_cu_limits[g_n_cu_limits++].limit = cu_limit;
    return CCVS_OK;
}

int ccvs_cu_limit_of(void* stream) {
    std::lock_guard<std::mutex> lock(g_cu_mutex);
    for (int i = 0; i < g_n_cu_limits; ++i)
        if (g_cu_limits[i].stream == stream) return g_cu_limits[i].limit;
    return 0;
}

// save_video_batch (helpers/generator.py:306-309): clamp to [lo,hi], rescale to [0,1],
// x255, truncate to uint8, NCHW -> NHWC.
__global__ __launch_bounds__(256) void pack_u8_kernel(const float* __restrict__ vid, uint8_t* __restrict__ out, long N, int HW, float lo,
                                                      float hi) {
    const long total = N * HW;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const long n = i / HW;
        const int p = (int)(i - n * HW);
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            float v = vid[(n * 3 + c) * HW + p];
            v = fminf(fmaxf(v, lo), hi);
            v = (v - lo) / (hi - lo);
            out[i * 3 + c] = (uint8_t)(v * 255.f);
        }
    }
}

extern "C" int ccvs_pack_u8(const float* vid, uint8_t* out, int64_t N, int32_t H, int32_t W, float lo, float hi, void* stream) {
    CCVS_REQUIRE(vid && out, "ccvs_pack_u8: null pointer");
    CCVS_REQUIRE(N > 0 && H > 0 && W > 0 && hi > lo, "ccvs_pack_u8: bad arguments");
    const long total = (long)N * H * W;
    const unsigned blocks = limited_grid(cdiv64(total, 256) < 1048576 ? cdiv64(total, 256) : 1048576, stream, 8);
    hipLaunchKernelGGL(pack_u8_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, vid, out, (long)N, H * W, lo, hi);
    CCVS_CHECK_LAUNCH("ccvs_pack_u8");
    return CCVS_OK;
}

// save_video_batch with imagenet_norm (helpers/generator.py:303-309): vid *= std; vid += mean; clamp(0, 1); x255; truncate.
// Each step rounds to fp32 on its own, as the reference's separate tensor ops do (no fused multiply-add).
__global__ __launch_bounds__(256) void pack_u8_norm_kernel(const float* __restrict__ vid, uint8_t* __restrict__ out, long N, int HW, float m0,
                                                           float m1, float m2, float a0, float a1, float a2) {
#pragma clang fp contract(off)
    const long total = N * HW;
    const float mul[3] = {m0, m1, m2}, add[3] = {a0, a1, a2};
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const long n = i / HW;
        const int p = (int)(i - n * HW);
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            float v = vid[(n * 3 + c) * HW + p];
            v = v * mul[c];
            v = v + add[c];
            v = fminf(fmaxf(v, 0.f), 1.f);
            out[i * 3 + c] = (uint8_t)(v * 255.f);
        }
    }
}

extern "C" int ccvs_pack_u8_norm(const float* vid, uint8_t* out, int64_t N, int32_t H, int32_t W, const float* std3, const float* mean3,
                                 void* stream) {
    CCVS_REQUIRE(vid && out && std3 && mean3, "ccvs_pack_u8_norm: null pointer");
    CCVS_REQUIRE(N > 0 && H > 0 && W > 0, "ccvs_pack_u8_norm: bad arguments");
    const long total = (long)N * H * W;
    const unsigned blocks = limited_grid(cdiv64(total, 256) < 1048576 ? cdiv64(total, 256) : 1048576, stream, 8);
    hipLaunchKernelGGL(pack_u8_norm_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, vid, out, (long)N, H * W, std3[0], std3[1], std3[2],
                       mean3[0], mean3[1], mean3[2]);
    CCVS_CHECK_LAUNCH("ccvs_pack_u8_norm");
    return CCVS_OK;
}
