// FIR resampling (Blur) and the learned depthwise x2 upsampling.  HBM-bound stencils.
//
// ccvs_upfirdn2d  <- upfirdn2d(input, kernel, up, down, pad)  modules/upfirdn2d.py:145-203,
//                    upfirdn2d_kernel.cu:107-207, for kernel outer([1,3,3,1])/64*gain.
// ccvs_dwconvT4x4s2 <- nn.ConvTranspose2d(C,C,4,stride=2,padding=1,groups=C,bias=False)
//                    skip_autoencoder.py:153-154,168.
#include "common.h"

struct FirK {
    const float* x;
    float* y;
    const float* res;
    long NC;
    int H, W, Ho, Wo, up, down, pad0;
    float gain;
    int act;
    float out_scale;
};

// General form: one thread per output pixel, 16 taps on the zero-inserted grid.
__global__ __launch_bounds__(256) void upfirdn2d_generic_kernel(FirK p) {
    const long total = p.NC * p.Ho * p.Wo;
    const float t4[4] = {1.f, 3.f, 3.f, 1.f};
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int ox = (int)(i % p.Wo);
        const long t = i / p.Wo;
        const int oy = (int)(t % p.Ho);
        const long nc = t / p.Ho;
        const float* xp = p.x + nc * (long)p.H * p.W;
        float acc = 0.f;
#pragma unroll
        for (int ky = 0; ky < 4; ++ky) {
            const int u = oy * p.down + ky - p.pad0;  // row on the zero-inserted grid
            if (u < 0 || u >= p.H * p.up || (u % p.up) != 0) continue;
            const int iy = u / p.up;
            float row = 0.f;
#pragma unroll
            for (int kx = 0; kx < 4; ++kx) {
                const int v = ox * p.down + kx - p.pad0;
                if (v < 0 || v >= p.W * p.up || (v % p.up) != 0) continue;
                row += t4[kx] * xp[(long)iy * p.W + v / p.up];
            }
            acc += t4[ky] * row;
        }
        float v = acc * (p.gain * (1.f / 64.f));
        if (p.act == CCVS_ACT_LRELU) v = lrelu01(v);
        if (p.res) v += p.res[i];
        p.y[i] = v * p.out_scale;
    }
}

// up = down = 1 fast path: a thread produces 4 consecutive outputs of one row from a
// 4 x 7 input window held in registers (7 loads per output row instead of 16 per pixel),
// separable 1-3-3-1 passes.
__global__ __launch_bounds__(256) void blur4x4_kernel(FirK p) {
    const int Wq = (p.Wo + 3) >> 2;
    const long total = p.NC * p.Ho * Wq;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int q = (int)(i % Wq);
        const long t = i / Wq;
        const int oy = (int)(t % p.Ho);
        const long nc = t / p.Ho;
        const int ox0 = q * 4;
        const float* xp = p.x + nc * (long)p.H * p.W;
        float h[4][4];  // horizontal pass results for the 4 rows x 4 outputs
#pragma unroll
        for (int ky = 0; ky < 4; ++ky) {
            const int iy = oy + ky - p.pad0;
            float r[7];
            const bool rowok = (iy >= 0 && iy < p.H);
            const long rowoff = (long)min(max(iy, 0), p.H - 1) * p.W;
#pragma unroll
            for (int c = 0; c < 7; ++c) {  // unconditional loads from clamped addresses, zeroed afterwards
                const int ix = ox0 + c - p.pad0;
                const float t = xp[rowoff + min(max(ix, 0), p.W - 1)];
                r[c] = (rowok && ix >= 0 && ix < p.W) ? t : 0.f;
            }
#pragma unroll
            for (int o = 0; o < 4; ++o) h[ky][o] = (r[o] + r[o + 3]) + 3.f * (r[o + 1] + r[o + 2]);
        }
        const float g = p.gain * (1.f / 64.f);
#pragma unroll
        for (int o = 0; o < 4; ++o) {
            const int ox = ox0 + o;
            if (ox < p.Wo) {
                float v = ((h[0][o] + h[3][o]) + 3.f * (h[1][o] + h[2][o])) * g;
                if (p.act == CCVS_ACT_LRELU) v = lrelu01(v);
                const long oi = (nc * p.Ho + oy) * (long)p.Wo + ox;
                if (p.res) v += p.res[oi];
                p.y[oi] = v * p.out_scale;
            }
        }
    }
}

extern "C" int ccvs_upfirdn2d(const float* x, float* y, const float* residual, int64_t NC, int32_t H, int32_t W, int32_t up,
                              int32_t down, int32_t pad0, int32_t pad1, float gain, int32_t act, float out_scale, void* stream) {
    CCVS_REQUIRE(x && y, "ccvs_upfirdn2d: null pointer");
    CCVS_REQUIRE(NC > 0 && H > 0 && W > 0, "ccvs_upfirdn2d: empty tensor");
    CCVS_REQUIRE((up == 1 || up == 2) && (down == 1 || down == 2), "ccvs_upfirdn2d: up=%d down=%d unsupported", up, down);
    FirK k;
    k.x = x; k.y = y; k.res = residual; k.NC = NC; k.H = H; k.W = W; k.up = up; k.down = down; k.pad0 = pad0;
    k.Ho = (H * up + pad0 + pad1 - 4) / down + 1;
    k.Wo = (W * up + pad0 + pad1 - 4) / down + 1;
    CCVS_REQUIRE(k.Ho > 0 && k.Wo > 0, "ccvs_upfirdn2d: empty output");
    k.gain = gain; k.act = act; k.out_scale = out_scale;
    hipStream_t st = (hipStream_t)stream;
    if (up == 1 && down == 1) {
        const long work = NC * k.Ho * ((k.Wo + 3) / 4);
        const int blocks = (int)(cdiv64(work, 256) < 65536 * 16 ? cdiv64(work, 256) : 65536 * 16);
        hipLaunchKernelGGL(blur4x4_kernel, dim3(blocks), dim3(256), 0, st, k);
    } else {
        const long work = NC * k.Ho * k.Wo;
        const int blocks = (int)(cdiv64(work, 256) < 65536 * 16 ? cdiv64(work, 256) : 65536 * 16);
        hipLaunchKernelGGL(upfirdn2d_generic_kernel, dim3(blocks), dim3(256), 0, st, k);
    }
    CCVS_CHECK_LAUNCH("ccvs_upfirdn2d");
    return CCVS_OK;
}

// out[c][y][x] = sum over the (<=2 x 2) taps with y + 1 = 2*iy + ky, x + 1 = 2*ix + kx.
__global__ __launch_bounds__(256) void dwconvT4x4s2_kernel(const float* __restrict__ x, long x_sN, const float* __restrict__ w,
                                                           float* __restrict__ y, long y_sN, long N, int C, int H, int W) {
    const int Ho = 2 * H, Wo = 2 * W;
    const long total = N * C * (long)Ho * Wo;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int ox = (int)(i % Wo);
        const long t = i / Wo;
        const int oy = (int)(t % Ho);
        const long nc = t / Ho;
        const int c = (int)(nc % C);
        const long n = nc / C;
        const float* xp = x + n * x_sN + (long)c * H * W;
        const float* wp = w + c * 16;
        float acc = 0.f;
        const int ky0 = (oy + 1) & 1, kx0 = (ox + 1) & 1;
#pragma unroll
        for (int a = 0; a < 2; ++a) {
            const int ky = ky0 + 2 * a;
            const int iy = (oy + 1 - ky) >> 1;
            if (iy < 0 || iy >= H) continue;
#pragma unroll
            for (int b = 0; b < 2; ++b) {
                const int kx = kx0 + 2 * b;
                const int ix = (ox + 1 - kx) >> 1;
                if (ix < 0 || ix >= W) continue;
                acc += xp[(long)iy * W + ix] * wp[ky * 4 + kx];
            }
        }
        y[n * y_sN + ((long)c * Ho + oy) * Wo + ox] = acc;
    }
}

extern "C" int ccvs_dwconvT4x4s2(const float* x, int64_t x_sN, const float* w, float* y, int64_t y_sN, int32_t N, int32_t C, int32_t H,
                                 int32_t W, void* stream) {
    CCVS_REQUIRE(x && w && y, "ccvs_dwconvT4x4s2: null pointer");
    CCVS_REQUIRE(N > 0 && C > 0 && H > 0 && W > 0, "ccvs_dwconvT4x4s2: empty tensor");
    const long work = (long)N * C * 4 * H * W;
    const int blocks = (int)(cdiv64(work, 256) < 65536 * 16 ? cdiv64(work, 256) : 65536 * 16);
    hipLaunchKernelGGL(dwconvT4x4s2_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, x, (long)x_sN, w, y, (long)y_sN, (long)N, C, H, W);
    CCVS_CHECK_LAUNCH("ccvs_dwconvT4x4s2");
    return CCVS_OK;
}
