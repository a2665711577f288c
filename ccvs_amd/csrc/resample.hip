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

// up = down = 1: a 256-thread workgroup owns a 32 x 64 output tile of one channel plane (two rows of four outputs per thread;
// with 16 rows the loads of a workgroup were too few to cover their latency: 2.9 TB/s); the
// (32+3) x (64+3) input tile is staged once in LDS with coalesced loads (every input element is read
// from HBM once, unconditionally from a clamped address), then each thread produces 4 consecutive
// outputs of one row with the separable 1-3-3-1 passes and writes them as one 16-byte store.
#define BLUR_TH 32
#define BLUR_TW 64
__global__ __launch_bounds__(256) void blur4x4_tile_kernel(FirK p, int tiles_x, GridWalk gw) {
    __shared__ __attribute__((aligned(16))) float tile[(BLUR_TH + 3) * (BLUR_TW + 4)];
    constexpr int IWp = BLUR_TW + 4, IH = BLUR_TH + 3, IW = BLUR_TW + 3;
    const int tid = threadIdx.x;
    GRID_WALK_BEGIN(gw, bx, by, bz)
    (void)bz;
    const int ty = bx / tiles_x, tx = bx - ty * tiles_x;
    const long nc = by;
    if (w_ != (long)blockIdx.x) __syncthreads();   // persistent walk: the tile of the previous block has been read by everyone
    const float* xp = p.x + nc * (long)p.H * p.W;
    const int iy0 = ty * BLUR_TH - p.pad0, ix0 = tx * BLUR_TW - p.pad0;
    // 16 bytes per lane where the four columns lie inside the row (dword-aligned is enough for the load on gfx950; the LDS
    // slot is 16-byte aligned: the tile starts at ix0); clamped scalar loads with zero fill at the image border
    for (int e = tid; e < IH * (IWp / 4); e += 256) {
        const int r = e / (IWp / 4), c = (e - r * (IWp / 4)) * 4;
        const int iy = iy0 + r, ix = ix0 + c;
        f32x4 t4;
        if (iy >= 0 && iy < p.H && ix >= 0 && ix + 3 < p.W) {
            const F32Quad q = *reinterpret_cast<const F32Quad*>(xp + (long)iy * p.W + ix);
            t4 = f32x4{q.v[0], q.v[1], q.v[2], q.v[3]};
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float t = xp[(long)min(max(iy, 0), p.H - 1) * p.W + min(max(ix + j, 0), p.W - 1)];
                t4[j] = (iy >= 0 && iy < p.H && ix + j >= 0 && ix + j < p.W) ? t : 0.f;
            }
        }
        *reinterpret_cast<f32x4*>(&tile[r * IWp + c]) = t4;
    }
    __syncthreads();
#pragma unroll
    for (int rr = 0; rr < BLUR_TH / 16; ++rr) {
    const int row = (tid >> 4) + 16 * rr, col = (tid & 15) * 4;
    const int oy = ty * BLUR_TH + row, ox0 = tx * BLUR_TW + col;
    if (oy < p.Ho && ox0 < p.Wo) {
    float h[4][4];
#pragma unroll
    for (int ky = 0; ky < 4; ++ky) {
        // the 7 values of the row window as two 16-byte LDS reads (col and the row stride are multiples of 4 floats): with 28
        // dword reads per thread, lanes 16 bytes apart on rows 272 bytes apart, 58 % of the kernel's LDS cycles were bank
        // conflicts (SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE, tools/pmc_decoder_kernels.sh)
        const f32x4 ra = *reinterpret_cast<const f32x4*>(&tile[(row + ky) * IWp + col]);
        const f32x4 rb = *reinterpret_cast<const f32x4*>(&tile[(row + ky) * IWp + col + 4]);
        const float r[7] = {ra[0], ra[1], ra[2], ra[3], rb[0], rb[1], rb[2]};
#pragma unroll
        for (int o = 0; o < 4; ++o) h[ky][o] = (r[o] + r[o + 3]) + 3.f * (r[o + 1] + r[o + 2]);
    }
    const float g = p.gain * (1.f / 64.f);
    float v[4];
#pragma unroll
    for (int o = 0; o < 4; ++o) {
        v[o] = ((h[0][o] + h[3][o]) + 3.f * (h[1][o] + h[2][o])) * g;
        if (p.act == CCVS_ACT_LRELU) v[o] = lrelu01(v[o]);
    }
    const long oi = (nc * p.Ho + oy) * (long)p.Wo + ox0;
    const int nv = min(4, p.Wo - ox0);
    if (nv == 4) {   // 16-byte store whatever the row's alignment (Wo = 254: every other row starts 8 bytes off)
        if (p.res) {
            const F32Quad r4 = *reinterpret_cast<const F32Quad*>(p.res + oi);
#pragma unroll
            for (int o = 0; o < 4; ++o) v[o] += r4.v[o];
        }
        F32Quad o4;
#pragma unroll
        for (int o = 0; o < 4; ++o) o4.v[o] = v[o] * p.out_scale;
        *reinterpret_cast<F32Quad*>(p.y + oi) = o4;
    } else {
        for (int o = 0; o < nv; ++o) p.y[oi + o] = (v[o] + (p.res ? p.res[oi + o] : 0.f)) * p.out_scale;
    }
    }
    }
    GRID_WALK_END
}

// up = 2, down = 1, pad (2,1): the decoder's skip up-sampling.  On the zero-inserted grid only every
// other tap hits a sample, so an output quad (2i..2i+1, 2j..2j+1) is a 1-3 / 3-1 blend of the 3 x 3
// input neighbourhood: even rows 1*x[i-1] + 3*x[i], odd rows 3*x[i] + 1*x[i+1] (same along x).
__global__ __launch_bounds__(256) void upsample2_kernel(FirK p) {
    const long total = p.NC * p.H * p.W;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int jx = (int)(i % p.W);
        const long t = i / p.W;
        const int iy = (int)(t % p.H);
        const long nc = t / p.H;
        const float* xp = p.x + nc * (long)p.H * p.W;
        float v[3][3];
#pragma unroll
        for (int a = 0; a < 3; ++a)
#pragma unroll
            for (int b = 0; b < 3; ++b) {
                const int yy = iy + a - 1, xx = jx + b - 1;
                const float tv = xp[(long)min(max(yy, 0), p.H - 1) * p.W + min(max(xx, 0), p.W - 1)];
                v[a][b] = (yy >= 0 && yy < p.H && xx >= 0 && xx < p.W) ? tv : 0.f;
            }
        float hr[3][2];  // horizontal pass: even column 1*left + 3*mid, odd column 3*mid + 1*right
#pragma unroll
        for (int a = 0; a < 3; ++a) { hr[a][0] = v[a][0] + 3.f * v[a][1]; hr[a][1] = 3.f * v[a][1] + v[a][2]; }
        const float g = p.gain * (1.f / 64.f);
#pragma unroll
        for (int a = 0; a < 2; ++a) {
            float o2[2];
#pragma unroll
            for (int b = 0; b < 2; ++b) {
                float r = (a == 0 ? hr[0][b] + 3.f * hr[1][b] : 3.f * hr[1][b] + hr[2][b]) * g;
                if (p.act == CCVS_ACT_LRELU) r = lrelu01(r);
                o2[b] = r;
            }
            const long oi = (nc * p.Ho + 2 * iy + a) * (long)p.Wo + 2 * jx;
            if (p.res) { o2[0] += p.res[oi]; o2[1] += p.res[oi + 1]; }
            *reinterpret_cast<float2*>(p.y + oi) = make_float2(o2[0] * p.out_scale, o2[1] * p.out_scale);
        }
    }
}

// The same for TWO input pixels (jx, jx + 1) per thread (W even): the 3 x 4 neighbourhood gives a 2 x 4 output block, written
// as two 16-byte stores.  Per output the arithmetic is that of upsample2_kernel.
__global__ __launch_bounds__(256) void upsample2x2_kernel(FirK p) {
    const int W2 = p.W >> 1;
    const long total = p.NC * p.H * W2;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int jx = (int)(i % W2) * 2;
        const long t = i / W2;
        const int iy = (int)(t % p.H);
        const long nc = t / p.H;
        const float* xp = p.x + nc * (long)p.H * p.W;
        float v[3][4];
#pragma unroll
        for (int a = 0; a < 3; ++a)
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                const int yy = iy + a - 1, xx = jx + b - 1;
                const float tv = xp[(long)min(max(yy, 0), p.H - 1) * p.W + min(max(xx, 0), p.W - 1)];
                v[a][b] = (yy >= 0 && yy < p.H && xx >= 0 && xx < p.W) ? tv : 0.f;
            }
        float hr[3][4];
#pragma unroll
        for (int a = 0; a < 3; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b) { hr[a][2 * b] = v[a][b] + 3.f * v[a][b + 1]; hr[a][2 * b + 1] = 3.f * v[a][b + 1] + v[a][b + 2]; }
        const float g = p.gain * (1.f / 64.f);
#pragma unroll
        for (int a = 0; a < 2; ++a) {
            const long oi = (nc * p.Ho + 2 * iy + a) * (long)p.Wo + 2 * jx;
            F32Quad o4;
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                float r = (a == 0 ? hr[0][b] + 3.f * hr[1][b] : 3.f * hr[1][b] + hr[2][b]) * g;
                if (p.act == CCVS_ACT_LRELU) r = lrelu01(r);
                o4.v[b] = r;
            }
            if (p.res) {
                const F32Quad r4 = *reinterpret_cast<const F32Quad*>(p.res + oi);
#pragma unroll
                for (int b = 0; b < 4; ++b) o4.v[b] += r4.v[b];
            }
#pragma unroll
            for (int b = 0; b < 4; ++b) o4.v[b] *= p.out_scale;
            *reinterpret_cast<F32Quad*>(p.y + oi) = o4;
        }
    }
}

// up = 1, down = 2: the encoder's blur + decimate.  A thread owns 4 consecutive outputs of a row: their 4 x 10 input window
// comes in as three 16-byte loads per row where it lies inside the row, and they leave as one 16-byte store.  Per output
// the sums run in the order of upfirdn2d_generic_kernel (taps outside the image contribute zeros instead of being skipped).
__global__ __launch_bounds__(256) void down2_kernel(FirK p) {
    const int Wq = (p.Wo + 3) >> 2;
    const long total = p.NC * p.Ho * Wq;
    const float t4[4] = {1.f, 3.f, 3.f, 1.f};
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int ox0 = (int)(i % Wq) * 4;
        const long t = i / Wq;
        const int oy = (int)(t % p.Ho);
        const long nc = t / p.Ho;
        const float* xp = p.x + nc * (long)p.H * p.W;
        const int v0 = ox0 * 2 - p.pad0;            // first input column of the window
        const bool inside = v0 >= 0 && v0 + 12 <= p.W;
        float acc[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ky = 0; ky < 4; ++ky) {
            const int iy = oy * 2 + ky - p.pad0;
            if (iy < 0 || iy >= p.H) continue;     // (whole row outside: adds nothing in the generic kernel either)
            const float* rp = xp + (long)iy * p.W;
            float w[12];
            if (inside) {
#pragma unroll
                for (int q = 0; q < 3; ++q) {
                    const F32Quad r4 = *reinterpret_cast<const F32Quad*>(rp + v0 + 4 * q);
#pragma unroll
                    for (int e = 0; e < 4; ++e) w[4 * q + e] = r4.v[e];
                }
            } else {
#pragma unroll
                for (int e = 0; e < 10; ++e) {
                    const int ix = v0 + e;
                    const float tv = rp[min(max(ix, 0), p.W - 1)];
                    w[e] = (ix >= 0 && ix < p.W) ? tv : 0.f;
                }
                w[10] = 0.f; w[11] = 0.f;
            }
#pragma unroll
            for (int o = 0; o < 4; ++o) {
                float row = 0.f;
#pragma unroll
                for (int kx = 0; kx < 4; ++kx) row += t4[kx] * w[2 * o + kx];
                acc[o] += t4[ky] * row;
            }
        }
        const long oi = (nc * p.Ho + oy) * (long)p.Wo + ox0;
        const int nv = min(4, p.Wo - ox0);
        float v[4];
#pragma unroll
        for (int o = 0; o < 4; ++o) {
            v[o] = acc[o] * (p.gain * (1.f / 64.f));
            if (p.act == CCVS_ACT_LRELU) v[o] = lrelu01(v[o]);
        }
        if (nv == 4) {
            if (p.res) {
                const F32Quad r4 = *reinterpret_cast<const F32Quad*>(p.res + oi);
#pragma unroll
                for (int o = 0; o < 4; ++o) v[o] += r4.v[o];
            }
            F32Quad o4;
#pragma unroll
            for (int o = 0; o < 4; ++o) o4.v[o] = v[o] * p.out_scale;
            *reinterpret_cast<F32Quad*>(p.y + oi) = o4;
        } else {
            for (int o = 0; o < nv; ++o) p.y[oi + o] = (v[o] + (p.res ? p.res[oi + o] : 0.f)) * p.out_scale;
        }
    }
}

// The same blur + decimate through an LDS tile: a workgroup owns 16 x 64 outputs of one plane; their 34 x 130 input window is
// staged once (16 bytes per lane, zeros outside the image) instead of being fetched as twelve overlapping 16-byte loads per
// thread through L1, and each thread reads the 4 x 10 window of its four outputs as three 16-byte LDS reads per row.  Per
// output the same sums in the same order as down2_kernel (rows outside the image add zeros instead of being skipped).
#define DOWN_TH 16
#define DOWN_TW 64
__global__ __launch_bounds__(256) void down2_tile_kernel(FirK p, int tiles_x, GridWalk gw) {
    constexpr int IH = 2 * DOWN_TH + 2, IWp = 2 * DOWN_TW + 4;   // 34 rows x 132 floats (130 used)
    __shared__ __attribute__((aligned(16))) float tile[IH * IWp];
    const int tid = threadIdx.x;
    const float t4[4] = {1.f, 3.f, 3.f, 1.f};
    GRID_WALK_BEGIN(gw, bx, by, bz)
    (void)bz;
    const int ty = bx / tiles_x, tx = bx - ty * tiles_x;
    const long nc = by;
    if (w_ != (long)blockIdx.x) __syncthreads();
    const float* xp = p.x + nc * (long)p.H * p.W;
    const int iy0 = ty * DOWN_TH * 2 - p.pad0, ix0 = tx * DOWN_TW * 2 - p.pad0;
    for (int e = tid; e < IH * (IWp / 4); e += 256) {
        const int r = e / (IWp / 4), c = (e - r * (IWp / 4)) * 4;
        const int iy = iy0 + r, ix = ix0 + c;
        f32x4 v4;
        if (iy >= 0 && iy < p.H && ix >= 0 && ix + 3 < p.W) {
            const F32Quad q = *reinterpret_cast<const F32Quad*>(xp + (long)iy * p.W + ix);
            v4 = f32x4{q.v[0], q.v[1], q.v[2], q.v[3]};
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float t = xp[(long)min(max(iy, 0), p.H - 1) * p.W + min(max(ix + j, 0), p.W - 1)];
                v4[j] = (iy >= 0 && iy < p.H && ix + j >= 0 && ix + j < p.W) ? t : 0.f;
            }
        }
        *reinterpret_cast<f32x4*>(&tile[r * IWp + c]) = v4;
    }
    __syncthreads();
    const int row = tid >> 4, cg = tid & 15;
    const int oy = ty * DOWN_TH + row, ox0 = tx * DOWN_TW + cg * 4;
    if (oy < p.Ho && ox0 < p.Wo) {
        float acc[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ky = 0; ky < 4; ++ky) {
            float w[12];
#pragma unroll
            for (int q = 0; q < 3; ++q) {
                const f32x4 r4 = *reinterpret_cast<const f32x4*>(&tile[(2 * row + ky) * IWp + 8 * cg + 4 * q]);
#pragma unroll
                for (int e = 0; e < 4; ++e) w[4 * q + e] = r4[e];
            }
#pragma unroll
            for (int o = 0; o < 4; ++o) {
                float rsum = 0.f;
#pragma unroll
                for (int kx = 0; kx < 4; ++kx) rsum += t4[kx] * w[2 * o + kx];
                acc[o] += t4[ky] * rsum;
            }
        }
        const long oi = (nc * p.Ho + oy) * (long)p.Wo + ox0;
        const int nv = min(4, p.Wo - ox0);
        float v[4];
#pragma unroll
        for (int o = 0; o < 4; ++o) {
            v[o] = acc[o] * (p.gain * (1.f / 64.f));
            if (p.act == CCVS_ACT_LRELU) v[o] = lrelu01(v[o]);
        }
        if (nv == 4) {
            if (p.res) {
                const F32Quad r4 = *reinterpret_cast<const F32Quad*>(p.res + oi);
#pragma unroll
                for (int o = 0; o < 4; ++o) v[o] += r4.v[o];
            }
            F32Quad o4;
#pragma unroll
            for (int o = 0; o < 4; ++o) o4.v[o] = v[o] * p.out_scale;
            *reinterpret_cast<F32Quad*>(p.y + oi) = o4;
        } else {
            for (int o = 0; o < nv; ++o) p.y[oi + o] = (v[o] + (p.res ? p.res[oi + o] : 0.f)) * p.out_scale;
        }
    }
    GRID_WALK_END
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
        const int tiles_x = cdiv(k.Wo, BLUR_TW), tiles_y = cdiv(k.Ho, BLUR_TH);
        const GridWalk gw = grid_walk((long)tiles_x * tiles_y, NC, 1);
        hipLaunchKernelGGL(blur4x4_tile_kernel, dim3(limited_grid(gw.total, stream, 8)), dim3(256), 0, st, k, tiles_x, gw);
    } else if (up == 2 && down == 1 && pad0 == 2 && pad1 == 1) {
        const long work = NC * (long)H * W;
        const unsigned blocks = limited_grid(cdiv64(work, 256) < 65536 * 16 ? cdiv64(work, 256) : 65536 * 16, stream, 8);
        if (W % 2 == 0) {
            const long work2 = NC * (long)H * (W / 2);
            const unsigned blocks2 = limited_grid(cdiv64(work2, 256) < 65536 * 16 ? cdiv64(work2, 256) : 65536 * 16, stream, 8);
            hipLaunchKernelGGL(upsample2x2_kernel, dim3(blocks2), dim3(256), 0, st, k);
        } else {
            hipLaunchKernelGGL(upsample2_kernel, dim3(blocks), dim3(256), 0, st, k);
        }
    } else if (up == 1 && down == 2 && k.Wo >= 32 && k.Ho >= 8) {   // (small planes: the direct form below)
        const int tiles_x = cdiv(k.Wo, DOWN_TW), tiles_y = cdiv(k.Ho, DOWN_TH);
        const GridWalk gw = grid_walk((long)tiles_x * tiles_y, NC, 1);
        hipLaunchKernelGGL(down2_tile_kernel, dim3(limited_grid(gw.total, stream, 8)), dim3(256), 0, st, k, tiles_x, gw);
    } else if (up == 1 && down == 2) {
        const long work = NC * k.Ho * ((k.Wo + 3) / 4);
        const unsigned blocks = limited_grid(cdiv64(work, 256) < 65536 * 16 ? cdiv64(work, 256) : 65536 * 16, stream, 8);
        hipLaunchKernelGGL(down2_kernel, dim3(blocks), dim3(256), 0, st, k);
    } else {
        const long work = NC * k.Ho * k.Wo;
        const unsigned blocks = limited_grid(cdiv64(work, 256) < 65536 * 16 ? cdiv64(work, 256) : 65536 * 16, stream, 8);
        hipLaunchKernelGGL(upfirdn2d_generic_kernel, dim3(blocks), dim3(256), 0, st, k);
    }
    CCVS_CHECK_LAUNCH("ccvs_upfirdn2d");
    return CCVS_OK;
}

// Transposed 4x4 stride-2 pad-1 depthwise filter, gather form: y + 1 = 2*iy + ky.  One thread per INPUT
// pixel (i, j) produces the output quad (2i..2i+1, 2j..2j+1) from the 3 x 3 input neighbourhood:
// row 2i takes (ky=1, iy=i), (ky=3, iy=i-1); row 2i+1 takes (ky=0, iy=i+1), (ky=2, iy=i); same along x.
__global__ __launch_bounds__(256) void dwconvT4x4s2_kernel(const float* __restrict__ x, long x_sN, const float* __restrict__ w,
                                                           float* __restrict__ y, long y_sN, long N, int C, int H, int W) {
    const int Ho = 2 * H, Wo = 2 * W;
    const long total = N * C * (long)H * W;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int jx = (int)(i % W);
        const long t = i / W;
        const int iy = (int)(t % H);
        const long nc = t / H;
        const int c = (int)(nc % C);
        const long n = nc / C;
        const float* xp = x + n * x_sN + (long)c * H * W;
        const float* wp = w + c * 16;
        float v[3][3];
#pragma unroll
        for (int a = 0; a < 3; ++a)
#pragma unroll
            for (int b = 0; b < 3; ++b) {
                const int yy = iy + a - 1, xx = jx + b - 1;
                const float tv = xp[(long)min(max(yy, 0), H - 1) * W + min(max(xx, 0), W - 1)];
                v[a][b] = (yy >= 0 && yy < H && xx >= 0 && xx < W) ? tv : 0.f;
            }
        // (output parity, tap) -> neighbourhood index: parity 0: ky {1,3} -> rows {1,0}; parity 1: ky {0,2} -> rows {2,1}
        const int kyt[2][2] = {{1, 3}, {0, 2}}, nyt[2][2] = {{1, 0}, {2, 1}};
#pragma unroll
        for (int a = 0; a < 2; ++a) {
            float o2[2];
#pragma unroll
            for (int b = 0; b < 2; ++b) {
                float acc = 0.f;
#pragma unroll
                for (int ta = 0; ta < 2; ++ta)
#pragma unroll
                    for (int tb = 0; tb < 2; ++tb) acc += v[nyt[a][ta]][nyt[b][tb]] * wp[kyt[a][ta] * 4 + kyt[b][tb]];
                o2[b] = acc;
            }
            *reinterpret_cast<float2*>(y + n * y_sN + ((long)c * Ho + 2 * iy + a) * Wo + 2 * jx) = make_float2(o2[0], o2[1]);
        }
    }
}

// The same for TWO input pixels (jx, jx + 1) per thread (W even): 3 x 4 neighbourhood in, a 2 x 4 output block out as two
// 16-byte stores; per output the sum is that of dwconvT4x4s2_kernel.
__global__ __launch_bounds__(256) void dwconvT4x4s2x2_kernel(const float* __restrict__ x, long x_sN, const float* __restrict__ w,
                                                             float* __restrict__ y, long y_sN, long N, int C, int H, int W) {
    const int Ho = 2 * H, Wo = 2 * W, W2 = W >> 1;
    const long total = N * C * (long)H * W2;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int jx = (int)(i % W2) * 2;
        const long t = i / W2;
        const int iy = (int)(t % H);
        const long nc = t / H;
        const int c = (int)(nc % C);
        const long n = nc / C;
        const float* xp = x + n * x_sN + (long)c * H * W;
        const float* wp = w + c * 16;
        float v[3][4];
#pragma unroll
        for (int a = 0; a < 3; ++a)
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                const int yy = iy + a - 1, xx = jx + b - 1;
                const float tv = xp[(long)min(max(yy, 0), H - 1) * W + min(max(xx, 0), W - 1)];
                v[a][b] = (yy >= 0 && yy < H && xx >= 0 && xx < W) ? tv : 0.f;
            }
        const int kyt[2][2] = {{1, 3}, {0, 2}}, nyt[2][2] = {{1, 0}, {2, 1}};
#pragma unroll
        for (int a = 0; a < 2; ++a) {
            F32Quad o4;
#pragma unroll
            for (int bi = 0; bi < 2; ++bi)       // input column jx + bi
#pragma unroll
                for (int b = 0; b < 2; ++b) {    // output parity along x
                    float acc = 0.f;
#pragma unroll
                    for (int ta = 0; ta < 2; ++ta)
#pragma unroll
                        for (int tb = 0; tb < 2; ++tb) acc += v[nyt[a][ta]][bi + nyt[b][tb]] * wp[kyt[a][ta] * 4 + kyt[b][tb]];
                    o4.v[2 * bi + b] = acc;
                }
            *reinterpret_cast<F32Quad*>(y + n * y_sN + ((long)c * Ho + 2 * iy + a) * Wo + 2 * jx) = o4;
        }
    }
}

// FOUR input pixels (jx .. jx + 3) per thread (W % 4 == 0): a 3 x 6 neighbourhood in -- per row one aligned 16-byte load plus the two
// edge columns -- and a 2 x 8 output block out as four 16-byte stores.  The two-pixel form spends its time on index arithmetic and
// 28 load instructions per 32 bytes stored (2.5 TB/s); this one issues 25 per 128 bytes.  Per output the sum of dwconvT4x4s2_kernel.
__global__ __launch_bounds__(256) void dwconvT4x4s2x4_kernel(const float* __restrict__ x, long x_sN, const float* __restrict__ w,
                                                             float* __restrict__ y, long y_sN, long N, int C, int H, int W) {
    const int Ho = 2 * H, Wo = 2 * W, W4 = W >> 2;
    const long total = N * C * (long)H * W4;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int jx = (int)(i % W4) * 4;
        const long t = i / W4;
        const int iy = (int)(t % H);
        const long nc = t / H;
        const int c = (int)(nc % C);
        const long n = nc / C;
        const float* xp = x + n * x_sN + (long)c * H * W;
        const float* wp = w + c * 16;
        float v[3][6];
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            const int yy = iy + a - 1;
            const bool rin = yy >= 0 && yy < H;
            const float* rp = xp + (long)min(max(yy, 0), H - 1) * W;
            const F32Quad m = *reinterpret_cast<const F32Quad*>(rp + jx);
            const float l = rp[max(jx - 1, 0)], r = rp[min(jx + 4, W - 1)];
            v[a][0] = (rin && jx > 0) ? l : 0.f;
#pragma unroll
            for (int b = 0; b < 4; ++b) v[a][1 + b] = rin ? m.v[b] : 0.f;
            v[a][5] = (rin && jx + 4 < W) ? r : 0.f;
        }
        const int kyt[2][2] = {{1, 3}, {0, 2}}, nyt[2][2] = {{1, 0}, {2, 1}};
#pragma unroll
        for (int a = 0; a < 2; ++a) {
            float* yp = y + n * y_sN + ((long)c * Ho + 2 * iy + a) * Wo + 2 * jx;
#pragma unroll
            for (int h = 0; h < 2; ++h) {        // two 16-byte stores per output row
                F32Quad o4;
#pragma unroll
                for (int bj = 0; bj < 2; ++bj) {
                    const int bi = 2 * h + bj;   // input column jx + bi
#pragma unroll
                    for (int b = 0; b < 2; ++b) {
                        float acc = 0.f;
#pragma unroll
                        for (int ta = 0; ta < 2; ++ta)
#pragma unroll
                            for (int tb = 0; tb < 2; ++tb) acc += v[nyt[a][ta]][bi + nyt[b][tb]] * wp[kyt[a][ta] * 4 + kyt[b][tb]];
                        o4.v[2 * bj + b] = acc;
                    }
                }
                *reinterpret_cast<F32Quad*>(yp + 4 * h) = o4;
            }
        }
    }
}

extern "C" int ccvs_dwconvT4x4s2(const float* x, int64_t x_sN, const float* w, float* y, int64_t y_sN, int32_t N, int32_t C, int32_t H,
                                 int32_t W, void* stream) {
    CCVS_REQUIRE(x && w && y, "ccvs_dwconvT4x4s2: null pointer");
    CCVS_REQUIRE(N > 0 && C > 0 && H > 0 && W > 0, "ccvs_dwconvT4x4s2: empty tensor");
    const long work = (long)N * C * H * W;
    const unsigned blocks = limited_grid(cdiv64(work, 256) < 65536 * 16 ? cdiv64(work, 256) : 65536 * 16, stream, 8);
    static const int quad_form = getenv("CCVS_DWCONVT_X4") ? atoi(getenv("CCVS_DWCONVT_X4")) : 1;
    if (quad_form && W % 4 == 0 && x_sN % 4 == 0) {
        const long work4 = (long)N * C * H * (W / 4);
        const unsigned blocks4 = limited_grid(cdiv64(work4, 256) < 65536 * 16 ? cdiv64(work4, 256) : 65536 * 16, stream, 8);
        hipLaunchKernelGGL(dwconvT4x4s2x4_kernel, dim3(blocks4), dim3(256), 0, (hipStream_t)stream, x, (long)x_sN, w, y, (long)y_sN, (long)N, C, H, W);
    } else if (W % 2 == 0) {
        const long work2 = (long)N * C * H * (W / 2);
        const unsigned blocks2 = limited_grid(cdiv64(work2, 256) < 65536 * 16 ? cdiv64(work2, 256) : 65536 * 16, stream, 8);
        hipLaunchKernelGGL(dwconvT4x4s2x2_kernel, dim3(blocks2), dim3(256), 0, (hipStream_t)stream, x, (long)x_sN, w, y, (long)y_sN, (long)N, C, H, W);
    } else {
        hipLaunchKernelGGL(dwconvT4x4s2_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, x, (long)x_sN, w, y, (long)y_sN, (long)N, C, H, W);
    }
    CCVS_CHECK_LAUNCH("ccvs_dwconvT4x4s2");
    return CCVS_OK;
}
