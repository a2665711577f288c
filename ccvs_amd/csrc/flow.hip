// Cost volume, bilinear back-warp and the confidence fusion / occlusion blend of the
// decoder's InterBlock (reference skip_autoencoder.py:120-128,179-265).  HBM / LDS bound.
#include "common.h"
#include <stdlib.h>

// ---------------------------------------------------------------------------------------
// 7x7 displacement correlation (modules/correlation.py:11-100):
//   out[n][7*(dy+3)+(dx+3)][y][x] = (1/C) sum_c A[n/div][c][y*s][x*s] * B[n][c][(y+dy)*s][(x+dx)*s]
// One workgroup = 8x32 output pixels, one lane per pixel with 49 running sums in
// registers; per chunk of CORR_CC channels the B halo tile is staged in LDS (zero outside the
// image: the reference's padded `rearrange` copy is never materialised).  Only the samples
// B[(y+dy)*s][(x+dx)*s] are ever read, so the tile holds the stride-s SUBSAMPLED halo,
// (8+6) x (32+6) values per channel whatever s: a quarter of the staging at s = 2, and the
// lanes of a wave read consecutive LDS words (with the full-resolution tile they were 2 words
// apart: every read a two-way bank conflict).
// ---------------------------------------------------------------------------------------
#define CORR_CC 8
typedef float f32x2 __attribute__((ext_vector_type(2)));

template <int S>
__global__ __launch_bounds__(256) void correlation7x7_kernel(const float* __restrict__ first, const float* __restrict__ second,
                                                             float* __restrict__ out, int C, int H, int W, int Ho, int Wo,
                                                             int first_div, int lrelu, int tiles_x, GridWalk gw) {
    constexpr int TH = 8, TW = 32;
    constexpr int IH = TH + 6, IW = TW + 6;   // in units of s pixels
    __shared__ float bt[CORR_CC][IH * IW];
    const int tid = threadIdx.x;
    GRID_WALK_BEGIN(gw, bx, by, bz)
    (void)bz;
    const int ty = bx / tiles_x, tx = bx - ty * tiles_x;
    const int n = by;
    const int py = tid >> 5, px = tid & 31;
    const int oy = ty * TH + py, ox = tx * TW + px;
    const bool live = (oy < Ho && ox < Wo);
    const int iy0 = ty * TH * S - 3 * S, ix0 = tx * TW * S - 3 * S;
    const float* A = first + (long)(n / first_div) * C * H * W;
    const float* B = second + (long)n * C * H * W;

    float acc[49];
#pragma unroll
    for (int d = 0; d < 49; ++d) acc[d] = 0.f;

    // every global value of a channel chunk is requested in one burst, a whole chunk ahead of its use (see the two-pixel form)
    constexpr int NE = (IH * IW + 255) / 256;
    long eoff[NE];
#pragma unroll
    for (int j = 0; j < NE; ++j) {
        const int e = tid + 256 * j;
        eoff[j] = -2;
        if (e < IH * IW) {
            const int r = e / IW, c = e - r * IW;
            const int gy = iy0 + r * S, gx = ix0 + c * S;
            eoff[j] = (gy >= 0 && gy < H && gx >= 0 && gx < W) ? (long)gy * W + gx : -1;
        }
    }
    const long aoff = live ? (long)(oy * S) * W + ox * S : 0;
    float breg[NE][CORR_CC], areg[CORR_CC];
    auto fetch = [&](int c0) {
#pragma unroll
        for (int cc = 0; cc < CORR_CC; ++cc) {
            const long pl = (long)min(c0 + cc, C - 1) * H * W;
#pragma unroll
            for (int j = 0; j < NE; ++j) breg[j][cc] = B[pl + (eoff[j] >= 0 ? eoff[j] : 0)];
            areg[cc] = A[pl + aoff];
        }
    };
    fetch(0);
    for (int c0 = 0; c0 < C; c0 += CORR_CC) {
        __syncthreads();
        float ac[CORR_CC];
#pragma unroll
        for (int cc = 0; cc < CORR_CC; ++cc) {
            ac[cc] = (live && c0 + cc < C) ? areg[cc] : 0.f;
#pragma unroll
            for (int j = 0; j < NE; ++j)
                if (eoff[j] != -2) bt[cc][tid + 256 * j] = (eoff[j] >= 0 && c0 + cc < C) ? breg[j][cc] : 0.f;
        }
        __syncthreads();
        if (c0 + CORR_CC < C) fetch(c0 + CORR_CC);
#pragma unroll
        for (int cc = 0; cc < CORR_CC; ++cc) {
            const float a = ac[cc];
            const float* bp = &bt[cc][py * IW + px];
#pragma unroll
            for (int dy = 0; dy < 7; ++dy)
#pragma unroll
                for (int dx = 0; dx < 7; ++dx) acc[dy * 7 + dx] += a * bp[dy * IW + dx];
        }
    }
    if (live) {
        const float inv = 1.f / (float)C;
        float* o = out + (long)n * 49 * Ho * Wo + (long)oy * Wo + ox;
#pragma unroll
        for (int d = 0; d < 49; ++d) {
            float v = acc[d] * inv;
            if (lrelu) v = lrelu01(v);
            o[(long)d * Ho * Wo] = v;
        }
    }
    GRID_WALK_END   // (every channel chunk starts with a barrier: the next block's staging waits for this block's reads)
}

// Two pixels per lane (Wo even, planes of at least 64 columns): the one-pixel form is bound by the LDS pipe (49 ds_read_b32 per
// pixel and channel; SQ_LDS_IDX_ACTIVE / SQ_BUSY_CYCLES = 1.05, no conflicts).  A lane that owns the pixel pair (x, x + 1) needs the
// 8 consecutive values B[y + dy][x .. x + 7] per row and channel: four 8-byte LDS reads (256 B/clk) serve 14 products instead of
// fourteen 4-byte reads (128 B/clk), and the 49 results per pixel leave as 8-byte stores.  Workgroup = 8 x 64 output pixels;
// per (pixel, displacement) the channels are added in the same order as above: the same bits.
template <int S>
__global__ __launch_bounds__(256) void correlation7x7x2_kernel(const float* __restrict__ first, const float* __restrict__ second,
                                                               float* __restrict__ out, int C, int H, int W, int Ho, int Wo,
                                                               int first_div, int lrelu, int tiles_x, GridWalk gw) {
    constexpr int TH = 8, TW = 64;
    constexpr int IH = TH + 6, IW = TW + 6;   // in units of s pixels; IW even: a lane's pair is 8-byte aligned in LDS
    __shared__ __attribute__((aligned(16))) float bt[CORR_CC][IH * IW];
    const int tid = threadIdx.x;
    GRID_WALK_BEGIN(gw, bx, by, bz)
    (void)bz;
    const int ty = bx / tiles_x, tx = bx - ty * tiles_x;
    const int n = by;
    const int py = tid >> 5, px = (tid & 31) * 2;
    const int oy = ty * TH + py, ox = tx * TW + px;
    const bool live = (oy < Ho && ox < Wo);    // Wo even: the pair is in or out together
    const int iy0 = ty * TH * S - 3 * S, ix0 = tx * TW * S - 3 * S;
    const float* A = first + (long)(n / first_div) * C * H * W;
    const float* B = second + (long)n * C * H * W;

    float acc0[49], acc1[49];
#pragma unroll
    for (int d = 0; d < 49; ++d) { acc0[d] = 0.f; acc1[d] = 0.f; }

    // The kernel holds 98 sums per lane: one workgroup per CU, ONE wave per SIMD -- nothing hides a load but the wave's own
    // independent work.  Every global value of a channel chunk (this thread's share of the halo tile of `second`, and the two
    // `first` values of each of its channels) is therefore requested in ONE burst, a whole chunk before it is used: the chunk
    // c0 + CORR_CC is in flight while chunk c0 is multiplied.  (Rounds 1-3 fetched first[c] inside the channel loop, right in
    // front of its use -- a full memory round trip per channel: 64 us per tile against 4 us of arithmetic.)
    constexpr int NE = (IH * IW + 255) / 256;
    long eoff[NE];      // offset of halo element e = tid + 256 j inside a channel plane; -1: outside the image, -2: no such element
#pragma unroll
    for (int j = 0; j < NE; ++j) {
        const int e = tid + 256 * j;
        eoff[j] = -2;
        if (e < IH * IW) {
            const int r = e / IW, c = e - r * IW;
            const int gy = iy0 + r * S, gx = ix0 + c * S;
            eoff[j] = (gy >= 0 && gy < H && gx >= 0 && gx < W) ? (long)gy * W + gx : -1;
        }
    }
    const long aoff = live ? (long)(oy * S) * W + ox * S : 0;
    float breg[NE][CORR_CC], a0reg[CORR_CC], a1reg[CORR_CC];
    auto fetch = [&](int c0) {
#pragma unroll
        for (int cc = 0; cc < CORR_CC; ++cc) {
            const long pl = (long)min(c0 + cc, C - 1) * H * W;
#pragma unroll
            for (int j = 0; j < NE; ++j) breg[j][cc] = B[pl + (eoff[j] >= 0 ? eoff[j] : 0)];
            a0reg[cc] = A[pl + aoff];
            a1reg[cc] = A[pl + aoff + (live ? S : 0)];
        }
    };
    fetch(0);
    for (int c0 = 0; c0 < C; c0 += CORR_CC) {
        __syncthreads();
        float a0c[CORR_CC], a1c[CORR_CC];
#pragma unroll
        for (int cc = 0; cc < CORR_CC; ++cc) {
            const bool on = live && c0 + cc < C;
            a0c[cc] = on ? a0reg[cc] : 0.f;
            a1c[cc] = on ? a1reg[cc] : 0.f;
#pragma unroll
            for (int j = 0; j < NE; ++j)
                if (eoff[j] != -2) bt[cc][tid + 256 * j] = (eoff[j] >= 0 && c0 + cc < C) ? breg[j][cc] : 0.f;
        }
        __syncthreads();
        if (c0 + CORR_CC < C) fetch(c0 + CORR_CC);
#pragma unroll
        for (int cc = 0; cc < CORR_CC; ++cc) {
            const float a0 = a0c[cc], a1 = a1c[cc];
            const float* bp = &bt[cc][py * IW + px];
#pragma unroll
            for (int dy = 0; dy < 7; ++dy) {
                float r[8];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const f32x2 t2 = *reinterpret_cast<const f32x2*>(bp + dy * IW + 2 * q);   // 8-byte aligned: ds_read_b64
                    r[2 * q] = t2[0]; r[2 * q + 1] = t2[1];
                }
#pragma unroll
                for (int dx = 0; dx < 7; ++dx) {
                    acc0[dy * 7 + dx] += a0 * r[dx];
                    acc1[dy * 7 + dx] += a1 * r[dx + 1];
                }
            }
        }
    }
    if (live) {
        const float inv = 1.f / (float)C;
        float* o = out + (long)n * 49 * Ho * Wo + (long)oy * Wo + ox;
#pragma unroll
        for (int d = 0; d < 49; ++d) {
            float v0 = acc0[d] * inv, v1 = acc1[d] * inv;
            if (lrelu) { v0 = lrelu01(v0); v1 = lrelu01(v1); }
            F32Pair w2;
            w2.x = v0; w2.y = v1;
            *reinterpret_cast<F32Pair*>(o + (long)d * Ho * Wo) = w2;
        }
    }
    GRID_WALK_END
}

extern "C" int ccvs_correlation7x7(const float* first, const float* second, float* out, int32_t N, int32_t C, int32_t H, int32_t W,
                                   int32_t stride, int32_t first_div, int32_t lrelu, void* stream) {
    CCVS_REQUIRE(first && second && out, "ccvs_correlation7x7: null pointer");
    CCVS_REQUIRE(N > 0 && C > 0 && H > 0 && W > 0 && first_div >= 1, "ccvs_correlation7x7: bad shape");
    CCVS_REQUIRE(stride == 1 || stride == 2, "ccvs_correlation7x7: stride %d unsupported", stride);
    const int Ho = cdiv(H, stride), Wo = cdiv(W, stride);
    hipStream_t st = (hipStream_t)stream;
    static const int pair_form = getenv("CCVS_CORR_PAIR") ? atoi(getenv("CCVS_CORR_PAIR")) : 1;
    if (pair_form && Wo % 2 == 0 && Wo >= 64) {
        const int tiles_x2 = cdiv(Wo, 64), tiles_y2 = cdiv(Ho, 8);
        const GridWalk gw2 = grid_walk((long)tiles_x2 * tiles_y2, N, 1);
        const dim3 grid2(limited_grid(gw2.total, stream, 4));
        if (stride == 1)
            hipLaunchKernelGGL((correlation7x7x2_kernel<1>), grid2, dim3(256), 0, st, first, second, out, C, H, W, Ho, Wo, first_div, lrelu, tiles_x2, gw2);
        else
            hipLaunchKernelGGL((correlation7x7x2_kernel<2>), grid2, dim3(256), 0, st, first, second, out, C, H, W, Ho, Wo, first_div, lrelu, tiles_x2, gw2);
        CCVS_CHECK_LAUNCH("ccvs_correlation7x7");
        return CCVS_OK;
    }
    const int tiles_x = cdiv(Wo, 32), tiles_y = cdiv(Ho, 8);
    const GridWalk gw = grid_walk((long)tiles_x * tiles_y, N, 1);
    const dim3 grid(limited_grid(gw.total, stream, stride == 1 ? 8 : 4));
    if (stride == 1)
        hipLaunchKernelGGL((correlation7x7_kernel<1>), grid, dim3(256), 0, st, first, second, out, C, H, W, Ho, Wo, first_div, lrelu, tiles_x, gw);
    else
        hipLaunchKernelGGL((correlation7x7_kernel<2>), grid, dim3(256), 0, st, first, second, out, C, H, W, Ho, Wo, first_div, lrelu, tiles_x, gw);
    CCVS_CHECK_LAUNCH("ccvs_correlation7x7");
    return CCVS_OK;
}

// ---------------------------------------------------------------------------------------
// backwarp: grid_sample(bilinear, zeros, align_corners=False) on the pixel-centre grid
// (skip_autoencoder.py:120-128).  Sample position in input pixels, following the
// reference's arithmetic: g = (2x+1)/W - 1 + f / ((W-1)/2);  ix = ((g+1)*W - 1)/2.
// ---------------------------------------------------------------------------------------
struct Bilin {
    int o00, o01, o10, o11;  // plane offsets (clamped to 0 when outside, weight then 0)
    float w00, w01, w10, w11;
};

__device__ __forceinline__ Bilin bilin_setup(int x, int y, float fx, float fy, int H, int W) {
    const float gx = ((2.f * x + 1.f) / W - 1.f) + fx / ((W - 1.0f) / 2.0f);
    const float gy = ((2.f * y + 1.f) / H - 1.f) + fy / ((H - 1.0f) / 2.0f);
    const float ix = ((gx + 1.f) * W - 1.f) * 0.5f;
    const float iy = ((gy + 1.f) * H - 1.f) * 0.5f;
    const float x0f = floorf(ix), y0f = floorf(iy);
    const float tx = ix - x0f, ty = iy - y0f;
    Bilin b;
    b.w00 = (1.f - tx) * (1.f - ty); b.w01 = tx * (1.f - ty); b.w10 = (1.f - tx) * ty; b.w11 = tx * ty;
    // clamp before the int conversion so that huge flows cannot overflow
    const float xc = fminf(fmaxf(x0f, -2.f), (float)W + 1.f), yc = fminf(fmaxf(y0f, -2.f), (float)H + 1.f);
    const int x0 = (int)xc, y0 = (int)yc, x1 = x0 + 1, y1 = y0 + 1;
    const bool vx0 = (x0 >= 0 && x0 < W), vx1 = (x1 >= 0 && x1 < W), vy0 = (y0 >= 0 && y0 < H), vy1 = (y1 >= 0 && y1 < H);
    // corners outside the image: weight 0 and a clamped (valid) offset, so that the four loads of a
    // sample are unconditional (predicated loads in the channel loop would serialise)
    b.o00 = (vx0 && vy0) ? y0 * W + x0 : 0;  if (!(vx0 && vy0)) b.w00 = 0.f;
    b.o01 = (vx1 && vy0) ? y0 * W + x1 : 0;  if (!(vx1 && vy0)) b.w01 = 0.f;
    b.o10 = (vx0 && vy1) ? y1 * W + x0 : 0;  if (!(vx0 && vy1)) b.w10 = 0.f;
    b.o11 = (vx1 && vy1) ? y1 * W + x1 : 0;  if (!(vx1 && vy1)) b.w11 = 0.f;
    return b;
}

__device__ __forceinline__ float bilin_sample(const float* __restrict__ plane, const Bilin& b) {
    return ((plane[b.o00] * b.w00 + plane[b.o01] * b.w01) + plane[b.o10] * b.w10) + plane[b.o11] * b.w11;
}

// Four-pixel form of the warp kernels (W % 4 == 0): a lane owns 4 consecutive pixels of a row and writes them with ONE
// 16-byte store per channel -- with one pixel per lane and 4-byte stores the warps ran at 2.5 TB/s, 0.6 of what a plain
// copy reaches on this chip; this form reaches 3.96 (tools/mem_bench.py, 120 x 96 x 256^2).  The two x-taps of a sample row
// are adjacent in memory and come in one 8-byte load: each row is re-based on a column pair (xb, xb + 1) that lies inside
// the image and the weights move with it (a corner outside the image keeps weight 0), so a sample is the same four products
// summed in the same order as bilin_sample.  All accesses are dword-aligned only (F32Pair / F32Quad, common.h).
struct BilinPair {
    int o0, o1;              // offsets of the column pairs in rows y0, y1
    float a0, b0, a1, b1;    // weights of (pair.x, pair.y) in row y0, row y1
};

// the same set-up with the pair's column and the two rows kept as coordinates (the windowed kernels address an LDS copy of
// the source window with them); bilin_setup_pair IS this function, so the weights cannot differ between the two forms
struct BilinWin {
    BilinPair q;
    int xb, r0, r1;          // first column of the pairs; rows of the pairs (0 where the row is outside: its weights are 0)
};

__device__ __forceinline__ BilinWin bilin_setup_win(int x, int y, float fx, float fy, int H, int W) {
    const float gx = ((2.f * x + 1.f) / W - 1.f) + fx / ((W - 1.0f) / 2.0f);
    const float gy = ((2.f * y + 1.f) / H - 1.f) + fy / ((H - 1.0f) / 2.0f);
    const float ix = ((gx + 1.f) * W - 1.f) * 0.5f;
    const float iy = ((gy + 1.f) * H - 1.f) * 0.5f;
    const float x0f = floorf(ix), y0f = floorf(iy);
    const float tx = ix - x0f, ty = iy - y0f;
    const float w00 = (1.f - tx) * (1.f - ty), w01 = tx * (1.f - ty), w10 = (1.f - tx) * ty, w11 = tx * ty;
    const float xc = fminf(fmaxf(x0f, -2.f), (float)W + 1.f), yc = fminf(fmaxf(y0f, -2.f), (float)H + 1.f);
    const int x0 = (int)xc, y0 = (int)yc, y1 = y0 + 1;
    const bool vy0 = (y0 >= 0 && y0 < H), vy1 = (y1 >= 0 && y1 < H);
    BilinWin b;
    BilinPair& q = b.q;
    int xb = 0;
    q.a0 = 0.f; q.b0 = 0.f; q.a1 = 0.f; q.b1 = 0.f;
    if (x0 >= 0 && x0 + 1 < W) {          // both columns inside
        xb = x0; q.a0 = w00; q.b0 = w01; q.a1 = w10; q.b1 = w11;
    } else if (x0 == -1) {                // only x1 = 0 inside: it is the pair's first element
        xb = 0; q.a0 = w01; q.a1 = w11;
    } else if (x0 == W - 1) {             // only x0 = W - 1 inside: the pair's second element
        xb = W - 2; q.b0 = w00; q.b1 = w10;
    }
    if (!vy0) { q.a0 = 0.f; q.b0 = 0.f; }
    if (!vy1) { q.a1 = 0.f; q.b1 = 0.f; }
    b.xb = xb;
    b.r0 = vy0 ? y0 : 0;
    b.r1 = vy1 ? y1 : 0;
    q.o0 = b.r0 * W + xb;
    q.o1 = b.r1 * W + xb;
    return b;
}

__device__ __forceinline__ BilinPair bilin_setup_pair(int x, int y, float fx, float fy, int H, int W) {
    return bilin_setup_win(x, y, fx, fy, H, W).q;
}

// the four products of a sample, summed in ONE order wherever the two pairs come from (global memory or an LDS window)
__device__ __forceinline__ float bilin_mix(const F32Pair& r0, const F32Pair& r1, const BilinPair& q) {
    return ((r0.x * q.a0 + r0.y * q.b0) + r1.x * q.a1) + r1.y * q.b1;
}

__device__ __forceinline__ float bilin_sample_pair(const float* __restrict__ plane, const BilinPair& q) {
    const F32Pair r0 = *reinterpret_cast<const F32Pair*>(plane + q.o0);
    const F32Pair r1 = *reinterpret_cast<const F32Pair*>(plane + q.o1);
    return bilin_mix(r0, r1, q);
}

#define WARP_CCH 16  // channels per thread
#define WARP4_CCH 8  // ... of the four-pixel kernels

// The k context features of a decode step live in slots of the per-level context ring (and, point-to-point, in a
// separate tensor): item n of the batch of N*k pairs reads source ctx.p[n % k] + (n / k) * ctx.sN[n % k], so the
// reference's torch.stack / repeat of the contexts (skip_autoencoder.py:251) never has to be materialised.
// A dense [N,C,H,W] source is the list {k = 1, p[0] = x, sN[0] = x_sN}.
struct CtxList {
    int k;
    const float* p[CCVS_MAX_CTX];
    long sN[CCVS_MAX_CTX];
};

__global__ __launch_bounds__(256) void backwarp_kernel(CtxList ctx, long x_sC,
                                                       const float* __restrict__ flow, long flow_sN, float mult,
                                                       float* __restrict__ y, long y_sN, long y_sC, int C, int H, int W, GridWalk gw) {
    const int HW = H * W;
    GRID_WALK_BEGIN(gw, bx, by, bz)
    const int pix = bx * 256 + threadIdx.x;
    if (pix >= HW) continue;
    const int n = bz, c0 = by * WARP_CCH;
    const int jn = n % ctx.k;
    const float* x = ctx.p[jn] + (long)(n / ctx.k) * ctx.sN[jn];
    const int py = pix / W, px = pix - py * W;
    const float fx = flow[(long)n * flow_sN + pix] * mult, fy = flow[(long)n * flow_sN + HW + pix] * mult;
    const Bilin b = bilin_setup(px, py, fx, fy, H, W);
    const int cend = min(c0 + WARP_CCH, C);
    for (int c = c0; c < cend; ++c)
        y[(long)n * y_sN + (long)c * y_sC + pix] = bilin_sample(x + (long)c * x_sC, b);
    GRID_WALK_END
}

// First pixel of a thread's quad in the four-pixel warp kernels.  Plain: block bx = 1024 consecutive pixels (4 rows of a 256-wide
// image).  Tiled: block bx = a tile of 4 t x 256 / t pixels, t threads along a row (t = GridWalk.tiled = 16: 64 x 16; needs W % 4t == 0,
// H % (256 / t) == 0).  The gathers of a workgroup then fall into ~20 rows x 3 cache lines per channel instead of ~8 rows x 8 lines
// and a thread's neighbours above and below re-use its lines: with a smooth flow field of sigma 3.2 px at 256^2 the fusion / blend
// tail (k = 15) 2.35 -> 1.46 ms, the back-warp 2.33 -> 1.98, warp + projection 2.02 -> 1.59; with per-pixel noise of the same
// spread 4.02 -> 2.77, 3.53 -> 2.72, 3.42 -> 2.46; never slower (profiles/r04_warp_tile_ab.txt).  Same arithmetic per pixel either
// way (tests/test_ops_gpu.py::test_warp_kernels_tiled_pixel_order).  An LDS copy of the source window (min / max of the pair
// coordinates over the tile, 16-byte row loads, gathers from LDS) was built first and was SLOWER than the plain gathers on the same
// tiles: 3.0 ms against 1.5 -- two barriers and a dependent load phase per context buy nothing the L1 does not already give.
static int warp_tiled(int H, int W) {
    static const int t = getenv("CCVS_WARP_TILED") ? atoi(getenv("CCVS_WARP_TILED")) : 16;   // threads along a tile row; 0: plain
    if (t != 8 && t != 16 && t != 32 && t != 64) return 0;
    return (W % (4 * t) == 0 && H % (256 / t) == 0) ? t : 0;
}
__device__ __forceinline__ int quad_pixel(const GridWalk& gw, int bx, int W) {
    if (gw.tiled) {
        const int t = gw.tiled;
        const int tiles_x = W / (4 * t);
        const int ty = bx / tiles_x, tx = bx - ty * tiles_x;
        const int r = (int)threadIdx.x / t, c = (int)threadIdx.x - r * t;
        return (ty * (256 / t) + r) * W + (tx * t + c) * 4;
    }
    return (bx * 256 + (int)threadIdx.x) * 4;
}

__global__ __launch_bounds__(256) void backwarp4_kernel(CtxList ctx, long x_sC, const float* __restrict__ flow, long flow_sN, float mult,
                                                        float* __restrict__ y, long y_sN, long y_sC, int C, int H, int W, GridWalk gw) {
    const int HW = H * W;
    GRID_WALK_BEGIN(gw, bx, by, bz)
    const int pix = quad_pixel(gw, bx, W);
    if (pix >= HW) continue;
    const int n = bz, c0 = by * WARP4_CCH;
    const int jn = n % ctx.k;
    const float* x = ctx.p[jn] + (long)(n / ctx.k) * ctx.sN[jn];
    const int py = pix / W, px = pix - py * W;
    const F32Quad fx = *reinterpret_cast<const F32Quad*>(flow + (long)n * flow_sN + pix);
    const F32Quad fy = *reinterpret_cast<const F32Quad*>(flow + (long)n * flow_sN + HW + pix);
    BilinPair q[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) q[i] = bilin_setup_pair(px + i, py, fx.v[i] * mult, fy.v[i] * mult, H, W);
    const int cend = min(c0 + WARP4_CCH, C);
    for (int c = c0; c < cend; ++c) {
        const float* pl = x + (long)c * x_sC;
        F32Quad o;
#pragma unroll
        for (int i = 0; i < 4; ++i) o.v[i] = bilin_sample_pair(pl, q[i]);
        *reinterpret_cast<F32Quad*>(y + (long)n * y_sN + (long)c * y_sC + pix) = o;
    }
    GRID_WALK_END
}

static void launch_backwarp(const CtxList& l, long x_sC, const float* flow, long flow_sN, float mult, float* y, long y_sN, long y_sC, int N, int C,
                            int H, int W, void* stream) {
    if (W % 4 == 0) {
        GridWalk gw = grid_walk(cdiv(H * W / 4, 256), cdiv(C, WARP4_CCH), N);
        gw.tiled = warp_tiled(H, W);
        hipLaunchKernelGGL(backwarp4_kernel, dim3(limited_grid(gw.total, stream, 8)), dim3(256), 0, (hipStream_t)stream, l, x_sC, flow, flow_sN, mult,
                           y, y_sN, y_sC, C, H, W, gw);
    } else {
        const GridWalk gw = grid_walk(cdiv(H * W, 256), cdiv(C, WARP_CCH), N);
        hipLaunchKernelGGL(backwarp_kernel, dim3(limited_grid(gw.total, stream, 8)), dim3(256), 0, (hipStream_t)stream, l, x_sC, flow, flow_sN, mult,
                           y, y_sN, y_sC, C, H, W, gw);
    }
}

// ---------------------------------------------------------------------------------------
// Back-warp straight into the packed split-bf16 (P8) input of the first Subpixel convolution: [warped context | flow | occ]
// (skip_autoencoder.py:222-224) as [N][C/8 + 1][hi|lo][H][W] units of 8 bf16 -- the form the convolution kernel stages by
// LDS-DMA, no conversion.  A lane owns ONE pixel and 8 channels: its two 16-byte units are the store width the four-pixel form
// was built for, and consecutive lanes write consecutive units.  Group C/8 holds (flow x, flow y, occlusion, 0 x 5) as stored
// (the warp itself uses flow * mult).  Same samples as backwarp4_kernel (bilin_setup_pair / bilin_sample_pair), split like the
// convolution's staging waves split them (round to nearest even twice): the convolution sees the same operands.
// ---------------------------------------------------------------------------------------
typedef __bf16 wp_bf16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned wp_pk_bf16(float a, float b) {
    const f32x2 v = {a, b};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, wp_bf16x2));
}
__device__ __forceinline__ void wp_split8(const float (&v)[8], uint4& hi, uint4& lo) {
    unsigned h[4], l[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        h[i] = wp_pk_bf16(v[2 * i], v[2 * i + 1]);
        l[i] = wp_pk_bf16(v[2 * i] - __uint_as_float(h[i] << 16), v[2 * i + 1] - __uint_as_float(h[i] & 0xffff0000u));
    }
    hi = make_uint4(h[0], h[1], h[2], h[3]);
    lo = make_uint4(l[0], l[1], l[2], l[3]);
}

__global__ __launch_bounds__(256) void backwarp_p8_kernel(CtxList ctx, long x_sC, const float* __restrict__ fo, long fo_sN, float mult,
                                                          uint4* __restrict__ y, int C, int H, int W, GridWalk gw) {
    // lane = 4 consecutive pixels x 8 channels, sampled exactly like backwarp4_kernel (same values bit for bit); the 32 results
    // leave as 4 x (hi, lo) 16-byte units, one pixel each.  (One pixel per lane -- consecutive lanes writing consecutive units --
    // was built first and lost 40-70 % on the LOAD side: 4.1-5.0 ms against 2.9 on 120 x 96 x 256^2.)
    __shared__ uint4 wp_stage[2048];
    const int HW = H * W;
    const int G = C >> 3;
    GRID_WALK_BEGIN(gw, bx, by, bz)
    const int pix_raw = (bx * 256 + threadIdx.x) * 4;
    const bool live = pix_raw < HW;       // (every thread reaches the barriers below)
    const int pix = live ? pix_raw : 0;
    const int n = bz, g = by;
    const float* f = fo + (long)n * fo_sN + pix;
    const F32Quad fx = *reinterpret_cast<const F32Quad*>(f);
    const F32Quad fy = *reinterpret_cast<const F32Quad*>(f + HW);
    float v[8][4];
    if (g == G) {
        const F32Quad oc = *reinterpret_cast<const F32Quad*>(f + 2 * (long)HW);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            v[0][i] = fx.v[i]; v[1][i] = fy.v[i]; v[2][i] = oc.v[i];
#pragma unroll
            for (int c = 3; c < 8; ++c) v[c][i] = 0.f;
        }
    } else {
        const int jn = n % ctx.k;
        const float* x = ctx.p[jn] + (long)(n / ctx.k) * ctx.sN[jn] + (long)(8 * g) * x_sC;
        const int py = pix / W, px = pix - py * W;
        BilinPair q[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) q[i] = bilin_setup_pair(px + i, py, fx.v[i] * mult, fy.v[i] * mult, H, W);
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            const float* pl = x + (long)c * x_sC;
#pragma unroll
            for (int i = 0; i < 4; ++i) v[c][i] = bilin_sample_pair(pl, q[i]);
        }
    }
    // The workgroup's 1024 pixels x (hi, lo) units go through LDS so that consecutive lanes store consecutive units (a lane's own
    // four units are 64 bytes apart from its neighbour's: written directly, every store instruction touched 32 lines for a quarter
    // of each -- 4.3 ms against 2.9 for the fp32 form).
    __syncthreads();   // (the previous block's reads of the staging area)
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const float vi[8] = {v[0][i], v[1][i], v[2][i], v[3][i], v[4][i], v[5][i], v[6][i], v[7][i]};
        uint4 hi, lo;
        wp_split8(vi, hi, lo);
        wp_stage[i * 256 + threadIdx.x] = hi;          // pixel 4 t + i of the block
        wp_stage[1024 + i * 256 + threadIdx.x] = lo;
    }
    __syncthreads();
    const int pix0 = bx * 1024;
    uint4* dst = y + (((long)n * (G + 1) + g) * 2) * HW + pix0;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int u = threadIdx.x + 256 * j;            // pixel of the block: staged at (u % 4) * 256 + u / 4
        if (pix0 + u < HW) {
            dst[u] = wp_stage[(u & 3) * 256 + (u >> 2)];
            dst[HW + u] = wp_stage[1024 + (u & 3) * 256 + (u >> 2)];
        }
    }
    GRID_WALK_END
}

static int fill_ctx(CtxList& l, const ccvs_ctx_list* c, const char* name);

extern "C" int ccvs_backwarp_p8_ctx(const ccvs_ctx_list* ctx, int64_t x_sC, const float* flow_occ, int64_t fo_sN, float flow_mult, void* y_p8,
                                    int32_t N, int32_t C, int32_t H, int32_t W, void* stream) {
    CCVS_REQUIRE(flow_occ && y_p8, "ccvs_backwarp_p8_ctx: null pointer");
    CtxList l = {};
    const int rc = fill_ctx(l, ctx, "ccvs_backwarp_p8_ctx");
    if (rc != CCVS_OK) return rc;
    CCVS_REQUIRE(N > 0 && N % l.k == 0 && C > 0 && C % 8 == 0 && H > 0 && W >= 4 && W % 4 == 0, "ccvs_backwarp_p8_ctx: bad shape (C %% 8 == 0, W %% 4 == 0)");
    const GridWalk gw = grid_walk(cdiv(H * W / 4, 256), C / 8 + 1, N);
    hipLaunchKernelGGL(backwarp_p8_kernel, dim3(limited_grid(gw.total, stream, 8)), dim3(256), 0, (hipStream_t)stream, l, (long)x_sC, flow_occ,
                       (long)fo_sN, flow_mult, (uint4*)y_p8, C, H, W, gw);
    CCVS_CHECK_LAUNCH("ccvs_backwarp_p8_ctx");
    return CCVS_OK;
}

static int fill_ctx(CtxList& l, const ccvs_ctx_list* c, const char* name) {
    if (!c || c->k < 1 || c->k > CCVS_MAX_CTX) { ccvs_set_error("%s: context list of 1..%d entries expected", name, CCVS_MAX_CTX); return CCVS_ERR_ARG; }
    l.k = c->k;
    for (int j = 0; j < c->k; ++j) {
        if (!c->p[j]) { ccvs_set_error("%s: null context %d", name, j); return CCVS_ERR_ARG; }
        l.p[j] = c->p[j];
        l.sN[j] = (long)c->sN[j];
    }
    return CCVS_OK;
}

extern "C" int ccvs_backwarp(const float* x, int64_t x_sN, int64_t x_sC, const float* flow, int64_t flow_sN, float flow_mult, float* y,
                             int64_t y_sN, int64_t y_sC, int32_t N, int32_t C, int32_t H, int32_t W, void* stream) {
    CCVS_REQUIRE(x && flow && y, "ccvs_backwarp: null pointer");
    CCVS_REQUIRE(N > 0 && C > 0 && H > 0 && W > 0, "ccvs_backwarp: bad shape");
    CtxList l = {};
    l.k = 1; l.p[0] = x; l.sN[0] = (long)x_sN;
    launch_backwarp(l, (long)x_sC, flow, (long)flow_sN, flow_mult, y, (long)y_sN, (long)y_sC, N, C, H, W, stream);
    CCVS_CHECK_LAUNCH("ccvs_backwarp");
    return CCVS_OK;
}

extern "C" int ccvs_backwarp_ctx(const ccvs_ctx_list* ctx, int64_t x_sC, const float* flow, int64_t flow_sN, float flow_mult, float* y,
                                 int64_t y_sN, int64_t y_sC, int32_t N, int32_t C, int32_t H, int32_t W, void* stream) {
    CCVS_REQUIRE(flow && y, "ccvs_backwarp_ctx: null pointer");
    CtxList l = {};
    const int rc = fill_ctx(l, ctx, "ccvs_backwarp_ctx");
    if (rc != CCVS_OK) return rc;
    CCVS_REQUIRE(N > 0 && N % l.k == 0 && C > 0 && H > 0 && W > 0, "ccvs_backwarp_ctx: bad shape");
    launch_backwarp(l, (long)x_sC, flow, (long)flow_sN, flow_mult, y, (long)y_sN, (long)y_sC, N, C, H, W, stream);
    CCVS_CHECK_LAUNCH("ccvs_backwarp_ctx");
    return CCVS_OK;
}

// ---------------------------------------------------------------------------------------
// backwarp + 1x1 projection in one pass: Matching warps each context feature with the carried flow and immediately
// projects it to max(16, C/4) channels for the cost volume (skip_autoencoder.py:186-190: proj(backwarp(inter, flow))).
// Done as two kernels the warped C-channel tensor is written and read back once (6 GB per level-5 call at BAIR size) only
// to be reduced 4x; here a lane owns one pixel, samples channel after channel (set-up shared by all channels) and keeps
// the CO projected channels in registers:  y[n][o][p] = lrelu(b[o] + sum_c W[o][c] * warp(x)[n][c][p]).
// The weight row of a channel is wave-uniform (scalar loads).  fp32 FMAs: exact products, unlike the split-bf16 convolution.
// ---------------------------------------------------------------------------------------
template <int CO>
__global__ __launch_bounds__(256) void warp_proj_kernel(CtxList ctx, long x_sC, const float* __restrict__ flow, long flow_sN, float mult,
                                                        const float* __restrict__ wt, const float* __restrict__ bias,
                                                        float* __restrict__ y, int Cin, int Cout, int H, int W, int act, GridWalk gw) {
    const int HW = H * W;
    GRID_WALK_BEGIN(gw, bx, by, bz)
    (void)bz;
    const int pix = bx * 256 + threadIdx.x;
    if (pix >= HW) continue;
    const int n = by;
    const int jn = n % ctx.k;
    const float* x = ctx.p[jn] + (long)(n / ctx.k) * ctx.sN[jn];
    const int py = pix / W, px = pix - py * W;
    const float fx = flow[(long)n * flow_sN + pix] * mult, fy = flow[(long)n * flow_sN + HW + pix] * mult;
    const Bilin b = bilin_setup(px, py, fx, fy, H, W);
    float acc[CO];
#pragma unroll
    for (int o = 0; o < CO; ++o) acc[o] = 0.f;
#pragma unroll 2
    for (int c = 0; c < Cin; ++c) {
        const float v = bilin_sample(x + (long)c * x_sC, b);
        const float* wr = wt + (long)c * CO;   // [Cin][CO]: the CO weights of input channel c, the same for every lane
#pragma unroll
        for (int o = 0; o < CO; ++o) acc[o] += wr[o] * v;
    }
    float* yo = y + (long)n * Cout * HW + pix;
#pragma unroll
    for (int o = 0; o < CO; ++o) {
        if (o < Cout) {
            float v = acc[o] + (bias ? bias[o] : 0.f);
            if (act == CCVS_ACT_LRELU) v = lrelu01(v);
            yo[(long)o * HW] = v;
        }
    }
    GRID_WALK_END
}

// Four pixels per lane (W % 4 == 0, CO <= 24: the accumulators are CO x 4 registers): 8-byte tap pairs in, one 16-byte store
// per output channel out.  Per output the sum runs over the input channels in the order of warp_proj_kernel.
template <int CO>
__global__ __launch_bounds__(256) void warp_proj4_kernel(CtxList ctx, long x_sC, const float* __restrict__ flow, long flow_sN, float mult,
                                                         const float* __restrict__ wt, const float* __restrict__ bias,
                                                         float* __restrict__ y, int Cin, int Cout, int H, int W, int act, GridWalk gw) {
    const int HW = H * W;
    GRID_WALK_BEGIN(gw, bx, by, bz)
    (void)bz;
    const int pix = quad_pixel(gw, bx, W);
    if (pix >= HW) continue;
    const int n = by;
    const int jn = n % ctx.k;
    const float* x = ctx.p[jn] + (long)(n / ctx.k) * ctx.sN[jn];
    const int py = pix / W, px = pix - py * W;
    const F32Quad fx = *reinterpret_cast<const F32Quad*>(flow + (long)n * flow_sN + pix);
    const F32Quad fy = *reinterpret_cast<const F32Quad*>(flow + (long)n * flow_sN + HW + pix);
    BilinPair q[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) q[i] = bilin_setup_pair(px + i, py, fx.v[i] * mult, fy.v[i] * mult, H, W);
    float acc[CO][4];
#pragma unroll
    for (int o = 0; o < CO; ++o)
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[o][i] = 0.f;
    // two channels per iteration: the 16 gather loads of both are in flight before the first is used (one channel at a time
    // a thread had 8 loads of 8 bytes outstanding); the sums keep their order (channel c before c + 1)
    int c = 0;
    for (; c + 2 <= Cin; c += 2) {
        const float* pl0 = x + (long)c * x_sC;
        const float* pl1 = pl0 + x_sC;
        F32Pair r00[4], r01[4], r10[4], r11[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            r00[i] = *reinterpret_cast<const F32Pair*>(pl0 + q[i].o0);
            r01[i] = *reinterpret_cast<const F32Pair*>(pl0 + q[i].o1);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            r10[i] = *reinterpret_cast<const F32Pair*>(pl1 + q[i].o0);
            r11[i] = *reinterpret_cast<const F32Pair*>(pl1 + q[i].o1);
        }
        float v0[4], v1[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            v0[i] = ((r00[i].x * q[i].a0 + r00[i].y * q[i].b0) + r01[i].x * q[i].a1) + r01[i].y * q[i].b1;
            v1[i] = ((r10[i].x * q[i].a0 + r10[i].y * q[i].b0) + r11[i].x * q[i].a1) + r11[i].y * q[i].b1;
        }
        const float* wr = wt + (long)c * CO;
#pragma unroll
        for (int o = 0; o < CO; ++o) {
            const float w0 = wr[o];
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[o][i] += w0 * v0[i];
        }
#pragma unroll
        for (int o = 0; o < CO; ++o) {
            const float w1 = wr[CO + o];
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[o][i] += w1 * v1[i];
        }
    }
    for (; c < Cin; ++c) {
        const float* pl = x + (long)c * x_sC;
        float v[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) v[i] = bilin_sample_pair(pl, q[i]);
        const float* wr = wt + (long)c * CO;
#pragma unroll
        for (int o = 0; o < CO; ++o) {
            const float wv = wr[o];
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[o][i] += wv * v[i];
        }
    }
    float* yo = y + (long)n * Cout * HW + pix;
#pragma unroll
    for (int o = 0; o < CO; ++o) {
        if (o < Cout) {
            const float bv = bias ? bias[o] : 0.f;
            F32Quad o4;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                float v = acc[o][i] + bv;
                if (act == CCVS_ACT_LRELU) v = lrelu01(v);
                o4.v[i] = v;
            }
            *reinterpret_cast<F32Quad*>(yo + (long)o * HW) = o4;
        }
    }
    GRID_WALK_END
}

extern "C" int ccvs_backwarp_proj_ctx(const ccvs_ctx_list* ctx, int64_t x_sC, const float* flow, int64_t flow_sN, float flow_mult,
                                      const float* w_t, const float* bias, float* y, int32_t N, int32_t Cin, int32_t Cout, int32_t CoutPad,
                                      int32_t H, int32_t W, int32_t act, void* stream) {
    CCVS_REQUIRE(flow && w_t && y, "ccvs_backwarp_proj_ctx: null pointer");
    CtxList l = {};
    const int rc = fill_ctx(l, ctx, "ccvs_backwarp_proj_ctx");
    if (rc != CCVS_OK) return rc;
    CCVS_REQUIRE(N > 0 && N % l.k == 0 && Cin > 0 && Cout > 0 && H > 0 && W > 0, "ccvs_backwarp_proj_ctx: bad shape");
    CCVS_REQUIRE(CoutPad >= Cout && (CoutPad == 16 || CoutPad == 24 || CoutPad == 48 || CoutPad == 96),
                 "ccvs_backwarp_proj_ctx: %d output channels (padded %d) unsupported (16, 24, 48, 96)", Cout, CoutPad);
    const GridWalk gw = grid_walk(cdiv(H * W, 256), N, 1);
    const dim3 grid(limited_grid(gw.total, stream, 4));
    hipStream_t st = (hipStream_t)stream;
#define WP_LAUNCH(CO)                                                                                                               \
    hipLaunchKernelGGL((warp_proj_kernel<CO>), grid, dim3(256), 0, st, l, (long)x_sC, flow, (long)flow_sN, flow_mult, w_t, bias, y, Cin, Cout, \
                       H, W, act, gw)
    if (W % 4 == 0 && CoutPad <= 24) {
        GridWalk gw4 = grid_walk(cdiv(H * W / 4, 256), N, 1);
        gw4.tiled = warp_tiled(H, W);
        const dim3 grid4(limited_grid(gw4.total, stream, 4));
        if (CoutPad == 16)
            hipLaunchKernelGGL((warp_proj4_kernel<16>), grid4, dim3(256), 0, st, l, (long)x_sC, flow, (long)flow_sN, flow_mult, w_t, bias, y, Cin, Cout, H, W, act, gw4);
        else
            hipLaunchKernelGGL((warp_proj4_kernel<24>), grid4, dim3(256), 0, st, l, (long)x_sC, flow, (long)flow_sN, flow_mult, w_t, bias, y, Cin, Cout, H, W, act, gw4);
    } else if (CoutPad == 16) WP_LAUNCH(16);
    else if (CoutPad == 24) WP_LAUNCH(24);
    else if (CoutPad == 48) WP_LAUNCH(48);
    else WP_LAUNCH(96);
#undef WP_LAUNCH
    CCVS_CHECK_LAUNCH("ccvs_backwarp_proj_ctx");
    return CCVS_OK;
}

// ---------------------------------------------------------------------------------------
// InterBlock tail (skip_autoencoder.py:254-264): warp each of the k context features with
// its final flow, fuse with confidences 1 - sigmoid(occ_k) + eps, blend into the decoder
// feature with sigmoid of the fused occlusion.  One lane per pixel, WARP_CCH channels per
// thread; the per-k sample set-up is recomputed per channel chunk (3 floats per k).
// ---------------------------------------------------------------------------------------
__device__ __forceinline__ float sigmoidf_(float v) { return 1.f / (1.f + expf(-v)); }

__global__ __launch_bounds__(256) void warp_fuse_blend_kernel(float* __restrict__ dec, long dec_sN, long dec_sC,
                                                              CtxList ctx, const float* __restrict__ flows,
                                                              long flows_sN, const float* __restrict__ occs, long occs_sN,
                                                              float mult, int k, int C, int H, int W, GridWalk gw) {
    const int HW = H * W;
    GRID_WALK_BEGIN(gw, bx, by, bz)
    const int pix = bx * 256 + threadIdx.x;
    if (pix >= HW) continue;
    const int n = bz, c0 = by * WARP_CCH;
    const int py = pix / W, px = pix - py * W;
    const int cn = min(WARP_CCH, C - c0);
    float acc[WARP_CCH];
#pragma unroll
    for (int j = 0; j < WARP_CCH; ++j) acc[j] = 0.f;
    float sum_conf = 0.f, sum_occ = 0.f;
    for (int kk = 0; kk < k; ++kk) {
        const long nk = (long)n * k + kk;
        const float fx = flows[nk * flows_sN + pix] * mult, fy = flows[nk * flows_sN + HW + pix] * mult;
        const float oc = occs[nk * occs_sN + pix];
        const float conf = (k > 1) ? (1.f - sigmoidf_(oc)) + 1e-6f : 1.f;
        sum_conf += conf;
        sum_occ += oc * conf;
        const Bilin b = bilin_setup(px, py, fx, fy, H, W);
        const float* base = ctx.p[kk] + (long)n * ctx.sN[kk] + (long)c0 * HW;
#pragma unroll
        for (int j = 0; j < WARP_CCH; ++j)
            if (j < cn) acc[j] += conf * bilin_sample(base + (long)j * HW, b);
    }
    const float occ = (k > 1) ? sum_occ / sum_conf : sum_occ;
    const float m = sigmoidf_(occ);
#pragma unroll
    for (int j = 0; j < WARP_CCH; ++j) {
        if (j < cn) {
            float* d = dec + (long)n * dec_sN + (long)(c0 + j) * dec_sC + pix;
            const float wv = (k > 1) ? acc[j] / sum_conf : acc[j];
            *d = m * (*d) + (1.f - m) * wv;
        }
    }
    GRID_WALK_END
}

template <int CCH, bool PREF>   // channels per thread; flows of the next context requested one context ahead
__global__ __launch_bounds__(256) void warp_fuse_blend4_kernel(float* __restrict__ dec, long dec_sN, long dec_sC, CtxList ctx,
                                                               const float* __restrict__ flows, long flows_sN, const float* __restrict__ occs,
                                                               long occs_sN, float mult, int k, int C, int H, int W, GridWalk gw) {
    const int HW = H * W;
    GRID_WALK_BEGIN(gw, bx, by, bz)
    const int pix = quad_pixel(gw, bx, W);
    if (pix >= HW) continue;
    const int n = bz, c0 = by * CCH;
    const int py = pix / W, px = pix - py * W;
    const int cn = min(CCH, C - c0);
    float acc[CCH][4];
#pragma unroll
    for (int j = 0; j < CCH; ++j)
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[j][i] = 0.f;
    float sum_conf[4] = {0.f, 0.f, 0.f, 0.f}, sum_occ[4] = {0.f, 0.f, 0.f, 0.f};
    // flows / occlusions of context kk + 1 are requested before context kk is sampled: fetched at the top of their own iteration
    // they put a second memory round trip in front of every context's gathers (k = 15 of them per thread, one after the other)
    F32Quad fx_n = *reinterpret_cast<const F32Quad*>(flows + (long)n * k * flows_sN + pix);
    F32Quad fy_n = *reinterpret_cast<const F32Quad*>(flows + (long)n * k * flows_sN + HW + pix);
    F32Quad oc_n = *reinterpret_cast<const F32Quad*>(occs + (long)n * k * occs_sN + pix);
    for (int kk = 0; kk < k; ++kk) {
        F32Quad fx = fx_n, fy = fy_n, oc = oc_n;
        if (!PREF && kk > 0) {
            const long nk = (long)n * k + kk;
            fx = *reinterpret_cast<const F32Quad*>(flows + nk * flows_sN + pix);
            fy = *reinterpret_cast<const F32Quad*>(flows + nk * flows_sN + HW + pix);
            oc = *reinterpret_cast<const F32Quad*>(occs + nk * occs_sN + pix);
        }
        if (PREF && kk + 1 < k) {
            const long nk1 = (long)n * k + kk + 1;
            fx_n = *reinterpret_cast<const F32Quad*>(flows + nk1 * flows_sN + pix);
            fy_n = *reinterpret_cast<const F32Quad*>(flows + nk1 * flows_sN + HW + pix);
            oc_n = *reinterpret_cast<const F32Quad*>(occs + nk1 * occs_sN + pix);
        }
        float conf[4];
        BilinPair q[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            conf[i] = (k > 1) ? (1.f - sigmoidf_(oc.v[i])) + 1e-6f : 1.f;
            sum_conf[i] += conf[i];
            sum_occ[i] += oc.v[i] * conf[i];
            q[i] = bilin_setup_pair(px + i, py, fx.v[i] * mult, fy.v[i] * mult, H, W);
        }
        const float* base = ctx.p[kk] + (long)n * ctx.sN[kk] + (long)c0 * HW;
#pragma unroll
        for (int j = 0; j < CCH; ++j)
            if (j < cn) {
#pragma unroll
                for (int i = 0; i < 4; ++i) acc[j][i] += conf[i] * bilin_sample_pair(base + (long)j * HW, q[i]);
            }
    }
    float m[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) m[i] = sigmoidf_((k > 1) ? sum_occ[i] / sum_conf[i] : sum_occ[i]);
#pragma unroll
    for (int j = 0; j < CCH; ++j) {
        if (j < cn) {
            F32Quad* d = reinterpret_cast<F32Quad*>(dec + (long)n * dec_sN + (long)(c0 + j) * dec_sC + pix);
            F32Quad dv = *d;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const float wv = (k > 1) ? acc[j][i] / sum_conf[i] : acc[j][i];
                dv.v[i] = m[i] * dv.v[i] + (1.f - m[i]) * wv;
            }
            *d = dv;
        }
    }
    GRID_WALK_END
}

static void launch_warp_fuse_blend(float* dec, long dec_sN, long dec_sC, const CtxList& l, const float* flows, long flows_sN, const float* occs,
                                   long occs_sN, float mult, int N, int k, int C, int H, int W, void* stream) {
    if (W % 4 == 0) {
        static const int cch = getenv("CCVS_FUSE_CCH") ? atoi(getenv("CCVS_FUSE_CCH")) : 8;
        static const int pref = getenv("CCVS_FUSE_PREF") ? atoi(getenv("CCVS_FUSE_PREF")) : 1;
        GridWalk gw = grid_walk(cdiv(H * W / 4, 256), cdiv(C, cch == 4 ? 4 : 8), N);
        gw.tiled = warp_tiled(H, W);
#define WFB_LAUNCH(CCHv, PREFv)                                                                                                         \
    hipLaunchKernelGGL((warp_fuse_blend4_kernel<CCHv, PREFv>), dim3(limited_grid(gw.total, stream, 8)), dim3(256), 0, (hipStream_t)stream, dec, \
                       dec_sN, dec_sC, l, flows, flows_sN, occs, occs_sN, mult, k, C, H, W, gw)
        if (cch == 4) { if (pref) WFB_LAUNCH(4, true); else WFB_LAUNCH(4, false); }
        else { if (pref) WFB_LAUNCH(8, true); else WFB_LAUNCH(8, false); }
#undef WFB_LAUNCH
    } else {
        const GridWalk gw = grid_walk(cdiv(H * W, 256), cdiv(C, WARP_CCH), N);
        hipLaunchKernelGGL(warp_fuse_blend_kernel, dim3(limited_grid(gw.total, stream, 8)), dim3(256), 0, (hipStream_t)stream, dec, dec_sN, dec_sC, l,
                           flows, flows_sN, occs, occs_sN, mult, k, C, H, W, gw);
    }
}

extern "C" int ccvs_warp_fuse_blend(float* dec, int64_t dec_sN, int64_t dec_sC, const float* ctx, const float* flows, int64_t flows_sN,
                                    const float* occs, int64_t occs_sN, float flow_mult, int32_t N, int32_t k, int32_t C, int32_t H,
                                    int32_t W, void* stream) {
    CCVS_REQUIRE(dec && ctx && flows && occs, "ccvs_warp_fuse_blend: null pointer");
    CCVS_REQUIRE(N > 0 && k > 0 && C > 0 && H > 0 && W > 0, "ccvs_warp_fuse_blend: bad shape");
    CCVS_REQUIRE(k <= CCVS_MAX_CTX, "ccvs_warp_fuse_blend: at most %d contexts", CCVS_MAX_CTX);
    CtxList l = {};
    l.k = k;
    for (int j = 0; j < k; ++j) { l.p[j] = ctx + (long)j * C * H * W; l.sN[j] = (long)k * C * H * W; }
    launch_warp_fuse_blend(dec, (long)dec_sN, (long)dec_sC, l, flows, (long)flows_sN, occs, (long)occs_sN, flow_mult, N, k, C, H, W, stream);
    CCVS_CHECK_LAUNCH("ccvs_warp_fuse_blend");
    return CCVS_OK;
}

extern "C" int ccvs_warp_fuse_blend_ctx(float* dec, int64_t dec_sN, int64_t dec_sC, const ccvs_ctx_list* ctx, const float* flows,
                                        int64_t flows_sN, const float* occs, int64_t occs_sN, float flow_mult, int32_t N, int32_t C,
                                        int32_t H, int32_t W, void* stream) {
    CCVS_REQUIRE(dec && flows && occs, "ccvs_warp_fuse_blend_ctx: null pointer");
    CtxList l = {};
    const int rc = fill_ctx(l, ctx, "ccvs_warp_fuse_blend_ctx");
    if (rc != CCVS_OK) return rc;
    CCVS_REQUIRE(N > 0 && C > 0 && H > 0 && W > 0, "ccvs_warp_fuse_blend_ctx: bad shape");
    launch_warp_fuse_blend(dec, (long)dec_sN, (long)dec_sC, l, flows, (long)flows_sN, occs, (long)occs_sN, flow_mult, N, l.k, C, H, W, stream);
    CCVS_CHECK_LAUNCH("ccvs_warp_fuse_blend_ctx");
    return CCVS_OK;
}


// ---------------------------------------------------------------------------------------
// Second half of the flow / occlusion heads (skip_autoencoder.py:176-177,204-205,225-226).
// A k x k convolution with only 3 outputs would use 3 of the 32 rows of an MFMA tile.  The heads
// are therefore run as a k x 1 convolution with 3k outputs, T[kx*3+co] = sum_{c,ky} W[co][c][ky][kx] * X
// on a map widened by the horizontal padding (one MFMA row per (kx, co): 27 of 32 rows busy for the
// 9x9 heads), and this kernel applies the horizontal taps:
//   y[n][co][yy][x] (+)= bias[co] + sum_kx T[n][kx*3+co][yy][x + kx]          (T width = W + k - 1)
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void tap_shift_add_kernel(const float* __restrict__ t, const float* __restrict__ bias,
                                                            float* __restrict__ y, long y_sN, long total, int k, int H, int W,
                                                            int accumulate, int vertical) {
    // horizontal: T is [N,3k,H,W+k-1] and tap kx reads column x+kx; vertical: T is [N,3k,H+k-1,W], tap ky reads row yy+ky
    const int Wt = vertical ? W : W + k - 1, Ht = vertical ? H + k - 1 : H;
    const long tap = (long)3 * Ht * Wt + (vertical ? Wt : 1);  // next tap: 3 maps further, one row / column further
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int x = (int)(i % W);
        long r = i / W;
        const int yy = (int)(r % H);
        r /= H;
        const int co = (int)(r % 3);
        const long n = r / 3;
        const float* tp = t + ((n * 3 * k + co) * Ht + yy) * (long)Wt + x;
        float acc = bias ? bias[co] : 0.f;
        for (int kk = 0; kk < k; ++kk) acc += tp[kk * tap];
        float* dst = y + n * y_sN + ((long)co * H + yy) * W + x;
        *dst = accumulate ? *dst + acc : acc;
    }
}

// W % 4 == 0: a lane owns four consecutive pixels of a row -- every tap is one 16-byte load (dword-aligned in the horizontal
// form), the result one 16-byte store; per pixel the same sum in the same order as the one-pixel form.
__global__ __launch_bounds__(256) void tap_shift_add4_kernel(const float* __restrict__ t, const float* __restrict__ bias,
                                                             float* __restrict__ y, long y_sN, long total4, int k, int H, int W,
                                                             int accumulate, int vertical) {
    const int Wt = vertical ? W : W + k - 1, Ht = vertical ? H + k - 1 : H;
    const long tap = (long)3 * Ht * Wt + (vertical ? Wt : 1);
    const int Wq = W >> 2;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total4; i += (long)gridDim.x * 256) {
        const int x = (int)(i % Wq) * 4;
        long r = i / Wq;
        const int yy = (int)(r % H);
        r /= H;
        const int co = (int)(r % 3);
        const long n = r / 3;
        const float* tp = t + ((n * 3 * k + co) * Ht + yy) * (long)Wt + x;
        const float b = bias ? bias[co] : 0.f;
        float acc[4] = {b, b, b, b};
        for (int kk = 0; kk < k; ++kk) {
            const F32Quad q = *reinterpret_cast<const F32Quad*>(tp + kk * tap);
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[j] += q.v[j];
        }
        F32Quad* dst = reinterpret_cast<F32Quad*>(y + n * y_sN + ((long)co * H + yy) * W + x);
        F32Quad o;
        if (accumulate) {
            const F32Quad d = *dst;
#pragma unroll
            for (int j = 0; j < 4; ++j) o.v[j] = d.v[j] + acc[j];
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j) o.v[j] = acc[j];
        }
        *dst = o;
    }
}

extern "C" int ccvs_tap_shift_add(const float* t, const float* bias, float* y, int64_t y_sN, int32_t N, int32_t k, int32_t H, int32_t W,
                                  int32_t accumulate, int32_t vertical, void* stream) {
    CCVS_REQUIRE(t && y, "ccvs_tap_shift_add: null pointer");
    CCVS_REQUIRE(N > 0 && k >= 1 && k <= 9 && H > 0 && W > 0, "ccvs_tap_shift_add: bad shape");
    const long total = (long)N * 3 * H * W;
    if (W % 4 == 0 && y_sN % 4 == 0) {
        const long total4 = total / 4;
        const unsigned blocks4 = limited_grid(cdiv64(total4, 256) < 65536 * 16 ? cdiv64(total4, 256) : 65536 * 16, stream, 8);
        hipLaunchKernelGGL(tap_shift_add4_kernel, dim3(blocks4), dim3(256), 0, (hipStream_t)stream, t, bias, y, (long)y_sN, total4, k, H, W, accumulate,
                           vertical ? 1 : 0);
        CCVS_CHECK_LAUNCH("ccvs_tap_shift_add");
        return CCVS_OK;
    }
    const unsigned blocks = limited_grid(cdiv64(total, 256) < 65536 * 16 ? cdiv64(total, 256) : 65536 * 16, stream, 8);
    hipLaunchKernelGGL(tap_shift_add_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, t, bias, y, (long)y_sN, total, k, H, W, accumulate,
                       vertical ? 1 : 0);
    CCVS_CHECK_LAUNCH("ccvs_tap_shift_add");
    return CCVS_OK;
}
