// Evaluation metrics on the clips the path produces (tools/pytorch_metrics/metrics.py:15-25).
//
//   get_psnr -> piq.psnr(x, y, data_range=1., reduction='mean')                  (piq 0.5.4, env.yml:234)
//   get_ssim -> skimage.metrics.structural_similarity(x[i, c], y[i, c]) per 2-D plane, defaults of scikit-image 0.17.2
//               (env.yml:164): 7 x 7 uniform window, sample covariance (49 / 48), K1 = 0.01, K2 = 0.03, data_range taken from
//               the dtype (float: 2), computed in float64, mean over the windows that lie inside the plane.
//
// Both are HBM-bound reductions: every pixel is read once (the SSIM tile re-reads a 3-pixel halo through L2).
#include "common.h"

// ---- PSNR -------------------------------------------------------------------------------------
// one workgroup per image: sum (x - y)^2 over C*H*W in float64, score = -10 log10(mse / R^2 + 1e-8)
__global__ __launch_bounds__(1024) void psnr_kernel(const float* __restrict__ x, const float* __restrict__ y, float* __restrict__ out, long per_image,
                                                     double inv_range) {
    const long base = (long)blockIdx.x * per_image;
    double acc = 0.0;
    const long n4 = ((reinterpret_cast<uintptr_t>(x + base) | reinterpret_cast<uintptr_t>(y + base)) & 15) == 0 ? per_image / 4 : 0;
    const float4* x4 = reinterpret_cast<const float4*>(x + base);
    const float4* y4 = reinterpret_cast<const float4*>(y + base);
    for (long i = threadIdx.x; i < n4; i += 1024) {
        const float4 a = x4[i], b = y4[i];
        const double d0 = ((double)a.x - (double)b.x) * inv_range, d1 = ((double)a.y - (double)b.y) * inv_range;
        const double d2 = ((double)a.z - (double)b.z) * inv_range, d3 = ((double)a.w - (double)b.w) * inv_range;
        acc += d0 * d0 + d1 * d1 + d2 * d2 + d3 * d3;
    }
    for (long i = 4 * n4 + threadIdx.x; i < per_image; i += 1024) {
        const double d = ((double)x[base + i] - (double)y[base + i]) * inv_range;
        acc += d * d;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o);
    __shared__ double part[16];
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        double s = 0.0;
        for (int w = 0; w < 16; ++w) s += part[w];   // fixed order: the same bits on every run
        out[blockIdx.x] = (float)(-10.0 * log10(s / (double)per_image + 1e-8));
    }
}

extern "C" int ccvs_psnr(const float* x, const float* y, float* out, int64_t N, int64_t per_image, float data_range, void* stream) {
    CCVS_REQUIRE(x && y && out, "ccvs_psnr: null pointer");
    CCVS_REQUIRE(N > 0 && N < 2147483647L && per_image > 0 && data_range > 0.f, "ccvs_psnr: bad arguments");
    hipLaunchKernelGGL(psnr_kernel, dim3((unsigned)N), dim3(1024), 0, (hipStream_t)stream, x, y, out, (long)per_image, 1.0 / (double)data_range);
    CCVS_CHECK_LAUNCH("ccvs_psnr");
    return CCVS_OK;
}

// ---- SSIM -------------------------------------------------------------------------------------
// A workgroup owns a 32 x 32 block of window CENTRES of one plane: the 38 x 38 pixels under them go to LDS, the five window sums
// (x, y, xx, yy, xy) are formed separably in float64 -- 7 columns, then 7 rows -- and the block's sum of S is written to
// part[plane][block]; a second kernel adds the blocks of a plane in index order (no atomics: run-to-run identical).
#define SS_T 32
#define SS_W 7
#define SS_H (SS_T + SS_W - 1)

__global__ __launch_bounds__(256) void ssim_tile_kernel(const float* __restrict__ x, const float* __restrict__ y, double* __restrict__ part, int H, int W,
                                                         int tiles_x, int tiles_y, double c1, double c2) {
#pragma clang fp contract(off)   // numpy's arithmetic: with fused multiply-adds SSIM(x, x) is 1 - 1e-16 instead of 1
    __shared__ float sx[SS_H][SS_H + 1], sy[SS_H][SS_H + 1];
    __shared__ double row[5][SS_H][SS_T + 1];
    __shared__ double wsum[4];
    const int plane = blockIdx.y, tile = blockIdx.x;
    const int ty = tile / tiles_x, tx = tile - ty * tiles_x;
    const int y0 = ty * SS_T, x0 = tx * SS_T;            // first window's top-left pixel = first centre - 3
    const int vh = H - (SS_W - 1), vw = W - (SS_W - 1);  // window positions of the plane
    const float* xp = x + (long)plane * H * W;
    const float* yp = y + (long)plane * H * W;
    for (int e = threadIdx.x; e < SS_H * SS_H; e += 256) {
        const int r = e / SS_H, c = e - r * SS_H;
        const int gy = y0 + r, gx = x0 + c;
        const bool in = gy < H && gx < W;
        sx[r][c] = in ? xp[(long)gy * W + gx] : 0.f;
        sy[r][c] = in ? yp[(long)gy * W + gx] : 0.f;
    }
    __syncthreads();
    for (int e = threadIdx.x; e < SS_H * SS_T; e += 256) {
        const int r = e / SS_T, c = e - r * SS_T;
        double a = 0, b = 0, aa = 0, bb = 0, ab = 0;
#pragma unroll
        for (int k = 0; k < SS_W; ++k) {
            const double u = sx[r][c + k], v = sy[r][c + k];
            a += u; b += v; aa += u * u; bb += v * v; ab += u * v;
        }
        row[0][r][c] = a; row[1][r][c] = b; row[2][r][c] = aa; row[3][r][c] = bb; row[4][r][c] = ab;
    }
    __syncthreads();
    double acc = 0.0;
    for (int e = threadIdx.x; e < SS_T * SS_T; e += 256) {
        const int r = e / SS_T, c = e - r * SS_T;
        if (y0 + r >= vh || x0 + c >= vw) continue;
        double s[5];
#pragma unroll
        for (int q = 0; q < 5; ++q) {
            double t = 0;
#pragma unroll
            for (int k = 0; k < SS_W; ++k) t += row[q][r + k][c];
            s[q] = t / 49.0;
        }
        const double ux = s[0], uy = s[1];
        const double cov = 49.0 / 48.0;   // use_sample_covariance
        const double vx = cov * (s[2] - ux * ux), vy = cov * (s[3] - uy * uy), vxy = cov * (s[4] - ux * uy);
        const double a1 = 2 * ux * uy + c1, a2 = 2 * vxy + c2, b1 = ux * ux + uy * uy + c1, b2 = vx + vy + c2;
        acc += (a1 * a2) / (b1 * b2);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o);
    if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) part[(long)plane * tiles_x * tiles_y + tile] = (wsum[0] + wsum[1]) + (wsum[2] + wsum[3]);
}

__global__ __launch_bounds__(64) void ssim_finish_kernel(const double* __restrict__ part, double* __restrict__ out, int tiles, double count) {
    if (threadIdx.x != 0) return;
    const double* p = part + (long)blockIdx.x * tiles;
    double s = 0.0;
    for (int t = 0; t < tiles; ++t) s += p[t];
    out[blockIdx.x] = s / count;   // a division like numpy's mean: n ones give exactly 1
}

extern "C" int64_t ccvs_ssim_workspace_bytes(int64_t planes, int32_t H, int32_t W) {
    if (planes <= 0 || H < SS_W || W < SS_W) return 0;
    return planes * cdiv64(H - (SS_W - 1), SS_T) * cdiv64(W - (SS_W - 1), SS_T) * (int64_t)sizeof(double);
}

extern "C" int ccvs_ssim(const float* x, const float* y, double* out, void* workspace, int64_t planes, int32_t H, int32_t W, double data_range,
                         void* stream) {
    CCVS_REQUIRE(H >= SS_W && W >= SS_W, "ccvs_ssim: the 7 x 7 window exceeds the %d x %d plane", H, W);
    CCVS_REQUIRE(x && y && out && workspace, "ccvs_ssim: null pointer");
    CCVS_REQUIRE(planes > 0 && planes <= 65535 && data_range > 0, "ccvs_ssim: 1 .. 65535 planes per call, data_range > 0");
    const int tiles_y = (int)cdiv64(H - (SS_W - 1), SS_T), tiles_x = (int)cdiv64(W - (SS_W - 1), SS_T);
    const double c1 = (0.01 * data_range) * (0.01 * data_range), c2 = (0.03 * data_range) * (0.03 * data_range);
    hipLaunchKernelGGL(ssim_tile_kernel, dim3(tiles_x * tiles_y, (unsigned)planes), dim3(256), 0, (hipStream_t)stream, x, y, (double*)workspace, H, W,
                       tiles_x, tiles_y, c1, c2);
    CCVS_CHECK_LAUNCH("ccvs_ssim");
    hipLaunchKernelGGL(ssim_finish_kernel, dim3((unsigned)planes), dim3(64), 0, (hipStream_t)stream, (const double*)workspace, out, tiles_x * tiles_y,
                       (double)(H - (SS_W - 1)) * (double)(W - (SS_W - 1)));
    CCVS_CHECK_LAUNCH("ccvs_ssim");
    return CCVS_OK;
}

// ---- bilinear resize (metrics.py:115-124 `upscale`: F.interpolate(videos, size, mode='bilinear'), align_corners=False) --------
// torch's formulation: src = max(scale * (dst + 0.5) - 0.5, 0) with scale = in / out in fp32, the two rows blended after the
// two columns.
__global__ __launch_bounds__(256) void resize_bilinear_kernel(const float* __restrict__ x, float* __restrict__ out, long planes, int H, int W, int OH,
                                                               int OW, float sh, float sw) {
#pragma clang fp contract(off)
    const long total = planes * OH * OW;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int ox = (int)(i % OW);
        const long r = i / OW;
        const int oy = (int)(r % OH);
        const long pl = r / OH;
        const float fy = fmaxf(sh * ((float)oy + 0.5f) - 0.5f, 0.f), fx = fmaxf(sw * ((float)ox + 0.5f) - 0.5f, 0.f);
        const int y0 = (int)fy, x0 = (int)fx;
        const int y1 = y0 + (y0 < H - 1 ? 1 : 0), x1 = x0 + (x0 < W - 1 ? 1 : 0);
        const float ly = fy - (float)y0, lx = fx - (float)x0, hy = 1.f - ly, hx = 1.f - lx;
        const float* p = x + pl * H * W;
        out[i] = hy * (hx * p[(long)y0 * W + x0] + lx * p[(long)y0 * W + x1]) + ly * (hx * p[(long)y1 * W + x0] + lx * p[(long)y1 * W + x1]);
    }
}

extern "C" int ccvs_resize_bilinear(const float* x, float* out, int64_t planes, int32_t H, int32_t W, int32_t OH, int32_t OW, void* stream) {
    CCVS_REQUIRE(x && out, "ccvs_resize_bilinear: null pointer");
    CCVS_REQUIRE(planes > 0 && H > 0 && W > 0 && OH > 0 && OW > 0, "ccvs_resize_bilinear: bad arguments");
    const long total = (long)planes * OH * OW;
    const long blocks = cdiv64(total, 256) < 1048576 ? cdiv64(total, 256) : 1048576;
    hipLaunchKernelGGL(resize_bilinear_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, x, out, (long)planes, H, W, OH, OW,
                       (float)H / (float)OH, (float)W / (float)OW);
    CCVS_CHECK_LAUNCH("ccvs_resize_bilinear");
    return CCVS_OK;
}
