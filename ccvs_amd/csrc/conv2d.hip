// Direct convolution as an implicit GEMM on the fp32 matrix cores (v_mfma_f32_32x32x2_f32).
//
// Replaces F.conv2d / F.conv_transpose2d of EqualConv2d.forward
// (reference models/skip_vid_generator/models/skip_autoencoder.py:53-59).
//
// Mapping (NCHW fp32, exact-f32 MFMA = a k-ordered fmaf chain):
//   GEMM  D[cout][pixel] += W[cout][k] * X[k][pixel],  k = (input channel, tap)
//   A operand = packed weights  (lane l: cout = l&31, k-slot = l>>5)
//   B operand = input pixels    (lane l: pixel = l&31, k-slot = l>>5)
//   D         = 32 couts x 32 pixels; a lane owns one pixel column -> for a fixed
//               accumulator register 32 lanes store 32 consecutive x: 128-B coalesced.
// A 256-thread workgroup (4 waves, one per SIMD) owns a TH x TW (=256) pixel tile and
// NT = 32*MB output channels.  Per chunk of CC input channels the input halo tile is
// staged once in LDS and reused by every tap and every cout; the weights of one tap row
// are staged beside it.  Zero padding, channel tails and the stride-2 transposed
// convolution (four output-parity classes, each a small dense conv on the input grid)
// all reduce to the same tap-table loop.
#include "common.h"
#include "conv_common.h"

#define CONV_CC 8     // input channels per LDS chunk (even: the MFMA consumes k in pairs)
#define CONV_MAX_E 5  // max halo-tile elements per thread per channel (256*5 >= 17*65)

template <int TW, int MB>
__global__ __launch_bounds__(256) void conv2d_mfma_kernel(ConvK p) {
    constexpr int TH = 256 / TW;
    constexpr int NT = 32 * MB;
    extern __shared__ __attribute__((aligned(16))) float smem[];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int ty = blockIdx.x / p.tiles_x, tx = blockIdx.x - ty * p.tiles_x;
    const int n0 = blockIdx.y * NT;
    int n = blockIdx.z, cls = 0;
    if (p.transposed) { cls = n & 3; n >>= 2; }
    const AxisTaps ay = axis_taps(p.kh, p.stride, p.pad, p.transposed, cls >> 1, p.Hout);
    const AxisTaps ax = axis_taps(p.kw, p.stride, p.pad, p.transposed, cls & 1, p.Wout);
    if (ty * TH >= ay.V || tx * TW >= ax.V) return;

    const int IH = (TH - 1) * ay.s + ay.ext + 1;
    const int IW = (TW - 1) * ax.s + ax.ext + 1;
    const int plane = IH * IW;
    const int iy0 = ty * TH * ay.s + ay.lo, ix0 = tx * TW * ax.s + ax.lo;
    float* in_tile = smem;                                  // [CC][plane]
    float* w_tile = smem + ((CONV_CC * plane + 3) & ~3);    // [ntx][CC][NT]

    // halo-tile element -> global offset inside a channel plane (same for every chunk)
    int off[CONV_MAX_E];
#pragma unroll
    for (int j = 0; j < CONV_MAX_E; ++j) {
        const int e = tid + 256 * j;
        off[j] = -2;
        if (e < plane) {
            const int r = e / IW, c = e - r * IW;
            const int gy = iy0 + r, gx = ix0 + c;
            off[j] = (gy >= 0 && gy < p.Hin && gx >= 0 && gx < p.Win) ? gy * p.Win + gx : -1;
        }
    }

    // this wave's two 32-pixel groups
    int bofs[2];
#pragma unroll
    for (int pp = 0; pp < 2; ++pp) {
        const int pj = (wave * 2 + pp) * 32 + (lane & 31);
        const int prow = pj / TW, pcol = pj - prow * TW;
        bofs[pp] = prow * ay.s * IW + pcol * ax.s;
    }
    const int khalf = lane >> 5;

    f32x16 acc[MB][2];
#pragma unroll
    for (int m = 0; m < MB; ++m)
#pragma unroll
        for (int pp = 0; pp < 2; ++pp)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[m][pp][r] = 0.f;

    const float* xn = p.x + (long)n * p.in_sN;
    const int wrow4 = ax.nt * CONV_CC * NT / 4;

    for (int c0 = 0; c0 < p.Cin; c0 += CONV_CC) {
        __syncthreads();  // every wave is done with the previous chunk's tiles
#pragma unroll
        for (int j = 0; j < CONV_MAX_E; ++j) {
            if (off[j] != -2) {
                const int e = tid + 256 * j;
                // unconditional loads from a clamped address, zeroed afterwards (a predicated load in an
                // unrolled loop makes hipcc branch + wait per element)
                const bool inside = off[j] >= 0;
                const int o = inside ? off[j] : 0;
#pragma unroll
                for (int c = 0; c < CONV_CC; ++c) {
                    const float t = xn[(long)min(c0 + c, p.Cin - 1) * p.in_sC + o];
                    in_tile[c * plane + e] = (inside && c0 + c < p.Cin) ? t : 0.f;
                }
            }
        }
        for (int a = 0; a < ay.nt; ++a) {
            if (a > 0) __syncthreads();  // previous tap row's weights consumed
            const int wy = ay.w0 + a * ay.dw;
            for (int i = tid; i < wrow4; i += 256) {
                const int idx = i * 4;
                const int b = idx / (CONV_CC * NT), rem = idx - b * (CONV_CC * NT);
                const int c = rem / NT, nn = rem - c * NT;
                const int tap = wy * p.kw + (ax.w0 + b * ax.dw);
                float4 v = *reinterpret_cast<const float4*>(p.w + ((long)tap * p.Cin + min(c0 + c, p.Cin - 1)) * p.CoutPad + n0 + nn);
                if (c0 + c >= p.Cin) v = make_float4(0.f, 0.f, 0.f, 0.f);
                *reinterpret_cast<float4*>(w_tile + idx) = v;
            }
            __syncthreads();
            const int dyl = (ay.d0 + a * ay.dd - ay.lo) * IW;
            for (int b = 0; b < ax.nt; ++b) {
                const int dl = dyl + (ax.d0 + b * ax.dd - ax.lo);
                const float* wt = w_tile + b * (CONV_CC * NT) + (lane & 31);
#pragma unroll
                for (int kk = 0; kk < CONV_CC / 2; ++kk) {
                    const int c = 2 * kk + khalf;
                    float av[MB], bv[2];
#pragma unroll
                    for (int m = 0; m < MB; ++m) av[m] = wt[c * NT + m * 32];
#pragma unroll
                    for (int pp = 0; pp < 2; ++pp) bv[pp] = in_tile[c * plane + bofs[pp] + dl];
#pragma unroll
                    for (int m = 0; m < MB; ++m)
#pragma unroll
                        for (int pp = 0; pp < 2; ++pp)
                            acc[m][pp] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[m], bv[pp], acc[m][pp], 0, 0, 0);
                }
            }
        }
    }

    // epilogue: bias, LeakyReLU(0.1), residual, scale, optional accumulate into y
#pragma unroll
    for (int pp = 0; pp < 2; ++pp) {
        const int pj = (wave * 2 + pp) * 32 + (lane & 31);
        const int prow = pj / TW, pcol = pj - prow * TW;
        const int vy = ty * TH + prow, vx = tx * TW + pcol;
        if (vy >= ay.V || vx >= ax.V) continue;
        const int oy = vy * ay.os + ay.oo, ox = vx * ax.os + ax.oo;
        const long opix = (long)oy * p.Wout + ox;
#pragma unroll
        for (int m = 0; m < MB; ++m) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int co = n0 + m * 32 + (r & 3) + 8 * (r >> 2) + 4 * khalf;
                if (co < p.Cout) {
                    float v = acc[m][pp][r];
                    if (p.pre) v += p.pre[(long)(n / p.pre_div) * p.pre_sN + (long)co * p.pre_sC + opix];
                    if (p.bias) v += p.bias[co];
                    if (p.act == CCVS_ACT_LRELU) v = lrelu01(v);
                    if (p.res) v += p.res[(long)n * p.res_sN + (long)co * p.res_sC + opix];
                    v *= p.out_scale;
                    float* dst = p.y + (long)n * p.out_sN + (long)co * p.out_sC + opix;
                    if (p.accumulate) v += *dst;
                    *dst = v;
                }
            }
        }
    }
}

template <int TW, int MB>
static int launch_conv(const ConvK& k, int halo_h, int halo_w, int ntx_max, int gz, hipStream_t st) {
    constexpr int NT = 32 * MB;
    const int plane = halo_h * halo_w;
    if (plane > 256 * CONV_MAX_E) {
        ccvs_set_error("ccvs_conv2d: halo tile %dx%d too large", halo_h, halo_w);
        return CCVS_ERR_ARG;
    }
    const size_t smem = (size_t)(((CONV_CC * plane + 3) & ~3) + ntx_max * CONV_CC * NT) * sizeof(float);
    dim3 grid(k.tiles_x * k.tiles_y, k.CoutPad / NT, gz);
    hipLaunchKernelGGL((conv2d_mfma_kernel<TW, MB>), grid, dim3(256), smem, st, k);
    CCVS_CHECK_LAUNCH("ccvs_conv2d");
    return CCVS_OK;
}

extern "C" int ccvs_conv2d(const float* x, const float* w_packed, const float* bias, const float* residual, float* y,
                           const ccvs_conv_desc* d, void* stream) {
    CCVS_REQUIRE(x && w_packed && y && d, "ccvs_conv2d: null pointer");
    CCVS_REQUIRE(d->N > 0 && d->Cin > 0 && d->Cout > 0 && d->Hin > 0 && d->Win > 0, "ccvs_conv2d: empty tensor");
    CCVS_REQUIRE((d->kh == d->kw || d->kw == 1) && d->kh >= 1 && d->kh <= 9, "ccvs_conv2d: kernel %dx%d unsupported", d->kh, d->kw);
    CCVS_REQUIRE(d->CoutPad % 32 == 0 && d->CoutPad >= d->Cout, "ccvs_conv2d: CoutPad %d invalid for Cout %d", d->CoutPad, d->Cout);
    int Hout, Wout;
    if (d->transposed) {
        CCVS_REQUIRE(d->kh == 3 && d->stride == 2 && d->pad == 0, "ccvs_conv2d: transposed supports k3 s2 p0 only");
        Hout = 2 * d->Hin + d->kh - 2;
        Wout = 2 * d->Win + d->kw - 2;
    } else {
        CCVS_REQUIRE(d->stride == 1 || d->stride == 2, "ccvs_conv2d: stride %d unsupported", d->stride);
        Hout = (d->Hin + 2 * d->pad - d->kh) / d->stride + 1;
        Wout = (d->Win + 2 * d->pad - d->kw) / d->stride + 1;
    }
    CCVS_REQUIRE(Hout == d->Hout && Wout == d->Wout, "ccvs_conv2d: output %dx%d expected, got %dx%d", Hout, Wout, d->Hout, d->Wout);

    ConvK k;
    k.x = x; k.w = w_packed; k.bias = bias; k.res = residual; k.y = y;
    k.N = d->N; k.Cin = d->Cin; k.Hin = d->Hin; k.Win = d->Win; k.in_sN = d->in_sN; k.in_sC = d->in_sC;
    k.Cout = d->Cout; k.CoutPad = d->CoutPad; k.Hout = d->Hout; k.Wout = d->Wout;
    k.out_sN = d->out_sN; k.out_sC = d->out_sC; k.res_sN = d->res_sN; k.res_sC = d->res_sC;
    k.kh = d->kh; k.kw = d->kw; k.stride = d->stride; k.pad = d->pad; k.transposed = d->transposed ? 1 : 0;
    k.act = d->act; k.accumulate = d->accumulate; k.out_scale = d->out_scale;
    k.pre = d->pre; k.pre_sN = d->pre_sN; k.pre_sC = d->pre_sC; k.pre_div = d->pre_div > 0 ? d->pre_div : 1;
    k.in_p8 = 0; k.out_p8 = 0;
    k.nwork = 0; k.gx = k.gy = 1; k.work0 = 0; k.xcd_chunk = 0; k.cu_limit = 0; k.zi = 0; k.ktail = 0;  // the fp32-MFMA form always runs the classic 3-D grid
    CCVS_REQUIRE(!d->in_p8 && !d->out_p8, "ccvs_conv2d: packed activations are a split-bf16 format (use ccvs_conv2d_bf16x3)");

    // virtual grid (largest parity class for the transposed form)
    const int VH = d->transposed ? (Hout + 1) / 2 : Hout;
    const int VW = d->transposed ? (Wout + 1) / 2 : Wout;
    const int TW = VW > 16 ? 32 : (VW > 8 ? 16 : 8);
    const int TH = 256 / TW;
    k.tiles_x = cdiv(VW, TW);
    k.tiles_y = cdiv(VH, TH);
    const int s = d->transposed ? 1 : d->stride;
    const int ext_y = d->transposed ? 1 : d->kh - 1, ext_x = d->transposed ? 1 : d->kw - 1;
    const int ntx_max = d->transposed ? 2 : d->kw;
    const int halo_h = (TH - 1) * s + ext_y + 1, halo_w = (TW - 1) * s + ext_x + 1;
    const int gz = d->N * (d->transposed ? 4 : 1);
    CCVS_REQUIRE(gz <= 65535, "ccvs_conv2d: batch %d too large for one launch", d->N);
    hipStream_t st = (hipStream_t)stream;
    const bool mb2 = (d->CoutPad % 64 == 0);
    if (TW == 32) return mb2 ? launch_conv<32, 2>(k, halo_h, halo_w, ntx_max, gz, st) : launch_conv<32, 1>(k, halo_h, halo_w, ntx_max, gz, st);
    if (TW == 16) return mb2 ? launch_conv<16, 2>(k, halo_h, halo_w, ntx_max, gz, st) : launch_conv<16, 1>(k, halo_h, halo_w, ntx_max, gz, st);
    return mb2 ? launch_conv<8, 2>(k, halo_h, halo_w, ntx_max, gz, st) : launch_conv<8, 1>(k, halo_h, halo_w, ntx_max, gz, st);
}
