// Convolution on the bf16 matrix cores with fp32-class accuracy ("split-bf16", 3 products).
//
// Same implicit GEMM, tiling, tap table and epilogue as conv2d.hip, but each fp32 operand is
// split into two bf16 halves  v = hi + lo  (hi = RNE_bf16(v), lo = RNE_bf16(v - hi), residual
// <= 2^-17 |v|) and the product is evaluated as  hi*hi + hi*lo + lo*hi  on
// v_mfma_f32_32x32x16_bf16 with fp32 accumulation.  The dropped lo*lo term is <= 2^-16 relative,
// i.e. the per-product error is ~1e-5 -- two orders below the 1e-3 pixel tolerance of the path --
// while the matrix pipe runs 16x the fp32-MFMA rate, so three products are still 5.3x faster than
// v_mfma_f32_32x32x2_f32 (157 TF -> 833 TF effective ceiling).
//
// Operand layout: a 32x32x16 MFMA takes 8 consecutive k per lane (lanes 0-31: k 0..7, lanes
// 32-63: k 8..15).  k = input channel, so both LDS images are "8 channels innermost":
//   input   [half][hi|lo][halo pixel][8 bf16]   -- converted from NCHW fp32 while staging
//   weights [tap][half][hi|lo][cout][8 bf16]    -- pre-split and pre-laid-out once by the host
// Every operand fetch is one ds_read_b128 with 16-B lane stride (conflict-free).
#pragma once
#include "common.h"
#include "conv_common.h"
#include <stdlib.h>
#include <stdio.h>

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
// The 16-byte stores of the epilogues.  -DCB_STORE_WT (an experiment, tools/r05/store_wt_ab.sh): write-through (`sc1`) -- the output
// tile leaves no dirty lines in the XCD's L2, so the write-back at the END of every other kernel on the chip (the release at a
// kernel boundary: 123 per token step beside these convolutions) finds nothing of ours to flush.
#ifdef CB_STORE_WT
__device__ __forceinline__ void cb_store16(void* dst, f32x4 v) {
    asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(dst), "v"(v) : "memory");
}
#else
__device__ __forceinline__ void cb_store16(void* dst, f32x4 v) { *reinterpret_cast<f32x4*>(dst) = v; }
#endif
__device__ __forceinline__ void cb_store16(void* dst, uint4 v) { cb_store16(dst, f32x4{__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w)}); }

#define CB_CC 16      // input channels per chunk = one MFMA K step
#define CB_MAX_E 5

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(3))) void lds_void;
typedef __attribute__((address_space(1))) const void glb_void;

// source of halo pixels outside the image when a packed (P8) activation tile is staged by LDS-DMA
static __device__ uint4 g_conv_zero16 = {0u, 0u, 0u, 0u};
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));

// two floats -> two bf16 (round to nearest even) in one v_cvt_pk_bf16_f32; element 0 in the low half
__device__ __forceinline__ unsigned pk_bf16(float a, float b) {
    const f32x2 v = {a, b};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2));
}

// 8 floats -> 8 bf16 hi (uint4) + 8 bf16 lo (uint4),  v = hi + lo + O(2^-17 |v|)
__device__ __forceinline__ void split8(const float (&v)[8], uint4& hi, uint4& lo) {
    unsigned h[4], l[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        h[i] = pk_bf16(v[2 * i], v[2 * i + 1]);
        l[i] = pk_bf16(v[2 * i] - __uint_as_float(h[i] << 16), v[2 * i + 1] - __uint_as_float(h[i] & 0xffff0000u));
    }
    hi = make_uint4(h[0], h[1], h[2], h[3]);
    lo = make_uint4(l[0], l[1], l[2], l[3]);
}

// Stage `n_iter`-strided elements of one 16-channel chunk of the input halo tile: NCHW fp32 ->
// [half][hi|lo][pixel][8 bf16].  Loads are unconditional from clamped addresses (a predicated load
// in an unrolled loop makes hipcc branch and wait per element) and zeroed afterwards.
__device__ __forceinline__ void stage_pixel(const float* __restrict__ xn, long in_sC, int Cin, int c0, int off, uint4* in_tile, int plane,
                                            int e) {
    const bool inside = off >= 0;
    const float* src = xn + (inside ? off : 0);
    const bool full = (c0 + CB_CC <= Cin);
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        float v[8];
        if (full) {
#pragma unroll
            for (int i = 0; i < 8; ++i) v[i] = src[(long)(c0 + 8 * h + i) * in_sC];
        } else {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int c = c0 + 8 * h + i;
                const float t = src[(long)min(c, Cin - 1) * in_sC];
                v[i] = c < Cin ? t : 0.f;
            }
        }
        if (!inside) {
#pragma unroll
            for (int i = 0; i < 8; ++i) v[i] = 0.f;
        }
        uint4 hi, lo;
        split8(v, hi, lo);
        in_tile[(h * 2 + 0) * plane + e] = hi;
        in_tile[(h * 2 + 1) * plane + e] = lo;
    }
}

// Epilogue of one 32-channel x 32-pixel accumulator block written straight from registers (stride-2 / transposed layers and the
// synchronous kernel: a lane owns one output pixel and 16 of the channels).  `s_waitcnt vmcnt` counts stores as well as loads, so
// a load issued behind a store cannot be waited for without also waiting until the store has been acknowledged by memory
// (~0.6 us): with bias, residual and store interleaved element by element -- the form of rounds 1-3 -- a wave paid that round trip
// 64-128 times per tile, several times the tile's matrix work.  Here the bias comes from LDS (`bias_s`: the 32 values of the
// block), the addend of all 16 elements is fetched first, and the 16 stores follow each other without a load in between.
// add_kind: 0 none, 1 pre-activation image, 2 residual, 3 accumulate, 4 several (element by element).
__device__ __forceinline__ void epi_block16(const ConvK& p, const f32x16& a, const float* bias_s, int n, int co0, long opix, int add_kind) {
    float* yb = p.y + (long)n * p.out_sN + opix;
    if (add_kind == 0) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int cl = (r & 3) + 8 * (r >> 2), co = co0 + cl;
            if (co < p.Cout) {
                float v = a[r] + bias_s[cl];
                if (p.act == CCVS_ACT_LRELU) v = lrelu01(v);
                yb[(long)co * p.out_sC] = v * p.out_scale;
            }
        }
    } else {
        const float* pb = p.pre ? p.pre + (long)(n / p.pre_div) * p.pre_sN + opix : nullptr;
        const float* rb = p.res ? p.res + (long)n * p.res_sN + opix : nullptr;
#pragma unroll
        for (int r0 = 0; r0 < 16; r0 += 8) {   // (two halves: 24 registers of addends instead of 48 -- the 32-channel kernels are capped at 128)
            float a1[8], a2[8], a3[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const int r = r0 + q;
                const long co = min(co0 + (r & 3) + 8 * (r >> 2), p.Cout - 1);
                a1[q] = pb ? pb[co * p.pre_sC] : 0.f;
                a2[q] = rb ? rb[co * p.res_sC] : 0.f;
                a3[q] = p.accumulate ? yb[co * p.out_sC] : 0.f;
            }
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const int r = r0 + q;
                const int cl = (r & 3) + 8 * (r >> 2), co = co0 + cl;
                if (co < p.Cout) {
                    float v = (a[r] + a1[q]) + bias_s[cl];
                    if (p.act == CCVS_ACT_LRELU) v = lrelu01(v);
                    v = (v + a2[q]) * p.out_scale;
                    yb[(long)co * p.out_sC] = v + a3[q];
                }
            }
            __builtin_amdgcn_sched_barrier(0);   // (or hipcc hoists the loads of every following block above these stores: ~100 registers, spilled)
        }
    }
}

__device__ __forceinline__ int conv_add_kind(const ConvK& p) {
    const int n_add = (p.pre ? 1 : 0) + (p.res ? 1 : 0) + (p.accumulate ? 1 : 0);
    return n_add == 0 ? 0 : (n_add > 1 ? 4 : (p.pre ? 1 : (p.res ? 2 : 3)));
}

// NW waves (4 or 8): with 8, every wave owns ONE 32-pixel block (half the accumulators) and twice as many threads stage the
// halo tile -- the workgroup is alone on its CU (LDS), so the extra waves are what overlaps its loads (env CCVS_CONV_SYNC_WAVES).
template <int TW, int MB, int NW>
__global__ __launch_bounds__(64 * NW) void conv2d_bf16x3_kernel(ConvK p, const uint4* __restrict__ wsplit, int CinG) {
    constexpr int TH = 256 / TW;
    constexpr int NT = 32 * MB;
    constexpr int NTH = 64 * NW, PP = 8 / NW;
    extern __shared__ __attribute__((aligned(16))) uint4 smem4[];
    __shared__ float bias_s[32 * MB];   // read in the epilogue (epi_block16)

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    CONV_TILE_COORDS(p, bx, by, bz)
    const int ty = bx / p.tiles_x, tx = bx - ty * p.tiles_x;
    const int n0 = by * NT;
    int n = bz, cls = 0;
    if (p.transposed) { cls = n & 3; n >>= 2; }
    const AxisTaps ay = axis_taps(p.kh, p.stride, p.pad, p.transposed, cls >> 1, p.Hout);
    const AxisTaps ax = axis_taps(p.kw, p.stride, p.pad, p.transposed, cls & 1, p.Wout);
    if (ty * TH >= ay.V || tx * TW >= ax.V) return;
    if (tid < 32 * MB) bias_s[tid] = (p.bias && n0 + tid < p.Cout) ? p.bias[n0 + tid] : 0.f;   // visible after the first barrier of the chunk loop

    const int IH = (TH - 1) * ay.s + ay.ext + 1;
    const int IW = (TW - 1) * ax.s + ax.ext + 1;
    const int plane = IH * IW;
    const int iy0 = ty * TH * ay.s + ay.lo, ix0 = tx * TW * ax.s + ax.lo;
    uint4* in_tile = smem4;              // [half 2][part 2][plane]
    uint4* w_tile = smem4 + 4 * plane;   // [ntx][half 2][part 2][NT]

    int off[CB_MAX_E];
#pragma unroll
    for (int j = 0; j < CB_MAX_E; ++j) {
        const int e = tid + NTH * j;
        off[j] = -2;
        if (e < plane) {
            const int r = e / IW, c = e - r * IW;
            const int gy = iy0 + r, gx = ix0 + c;
            off[j] = (gy >= 0 && gy < p.Hin && gx >= 0 && gx < p.Win) ? gy * p.Win + gx : -1;
        }
    }
    int bofs[PP];
#pragma unroll
    for (int pp = 0; pp < PP; ++pp) {
        const int pj = (wave * PP + pp) * 32 + (lane & 31);
        const int prow = pj / TW, pcol = pj - prow * TW;
        bofs[pp] = prow * ay.s * IW + pcol * ax.s;
    }
    const int khalf = lane >> 5;

    f32x16 acc[MB][PP];
#pragma unroll
    for (int m = 0; m < MB; ++m)
#pragma unroll
        for (int pp = 0; pp < PP; ++pp)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[m][pp][r] = 0.f;

    const float* xn = p.x + (long)n * p.in_sN;
    const int wunits = ax.nt * 4 * NT;  // 16-B units of one tap row

    for (int c0 = 0; c0 < p.Cin; c0 += CB_CC) {
        __syncthreads();
#pragma unroll
        for (int j = 0; j < CB_MAX_E; ++j) {
            if (off[j] != -2) stage_pixel(xn, p.in_sC, p.Cin, c0, off[j], in_tile, plane, tid + NTH * j);
        }
        const int cg0 = c0 >> 3;
        for (int a = 0; a < ay.nt; ++a) {
            if (a > 0) __syncthreads();
            const int wy = ay.w0 + a * ay.dw;
            for (int i = tid; i < wunits; i += NTH) {
                const int b = i / (4 * NT), rem = i - b * (4 * NT);
                const int hp = rem / NT, co = rem - hp * NT;   // hp = half*2 + part
                const int tap = wy * p.kw + (ax.w0 + b * ax.dw);
                const int cg = cg0 + (hp >> 1);  // < CinG: the host pads Cin to a multiple of 16
                w_tile[i] = wsplit[(((long)tap * CinG + cg) * 2 + (hp & 1)) * p.CoutPad + n0 + co];
            }
            __syncthreads();
            const int dyl = (ay.d0 + a * ay.dd - ay.lo) * IW;
            for (int b = 0; b < ax.nt; ++b) {
                const int dl = dyl + (ax.d0 + b * ax.dd - ax.lo);
                const uint4* wt = w_tile + (b * 4 + khalf * 2) * NT + (lane & 31);
                const uint4* it = in_tile + (khalf * 2) * plane + dl;
                bf16x8 bh[PP], bl[PP];
#pragma unroll
                for (int pp = 0; pp < PP; ++pp) {
                    bh[pp] = __builtin_bit_cast(bf16x8, it[bofs[pp]]);
                    bl[pp] = __builtin_bit_cast(bf16x8, it[plane + bofs[pp]]);
                }
#pragma unroll
                for (int m = 0; m < MB; ++m) {
                    const bf16x8 ah = __builtin_bit_cast(bf16x8, wt[m * 32]);
                    const bf16x8 al = __builtin_bit_cast(bf16x8, wt[NT + m * 32]);
#pragma unroll
                    for (int pp = 0; pp < PP; ++pp) {
                        acc[m][pp] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh[pp], acc[m][pp], 0, 0, 0);
                        acc[m][pp] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl[pp], acc[m][pp], 0, 0, 0);
                        acc[m][pp] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh[pp], acc[m][pp], 0, 0, 0);
                    }
                }
            }
        }
    }

    const int add_kind = conv_add_kind(p);
#pragma unroll
    for (int pp = 0; pp < PP; ++pp) {
        const int pj = (wave * PP + pp) * 32 + (lane & 31);
        const int prow = pj / TW, pcol = pj - prow * TW;
        const int vy = ty * TH + prow, vx = tx * TW + pcol;
        if (vy >= ay.V || vx >= ax.V) continue;
        const int oy = vy * ay.os + ay.oo, ox = vx * ax.os + ax.oo;
        const long opix = (long)oy * p.Wout + ox;
#pragma unroll
        for (int m = 0; m < MB; ++m) {
            if constexpr (MB == 1) {   // (register-capped kernels: the element-by-element form, written here -- through epi_block16 it spills)
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
            } else {
                epi_block16(p, acc[m][pp], bias_s + m * 32 + 4 * khalf, n, n0 + m * 32 + 4 * khalf, opix, add_kind);
            }
        }
    }
}

// ---------------------------------------------------------------------------------------
// Producer / consumer form (used whenever its LDS and register staging fit): a 512-thread workgroup whose waves
// 0-3 only issue MFMAs and whose waves 4-7 only stage -- each SIMD hosts one of each, the matrix pipe runs beside
// the VALU / memory / LDS pipes, so the fp32 -> split-bf16 conversion sits under the MFMAs.  A step is one tap row
// of one 16-channel chunk; the halo tile (per chunk) and the tap-row weights (per step) are double-buffered in LDS
// and handed over with ONE workgroup barrier per step.  The two roles run separate loops that meet at that barrier.
//
// Staging mode, template parameter NTY:
//   NTY = 1 | 3  "VEC", dense stride-1 layers on 16-byte aligned rows with 1 or 3 tap rows (the common case).
//        Activations: the halo tile widened to 4-pixel boundaries, one item = 4 pixels x 8 channels per staging
//        thread and chunk (8 x global_load_dwordx4 -> split -> 8 x ds_write_b128), requested a whole chunk before
//        it is converted.  Weights (pre-split by the host, no conversion): LDS-DMA (global_load_lds_dwordx4) issued
//        by the MFMA waves, so that the staging waves' in-order vmcnt queue holds activation loads only.
//   NTY = -8     packed split-bf16 input (ccvs_conv_desc.in_p8): activations by LDS-DMA too, no conversion.
//   NTY = 0      scalar staging, any tap count (k x 1 heads with odd padding, unaligned views): 16 dword loads per
//        staging thread and step, software-pipelined one step deep in registers together with the weights
//        (convert + store what was loaded during step s-1, then request the bundle of step s+2).
//   NTY = -2     the same with two pixel passes per thread and step (transposed layers: larger halo tiles).
// Stride-2 layers and halo tiles that do not fit the double buffer use the synchronous kernel above.
// ---------------------------------------------------------------------------------------
// (MB = 2 compiled for 4 waves per SIMD -- two workgroups per CU, 17 VGPRs spilled -- measured 236 vs 246 TFLOP/s on 128->64 3x3 at 256^2: not kept)
// PP = pixel blocks (of 32) per MFMA wave: 2 = the 256-pixel tile; 4 = a 512-pixel tile (TW x 512/TW), for layers with <= 64
// output channels.  Their MFMA waves are LDS-READ bound: with two pixel blocks a tap costs 2 MB + 4 operand reads for 6 MB
// MFMAs -- 0.67 reads per MFMA at MB = 2 against 0.5 at MB = 4, each read 8 cycles of the CU's LDS bandwidth against 32
// cycles of one SIMD's matrix pipe, four SIMDs wide (the ablation with no staging at all still takes 6.7 of 9.25 ms on
// 128->64 3x3, 2.4x its pure MFMA time).  Four pixel blocks held in registers against the same weight fragments bring a
// 64-channel layer to the read intensity of a 128-channel one (2 MB + 8 reads for 12 MB MFMAs) at the same accumulator
// count (MB x PP x 16 = 128).  VEC staging only (dense stride-1 rows), two activation items per staging thread.
// WPC = 2 (64 output channels per workgroup, dense 3 x 3 only): compiled for TWO workgroups per CU (<= 128 VGPRs, ~77 KB of
// LDS each, one register set in the staging waves).  A tile's prologue (workgroup launch, first chunk: ~5 us) and epilogue
// (131 KB of output per 128-channel tile: ~10 us at the rate the chip writes) are not overlapped with anything while a CU holds
// ONE workgroup -- CCVS_CONV_ABLATE runs: 14.5 us of the 36 us a 49->128 tile takes, 40 % -- so layers with few input channels
// per output byte gain from a second resident workgroup whose K loop runs meanwhile, although its activations are then
// staged once per 64 output channels instead of once per 128.
template <int TW, int MB, int NTY, int PP = 2, int WPC = 1>
__global__ __launch_bounds__(512, ((MB == 1 && PP == 2) || WPC == 2) ? 4 : 2) void conv2d_bf16x3_pc_kernel(ConvK p, const uint4* __restrict__ wsplit, int CinG, int ntx_max, int ablate) {
    static_assert(WPC == 1 || (WPC == 2 && MB == 2 && PP == 2 && (NTY == 3 || NTY == -83)), "two workgroups per CU: the 64-channel 3 x 3 forms only");
    static_assert(PP == 2 || (PP == 4 && (NTY > 0 || NTY == -83) && MB == 2), "the 512-pixel tile exists for the VEC / packed staging modes and 64 output channels");
    static_assert(conv_nty_fetch_bytes(NTY) != 0, "unknown staging mode: classify it in conv_common.h (conv_nty_*)");
    constexpr bool VEC = conv_nty_vec(NTY);
    // NTY == -8: the input is a packed split-bf16 activation (P8: [N][C/8][hi|lo][H][W] x 8 bf16, written by the epilogue of
    // the producing convolution): its halo tile is already in the LDS image's format, so the staging waves only issue
    // LDS-DMA (global_load_lds_dwordx4, zero source outside the image) -- no registers, no conversion.
    constexpr bool P8IN = conv_nty_p8(NTY);   // -83: packed input AND three tap rows known at compile time (3 x 3 layers)
    constexpr bool TAP3 = NTY == 3 || NTY == -83;    // the tap loop of the MFMA waves written out
    constexpr bool DMAW = VEC || P8IN;   // weights by LDS-DMA from the MFMA waves
    constexpr int CB_XQ = (NTY == -2) ? 2 : 1;  // scalar staging: halo-tile pixel passes per thread and step (passes <= CB_XQ * nt)
    constexpr int NPIX = 128 * PP;        // pixels of the tile: 4 MFMA waves x PP blocks of 32
    constexpr int TH = NPIX / TW;
    constexpr int NT = 32 * MB;
    constexpr int NP = 256;               // staging threads (waves 4-7)
    constexpr int NPW = 256;              // ... all of which stage weights
    constexpr int NPX = 256;              // ... and (VEC) one activation item each
    constexpr int NTXM = (MB == 1) ? 9 : 3;              // most taps per row (1 x k head kernels run with MB = 1)
    constexpr int CB_WR = (NTXM * 4 * NT + NPW - 1) / NPW;  // uint4 of tap-row weights per weight-staging thread
    extern __shared__ __attribute__((aligned(16))) uint4 smem4[];
    __shared__ float bias_s[32 * MB];   // the workgroup's bias values: read from LDS in the epilogue (see there: no vector-memory wait between its stores)

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // Wave priority of the WHOLE workgroup (bits 16-17 of `ablate`, set by the launcher from CCVS_CONV_PRIO): beside the token
    // loops of other batches every SIMD also hosts waves of the decode kernels, and vector / memory issue is arbitrated by
    // priority, then age (tools/conv_contention_probe.py: one memory-streaming wave per SIMD costs the convolution 20-35 %,
    // a VALU-spinning one 140-200 % -- whether its loads hit L2 or HBM makes no difference).
    {
        const int prio = (ablate >> 16) & 3;
        if (prio == 1) __builtin_amdgcn_s_setprio(1);
        else if (prio == 2) __builtin_amdgcn_s_setprio(2);
        else if (prio == 3) __builtin_amdgcn_s_setprio(3);
    }
    // (measured: making the role provably wave-uniform with readfirstlane, or s_setprio(1) on the MFMA
    //  waves, both cost ~25 % on the 195->128 3x3 shape with hipcc / ROCm 7.2 -- left as plain predication)
    const bool producer = wave >= 4;
    const int rt = producer ? tid - 256 : tid, rw = wave & 3;  // thread / wave index inside the role
    // VEC: the staging waves handle the activations only; the pre-split weights need no conversion and go
    // global -> LDS by LDS-DMA (global_load_lds_dwordx4) issued by the MFMA waves themselves.  Loads retire in order
    // per wave: with both operands in one wave's queue, waiting for next step's weights also waits for the
    // activation loads issued before them, which caps their latency budget at two steps; alone in the queue they
    // get a whole chunk (NTY steps).  (Measured alternative: waves 4-5 activations / 6-7 weights was 10 % slower --
    // the conversion work then sits on two of the four SIMDs.)
    const bool xrole = VEC && producer;
    const int rtw = rt;
    CONV_TILE_COORDS(p, bx, by, bz)
    const int ty = bx / p.tiles_x, tx = bx - ty * p.tiles_x;
    const int n0 = by * NT;
    int n = bz, cls = 0;
    if (!TAP3 && p.transposed) { cls = n & 3; n >>= 2; }
    // GEO3 (the written-out 3 x 3 instantiations: stride 1, padding 1, not transposed -- the launcher sees to it): the tap table and
    // with it the halo tile's geometry (IH, IW, row pitch, plane, item counts) are COMPILE-TIME constants.  The set-up in front of the
    // first barrier was ~1900 instructions per wave, two dozen of them integer divisions by these launch-uniform values (25
    // instructions each): 8.4 k cycles of a tile's 10.8 k-cycle prologue (profiles/r05_conv_ablate_cycles.txt) were this code, not latency.
    constexpr bool GEO3 = TAP3;
    const AxisTaps ay = GEO3 ? axis_taps_k3(p.Hout) : axis_taps(p.kh, p.stride, p.pad, p.transposed, cls >> 1, p.Hout);
    const AxisTaps ax = GEO3 ? axis_taps_k3(p.Wout) : axis_taps(p.kw, p.stride, p.pad, p.transposed, cls & 1, p.Wout);
    if (ty * TH >= ay.V || tx * TW >= ax.V) return;
    if (ablate & 32768) return;   // timing experiments: the cost of dispatching the workgroups alone
    {   // Phase stagger (bits 18-23 of `ablate`, CCVS_CONV_STAGGER): the first workgroup of every CU starts ph x stg x ~3.8 us late,
        // ph = 0..7 by dispatch order, so that the CUs do not run their tiles -- whose prologues read and whose epilogues write in
        // bursts -- in lock-step from the launch on
        const int stg = (ablate >> 18) & 63;
        if (stg) {
            const int lin = p.nwork > 0 ? (int)blockIdx.x : (int)(blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z));
            if (lin < 256) {
                const int ph = (lin >> 3) & 7;
                for (int i = 0; i < ph * stg; ++i) __builtin_amdgcn_s_sleep(127);
            }
        }
    }
    if (tid < 32 * MB) bias_s[tid] = (p.bias && n0 + tid < p.Cout) ? p.bias[n0 + tid] : 0.f;   // visible after the first step barrier

    const int IH = (TH - 1) * ay.s + ay.ext + 1;
    const int IW = (TW - 1) * ax.s + ax.ext + 1;
    const int iy0 = ty * TH * ay.s + ay.lo, ix0 = tx * TW * ax.s + ax.lo;
    // VEC: the LDS image starts at the 4-pixel boundary at or left of ix0 (xsh pixels earlier) and has NQ float4 columns
    const int xsh = VEC ? ((ax.lo % 4) + 4) % 4 : 0;
    const int NQ = (xsh + IW + 3) >> 2;
    // LDS row stride in pixels.  VEC: ODD, and consecutive staging lanes take consecutive ROWS of one float4 column: the 8
    // lanes a ds_write_b128 services together then hit 8 different 16-byte slots modulo 128 B (banks of a store:
    // (a/4) mod 32) -- with the rows 4*NQ slots apart and lanes along the row the same stores were 4-way conflicts
    // (75 % of the staging waves' LDS cycles, SQ_LDS_BANK_CONFLICT).
    const int IWS = VEC ? 4 * NQ + 1 : IW;
    const int plane = IH * IWS;
    const int ntxm_ = GEO3 ? 3 : ntx_max;
    const int in_sz = 4 * plane, w_sz = ntxm_ * 4 * NT;
    uint4* in_buf = smem4;               // [2][half 2][part 2][plane]
    uint4* w_buf = smem4 + 2 * in_sz;    // [2][ntx][half 2][part 2][NT]

    const float* xn = p.x + (long)n * p.in_sN;
    const int nt = ay.nt;
    const int nchunks = (p.Cin + CB_CC - 1) / CB_CC;
    // packed K tail (ccvs_conv_desc.w_ktail): the last chunk is ONE step, run after the step loop of the MFMA waves
    const bool ktail = NTY == 3 && p.ktail > 0;
    const int nsteps = ktail ? (nchunks - 1) * nt + 1 : nchunks * nt;
    const int wunits = ax.nt * 4 * NT;

    // ---- producer helpers -------------------------------------------------------------------
    // All per-lane address parts are computed ONCE; per step only wave-uniform (scalar) bases change,
    // so a load is `uniform base + 32-bit lane offset` with no vector address arithmetic.
    constexpr int NE = (VEC || P8IN) ? 1 : CB_MAX_E;
    int offs[NE];  // halo element -> plane offset; -2: no such element, -1: outside the image
    {
        // (row, column) of element rt by ONE division, then 256 elements further per round: an integer division by a run-time value
        // is ~25 instructions, and this set-up code runs in front of every tile (see GEO3 above)
        int er = 0, ec = 0;
        const int dr_ = NP / IW, dc_ = NP - dr_ * IW;    // uniform
        if (!VEC && !P8IN && producer) { er = rt / IW; ec = rt - er * IW; }
#pragma unroll
        for (int j = 0; j < NE; ++j) {
            const int e = rt + NP * j;
            offs[j] = -2;
            if (!VEC && !P8IN && producer && e < plane) {
                const int gy = iy0 + er, gx = ix0 + ec;
                offs[j] = (gy >= 0 && gy < p.Hin && gx >= 0 && gx < p.Win) ? gy * p.Win + gx : -1;
            }
            er += dr_; ec += dc_;
            if (ec >= IW) { ec -= IW; ++er; }
        }
    }
    auto pix_offset = [&](int j) -> int {  // register-array select without dynamic indexing
        int o = -2;
#pragma unroll
        for (int jj = 0; jj < NE; ++jj) o = (j == jj) ? offs[jj] : o;
        return o;
    };
    // VEC item(s) of this thread: half vh (8 channels), tile row vr, float4 column vq
    constexpr int NI = PP / 2;
    bool vitem[NI], vin[NI];
    int ve[NI], vh[NI];
    const float* vptr[NI];
#pragma unroll
    for (int j = 0; j < NI; ++j) { vitem[j] = false; vin[j] = false; ve[j] = 0; vh[j] = 0; vptr[j] = xn; }
    if (xrole) {
        // Item order inside a half: column PAIR slowest, then the row, then the column of the pair -- 8 consecutive lanes =
        // 2 neighbouring float4 columns x 4 rows.  Their ds_write_b128 still land in 8 different 16-byte slots modulo
        // 128 B (slot = 4 q + r (4 NQ + 1) mod 8 with 4 NQ + 1 = 1 or 5 mod 8), and their global loads touch 4 cache lines
        // instead of the 8 of the rows-fastest order (a wave-load: 32 lines instead of 64 for the texture addresser).
        const int NQP = (NQ + 1) >> 1;                       // column pairs
        const bool pairs = !(ablate & 8192);                 // 8192: the rows-fastest order of round 2
        const int per_half = pairs ? NQP * IH * 2 : IH * NQ;
        const int n_items = per_half * 2;
#pragma unroll
        for (int j = 0; j < NI; ++j) {
            const int it0_ = rt + NPX * j;
            const int it = min(it0_, n_items - 1);
            vh[j] = it / per_half;
            const int rem = it - vh[j] * per_half;
            int vq, vr;
            if (pairs) {
                const int t = rem >> 1;
                const int qh = t / IH;
                vr = t - qh * IH;
                vq = 2 * qh + (rem & 1);
            } else {
                vq = rem / IH;
                vr = rem - vq * IH;
            }
            vitem[j] = it0_ < n_items && vq < NQ;
            vq = min(vq, NQ - 1);
            const int gy = iy0 + vr, gxa = ix0 - xsh + 4 * vq;
            vin[j] = gy >= 0 && gy < p.Hin && gxa >= 0 && gxa < p.Win;  // aligned and Win % 4 == 0: all 4 pixels in or out
            vptr[j] = xn + (long)(8 * vh[j]) * p.in_sC + (vin[j] ? gy * p.Win + gxa : 0);
            ve[j] = vr * IWS + 4 * vq;
        }
    }
    // two register sets: a chunk's activation loads are requested TWO chunks before they are converted (see the staging loop)
    f32x4 xv[2][NI][VEC ? 8 : 1];
    int xvc0[2] = {-1, -1};  // first channel of the chunk held in each set; -1: nothing
    int wofs[CB_WR];  // lane part of the weight address (uint4 units): tap column, half, hi|lo, cout
#pragma unroll
    for (int i = 0; i < CB_WR; ++i) {
        const int ic = min(rtw + NPW * i, wunits - 1);
        const int b = ic / (4 * NT), rem = ic - b * (4 * NT);
        const int hp = rem / NT, co = rem - hp * NT;
        wofs[i] = ((b * ax.dw * CinG + (hp >> 1)) * 2 + (hp & 1)) * p.CoutPad + co;
    }
    u32x4 wraw[CB_WR];  // native vector type: HIP's uint4 struct array does not stay in registers across the loop
    float xraw[CB_XQ][VEC ? 1 : 16];
    int xoff[CB_XQ], xc0 = 0;
#pragma unroll
    for (int q = 0; q < CB_XQ; ++q) xoff[q] = -2;

    const int ay_w0 = ay.w0, ay_dw = ay.dw, ax_w0 = ax.w0, kw_ = p.kw, cout_pad = p.CoutPad, cin_ = p.Cin;
    const long in_sC = p.in_sC;
    auto load_w = [&](int ci_, int a_) {
        const uint4* base = wsplit + ((((long)(ay_w0 + a_ * ay_dw) * kw_ + ax_w0) * CinG + ci_ * 2) * 2) * cout_pad + n0;  // uniform
#pragma unroll
        for (int i = 0; i < CB_WR; ++i) wraw[i] = *reinterpret_cast<const u32x4*>(base + wofs[i]);
    };
    auto store_w = [&](uint4* dst) {
#pragma unroll
        for (int i = 0; i < CB_WR; ++i)
            if (rtw + NPW * i < wunits) *reinterpret_cast<u32x4*>(dst + rtw + NPW * i) = wraw[i];
    };
    // P8 input: slot s = rt + 256 j of the 4-plane LDS image [half][hi|lo][pixel] <- one uint4 of the packed tensor
    constexpr int NJ = P8IN ? (PP == 4 ? 10 : 8) : 1;   // 16-byte slots of the 4-plane halo image per staging thread
    int p8off[NJ], p8q[NJ];
    {
        // (plane q, row, column) of slot rt by two divisions, then 256 slots further per round (no division per slot)
        int sq = 0, sr = 0, sc = 0;
        const int dr_ = 256 / IW, dc_ = 256 - dr_ * IW;   // uniform
        if (P8IN && producer) {
            sq = rt / plane;
            const int e = rt - sq * plane;
            sr = e / IW;
            sc = e - sr * IW;
        }
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            p8off[j] = -2;  // no such slot
            p8q[j] = 0;
            const int s_ = rt + 256 * j;
            if (P8IN && producer && s_ < 4 * plane) {
                const int gy = iy0 + sr, gx = ix0 + sc;
                p8off[j] = (gy >= 0 && gy < p.Hin && gx >= 0 && gx < p.Win) ? gy * p.Win + gx : -1;
                p8q[j] = sq;
            }
            sr += dr_; sc += dc_;
            if (sc >= IW) { sc -= IW; ++sr; }
            if (sr >= IH) { sr -= IH; ++sq; }
            if (sr >= IH) { sr -= IH; ++sq; }   // (a plane of fewer than 256 slots: two wraps per round at most -- planes hold >= 128)
        }
    }
    const int gin = (p.Cin + 7) >> 3;
    const long hw_in = (long)p.Hin * p.Win;
    auto dma_x = [&](int c_, uint4* dst) {
        const uint4* xb = reinterpret_cast<const uint4*>(p.x) + (long)n * gin * 2 * hw_in;  // uniform
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            if (p8off[j] != -2) {
                const int g_ = 2 * c_ + (p8q[j] >> 1);
                const uint4* src = (p8off[j] >= 0 && g_ < gin) ? xb + ((long)g_ * 2 + (p8q[j] & 1)) * hw_in + p8off[j] : &g_conv_zero16;
                __builtin_amdgcn_global_load_lds((glb_void*)src, (lds_void*)(dst + rw * 64 + 256 * j), 16, 0, 0);
            }
        }
    };
    // LDS-DMA of one tap row of weights: wave-instruction i of wave rw fills 64 consecutive uint4 of the row image
    auto dma_w = [&](int ci_, int a_, uint4* dst) {
        const uint4* base = wsplit + ((((long)(ay_w0 + a_ * ay_dw) * kw_ + ax_w0) * CinG + ci_ * 2) * 2) * cout_pad + n0;  // uniform
#pragma unroll
        for (int i = 0; i < CB_WR; ++i)
            if (rtw + NPW * i < wunits)
                __builtin_amdgcn_global_load_lds((glb_void*)(base + wofs[i]), (lds_void*)(dst + rw * 64 + NPW * i), 16, 0, 0);
    };
    auto load_xv = [&](int set, int c_) {   // `set` is a compile-time constant at every call site
        xvc0[set] = c_ * CB_CC;
        if (ablate & 16) return;
        const bool full = xvc0[set] + CB_CC <= cin_;
#pragma unroll
        for (int j = 0; j < NI; ++j) {
#pragma unroll
            for (int i = 0; i < (VEC ? 8 : 1); ++i) {
                const int c = full ? xvc0[set] + i : min(xvc0[set] + 8 * vh[j] + i, cin_ - 1) - 8 * vh[j];
                xv[set][j][i] = *reinterpret_cast<const f32x4*>(vptr[j] + (long)c * in_sC);
            }
        }
    };
    auto store_xv = [&](int set, uint4* dst) {
        if (xvc0[set] < 0) return;
#pragma unroll
        for (int j = 0; j < NI; ++j) {
            if (!vitem[j]) continue;
            const bool plain = vin[j] && (xvc0[set] + CB_CC <= cin_);
#pragma unroll
            for (int px = 0; px < 4; ++px) {
                float v[8];
#pragma unroll
                for (int i = 0; i < 8; ++i) v[i] = xv[set][j][VEC ? i : 0][px];
                if (!plain) {
#pragma unroll
                    for (int i = 0; i < 8; ++i) v[i] = (vin[j] && xvc0[set] + 8 * vh[j] + i < cin_) ? v[i] : 0.f;
                }
                uint4 hi, lo;
                if (ablate & 8) {
                    hi = make_uint4(__float_as_uint(v[0]), __float_as_uint(v[1]), __float_as_uint(v[2]), __float_as_uint(v[3]));
                    lo = make_uint4(__float_as_uint(v[4]), __float_as_uint(v[5]), __float_as_uint(v[6]), __float_as_uint(v[7]));
                } else {
                    split8(v, hi, lo);
                }
                dst[(vh[j] * 2 + 0) * plane + ve[j] + px] = hi;
                dst[(vh[j] * 2 + 1) * plane + ve[j] + px] = lo;
            }
        }
    };
    auto load_x = [&](int c_, int slot) {
        xc0 = c_ * CB_CC;
        const bool full = (xc0 + CB_CC <= cin_);
#pragma unroll
        for (int q = 0; q < CB_XQ; ++q) {
            xoff[q] = pix_offset(slot + q * nt);
            if (xoff[q] != -2) {
                const int o = max(xoff[q], 0);
#pragma unroll
                for (int i = 0; i < (VEC ? 1 : 16); ++i) {
                    const float* cb = xn + (long)(full ? xc0 + i : min(xc0 + i, cin_ - 1)) * in_sC;  // uniform
                    xraw[q][i] = cb[o];
                }
            }
        }
    };
    auto store_x = [&](int slot, uint4* dst) {
#pragma unroll
        for (int q = 0; q < CB_XQ; ++q) {
            if (xoff[q] == -2) continue;
            const int e = rt + NP * (slot + q * nt);
            const bool plain = (xoff[q] >= 0) && (xc0 + CB_CC <= cin_);
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                float v[8];
#pragma unroll
                for (int i = 0; i < 8; ++i) v[i] = xraw[q][VEC ? 0 : 8 * h + i];
                if (!plain) {  // border pixel or channel tail: zero what lies outside
#pragma unroll
                    for (int i = 0; i < 8; ++i) v[i] = (xoff[q] >= 0 && xc0 + 8 * h + i < cin_) ? v[i] : 0.f;
                }
                uint4 hi, lo;
                split8(v, hi, lo);
                dst[(h * 2 + 0) * plane + e] = hi;
                dst[(h * 2 + 1) * plane + e] = lo;
            }
        }
    };

    // ---- consumer state ---------------------------------------------------------------------
    int bofs[PP];
#pragma unroll
    for (int pp = 0; pp < PP; ++pp) {
        const int pj = (rw * PP + pp) * 32 + (lane & 31);
        const int prow = pj / TW, pcol = pj - prow * TW;
        bofs[pp] = prow * ay.s * IWS + pcol * ax.s + xsh;
    }
    const int khalf = lane >> 5;
    f32x16 acc[MB][PP];
#pragma unroll
    for (int m = 0; m < MB; ++m)
#pragma unroll
        for (int pp = 0; pp < PP; ++pp)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[m][pp][r] = 0.f;

    if (producer) {
        // prologue: chunk 0's halo tile and step 0's weights, synchronously and straight into LDS
        if (P8IN) {
            dma_x(0, in_buf);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        } else if (xrole) {
            load_xv(0, 0);
            store_xv(0, in_buf);
        } else if (!VEC) {
            for (int j = 0; j < NE; ++j) {
                const int o = pix_offset(j);
                if (o != -2) stage_pixel(xn, p.in_sC, p.Cin, 0, o, in_buf, plane, rt + NP * j);
            }
        }
        if (!DMAW) {
            const uint4* base = wsplit + ((((long)ay_w0 * kw_ + ax_w0) * CinG) * 2) * cout_pad + n0;
#pragma unroll
            for (int i = 0; i < CB_WR; ++i)
                if (rtw + NPW * i < wunits) w_buf[rtw + NPW * i] = base[wofs[i]];
        }
    }

    if (ablate & 4096) return;   // timing experiments: dispatch + address set-up + the first chunk / first weights, no step loop
    // Step -1 only lets the producers fetch the first register bundle (everyone meets at the barrier);
    // steps 0 .. nsteps-1 are the real ones.  (ci, a) = (chunk, tap row) of step s.
    // The two roles run SEPARATE loops that meet at the same barrier once per step: in one shared loop the register
    // allocator keeps the staging bundle live across the MFMA code and the accumulators live across the staging code.
    if (producer && P8IN) {
        // chunk c+1 is requested at step (c, 0) and must have landed when step (c, nt-1) ends; raw s_barrier: a
        // __syncthreads() here would drain the DMA (vmcnt 0) at EVERY step
        const bool work = !(ablate & 1);
        __builtin_amdgcn_s_barrier();  // step -1
        for (int ci = 0; ci < nchunks; ++ci) {
            for (int a = 0; a < nt; ++a) {
                if (work && a == 0 && ci + 1 < nchunks) dma_x(ci + 1, in_buf + ((ci + 1) & 1) * in_sz);
                if (a == nt - 1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
            }
        }
    } else if (producer && VEC) {
        // Chunk c is requested at step (c-3, 0) and converted + stored at step (c-1, 0), TWO chunks later, from one of two
        // register sets (the loop is unrolled by two so that the sets stay statically indexed); these are the wave's only
        // outstanding loads.  One chunk of lead (3 steps of a 128-cout layer, but only ~1.5 us of a 64-cout one) is less than
        // a load takes once the token loops of other batches share the memory system, and the MFMA waves then wait at the
        // chunk's first barrier; the staging role has the registers to spare (the kernel's allocation is the MFMA role's).
        // Requests past the end re-read the last chunk (and are never stored).
        // The step barrier of the staging waves is a RAW s_barrier behind an LDS-only wait: __syncthreads() also drains
        // vmcnt, i.e. it would make the wave sit at the first barrier of a chunk until the global loads it has just requested
        // have landed -- and the MFMA waves with it.  The loads stay in flight across the barriers; hipcc waits for them
        // where their registers are first read (store_xv, two chunks later).
        constexpr int NTYc = VEC ? NTY : 1;
        const bool work = !(ablate & (1 | 256));   // 256: weights still arrive, activations are not staged
        const bool drain = ablate & 64;   // experiments: the old __syncthreads() steps
        const bool deep = WPC == 1 && !(ablate & 2048);   // 2048: one chunk of lead (the round-2 schedule; always with two workgroups per CU)
        if (work) {
            load_xv(0, min(1, nchunks - 1));
            if (deep) load_xv(1, min(2, nchunks - 1));
        }
        __syncthreads();
        for (int ci = 0; ci < nchunks; ci += 2) {
            if (work) {   // set 0 holds chunk ci + 1
                if (ci + 1 < nchunks) store_xv(0, in_buf + ((ci + 1) & 1) * in_sz);
                if (deep) load_xv(0, min(ci + 3, nchunks - 1));
                else load_xv(1, min(ci + 2, nchunks - 1));
            }
#pragma unroll
            for (int a = 0; a < NTYc; ++a) {
                if (a > 0 && ktail && ci == nchunks - 1) break;   // the packed tail chunk is one step
                if (drain) __syncthreads();
                else asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
            }
            if (ci + 1 >= nchunks) break;
            if (work) {   // set 1 holds chunk ci + 2
                if (ci + 2 < nchunks) store_xv(1, in_buf + ((ci + 2) & 1) * in_sz);
                if (deep) load_xv(1, min(ci + 4, nchunks - 1));
                else load_xv(0, min(ci + 3, nchunks - 1));
            }
#pragma unroll
            for (int a = 0; a < NTYc; ++a) {
                if (a > 0 && ktail && ci + 1 == nchunks - 1) break;
                if (drain) __syncthreads();
                else asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the trailing (never stored) requests
    } else if (producer) {
      int ci = -1, a = nt - 1;
      for (int s = -1; s < nsteps; ++s) {
        if (!(ablate & 1)) {
            if (s >= 0) {  // (1) convert + store the bundle loaded during the previous step
                if (s + 1 < nsteps) store_w(w_buf + ((s + 1) & 1) * w_sz);
                if (ci + 1 < nchunks) store_x(a, in_buf + ((ci + 1) & 1) * in_sz);
            }
            // (2) issue the loads of the following bundle: W_row(s+2), halo parts of slot a+1
            int a1 = a + 1, c1 = ci;
            if (a1 == nt) { a1 = 0; ++c1; }
            int a2 = a1 + 1, c2 = c1;
            if (a2 == nt) { a2 = 0; ++c2; }
            if (s + 2 < nsteps) load_w(c2, a2);
            const int cx = (a1 == 0) ? ci + 2 : ci + 1;  // chunk whose parts of slot a1 are stored next step
            if (cx < nchunks) {
                load_x(cx, a1);
            } else {
#pragma unroll
                for (int q = 0; q < CB_XQ; ++q) xoff[q] = -2;
            }
        }
        __syncthreads();
        if (++a == nt) { a = 0; ++ci; }
      }
    } else {
      int ci = -1, a = nt - 1;
      const int nloop = ktail ? nsteps - 1 : nsteps;
#ifdef CB_STAMPS   // debug build: T0 loop top, T1 weight DMA issued, T2 the step's matrix instructions issued, T3 behind the step barrier
      const bool stamp_on = p.dbg && tid == 0 && (int)(blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z)) == p.dbg_block;
#define CB_STAMP(K)                                                                         \
    if (stamp_on && s >= 0 && s < 24) {                                                     \
        unsigned long long t_;                                                              \
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");          \
        p.dbg[4 * s + (K)] = (long long)t_;                                                 \
    }
#else
#define CB_STAMP(K)
#endif
      for (int s = -1; s < nloop; ++s) {
        CB_STAMP(0)
        if (DMAW && !(ablate & (1 | 128)) && s + 1 < nsteps) {   // 128: activations still staged, no weight DMA  // weights of step s+1 by LDS-DMA; hipcc drains them (vmcnt 0) at the barrier
            int a1 = a + 1, c1 = ci;
            if (a1 == nt) { a1 = 0; ++c1; }
            dma_w(c1, a1, w_buf + ((s + 1) & 1) * w_sz);
        }
        CB_STAMP(1)
        if (s >= 0 && !(ablate & 2)) {
            const uint4* it0 = in_buf + (ci & 1) * in_sz + (khalf * 2) * plane + (ay.d0 + a * ay.dd - ay.lo) * IWS;
            const uint4* wt0 = w_buf + (s & 1) * w_sz + (khalf * 2) * NT + (lane & 31);
            if constexpr (PP == 4) {
                // Four pixel blocks of a tap stay in registers (fq); the weight fragments of the tap's MB output-channel blocks
                // sit in one of two sets (fw): the next tap's are read above the current tap's 12 MB MFMAs (the tap loop is
                // unrolled by two so that the sets stay statically indexed).  There is no second set for the pixels: in the
                // pass over the LAST channel block each pixel block is re-read for the next tap right after its last MFMA --
                // 9 to 0 MFMAs before the next tap needs it, the earliest-needed block first.
                const int ntx = ax.nt, xd0 = ax.d0 - ax.lo, xdd = ax.dd;
                bf16x8 fq[4][2], fw[2][MB][2];
#pragma unroll
                for (int pp = 0; pp < 4; ++pp) {
                    fq[pp][0] = __builtin_bit_cast(bf16x8, it0[xd0 + bofs[pp]]);
                    fq[pp][1] = __builtin_bit_cast(bf16x8, it0[xd0 + plane + bofs[pp]]);
                }
#pragma unroll
                for (int m = 0; m < MB; ++m) {
                    fw[0][m][0] = __builtin_bit_cast(bf16x8, wt0[m * 32]);
                    fw[0][m][1] = __builtin_bit_cast(bf16x8, wt0[NT + m * 32]);
                }
#define CB_TAP4(CUR, tb, more)                                                                                         \
    {                                                                                                                  \
        const uint4* itn_ = it0 + (xd0 + ((tb) + 1) * xdd);                                                            \
        if (more) {                                                                                                    \
            const uint4* wtn_ = wt0 + ((tb) + 1) * 4 * NT;                                                             \
            _Pragma("unroll") for (int m = 0; m < MB; ++m) {                                                           \
                fw[(CUR) ^ 1][m][0] = __builtin_bit_cast(bf16x8, wtn_[m * 32]);                                        \
                fw[(CUR) ^ 1][m][1] = __builtin_bit_cast(bf16x8, wtn_[NT + m * 32]);                                   \
            }                                                                                                          \
        }                                                                                                              \
        __builtin_amdgcn_sched_barrier(0);                                                                             \
        _Pragma("unroll") for (int m = 0; m < MB; ++m) {                                                               \
            _Pragma("unroll") for (int pp = 0; pp < 4; ++pp) {                                                         \
                acc[m][pp] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fw[CUR][m][1], fq[pp][0], acc[m][pp], 0, 0, 0);   \
                acc[m][pp] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fw[CUR][m][0], fq[pp][1], acc[m][pp], 0, 0, 0);   \
                acc[m][pp] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fw[CUR][m][0], fq[pp][0], acc[m][pp], 0, 0, 0);   \
                if (m == MB - 1 && (more)) {                                                                           \
                    fq[pp][0] = __builtin_bit_cast(bf16x8, itn_[bofs[pp]]);                                            \
                    fq[pp][1] = __builtin_bit_cast(bf16x8, itn_[plane + bofs[pp]]);                                    \
                }                                                                                                      \
            }                                                                                                          \
        }                                                                                                              \
    }
                if constexpr (TAP3) {   // (written out: see the 256-pixel form below)
                    CB_TAP4(0, 0, true)
                    CB_TAP4(1, 1, true)
                    CB_TAP4(0, 2, false)
                } else {
                    int tb = 0;
                    for (; tb + 2 <= ntx; tb += 2) {
                        CB_TAP4(0, tb, true)
                        CB_TAP4(1, tb + 1, (tb + 2 < ntx))
                    }
                    if (tb < ntx) { CB_TAP4(0, tb, false) }
                }
#undef CB_TAP4
            } else if constexpr (WPC == 2) {
                // two workgroups per CU: every SIMD hosts two MFMA waves, one of which computes while the other waits for its
                // fragments -- plain reads (no second register set: the 128-VGPR budget), weights one block ahead
                const int ntx = ax.nt, xd0 = ax.d0 - ax.lo, xdd = ax.dd;
#pragma unroll
                for (int tb = 0; tb < 3; ++tb) {
                    if (tb < ntx) {
                        const uint4* it_ = it0 + (xd0 + tb * xdd);
                        const uint4* wt_ = wt0 + tb * 4 * NT;
                        bf16x8 qb[2][2], wa[2][2];
#pragma unroll
                        for (int pp = 0; pp < 2; ++pp) {
                            qb[pp][0] = __builtin_bit_cast(bf16x8, it_[bofs[pp]]);
                            qb[pp][1] = __builtin_bit_cast(bf16x8, it_[plane + bofs[pp]]);
                        }
                        wa[0][0] = __builtin_bit_cast(bf16x8, wt_[0]);
                        wa[0][1] = __builtin_bit_cast(bf16x8, wt_[NT]);
#pragma unroll
                        for (int m = 0; m < MB; ++m) {
                            if (m + 1 < MB) {
                                wa[(m + 1) & 1][0] = __builtin_bit_cast(bf16x8, wt_[(m + 1) * 32]);
                                wa[(m + 1) & 1][1] = __builtin_bit_cast(bf16x8, wt_[NT + (m + 1) * 32]);
                            }
#pragma unroll
                            for (int pp = 0; pp < 2; ++pp) {
                                acc[m][pp] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wa[m & 1][1], qb[pp][0], acc[m][pp], 0, 0, 0);
                                acc[m][pp] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wa[m & 1][0], qb[pp][1], acc[m][pp], 0, 0, 0);
                                acc[m][pp] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wa[m & 1][0], qb[pp][0], acc[m][pp], 0, 0, 0);
                            }
                        }
                    }
                }
            } else {
            // Software-pipelined fragment reads: the ds_reads of the NEXT 32-cout block (and, on a tap's last block,
            // of the next tap's pixels) are issued before the current block's 6 MFMAs, into the other register
            // set -- hipcc does not do this by itself and the lone MFMA wave of a SIMD then idles a full LDS
            // latency after every 6 MFMAs.  Taps are unrolled by two so both sets stay statically indexed.
            bf16x8 fa[2][2];     // [set][0 hi | 1 lo]       weights of one 32-cout block
            bf16x8 fb[2][2][2];  // [set][pp][0 hi | 1 lo]   the two pixel blocks of one tap
            const int ntx = ax.nt, xd0 = ax.d0 - ax.lo, xdd = ax.dd;
#define CB_LD_B(SET, tb)                                                                        \
    {                                                                                           \
        const uint4* it_ = it0 + (xd0 + (tb) * xdd);                                            \
        _Pragma("unroll") for (int pp = 0; pp < 2; ++pp) {                                      \
            fb[SET][pp][0] = __builtin_bit_cast(bf16x8, it_[bofs[pp]]);                         \
            fb[SET][pp][1] = __builtin_bit_cast(bf16x8, it_[plane + bofs[pp]]);                 \
        }                                                                                       \
    }
#define CB_LD_A(SET, tb, m_)                                                                    \
    {                                                                                           \
        const uint4* wt_ = wt0 + (tb) * 4 * NT + (m_) * 32;                                     \
        fa[SET][0] = __builtin_bit_cast(bf16x8, wt_[0]);                                        \
        fa[SET][1] = __builtin_bit_cast(bf16x8, wt_[NT]);                                       \
    }
#define CB_TAP(BSET, A0, tb, has_next)                                                          \
    _Pragma("unroll") for (int m = 0; m < MB; ++m) {                                            \
        if (m + 1 < MB) {                                                                       \
            CB_LD_A(((A0) + m + 1) & 1, tb, m + 1)                                              \
        } else if (has_next) {                                                                  \
            CB_LD_A(((A0) + m + 1) & 1, (tb) + 1, 0)                                            \
            CB_LD_B((BSET) ^ 1, (tb) + 1)                                                       \
        }                                                                                       \
        __builtin_amdgcn_sched_barrier(0); /* keep the prefetch ABOVE the MFMAs it is meant to hide under */ \
        _Pragma("unroll") for (int pp = 0; pp < 2; ++pp) {                                      \
            acc[m][pp] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[((A0) + m) & 1][1], fb[BSET][pp][0], acc[m][pp], 0, 0, 0); \
            acc[m][pp] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[((A0) + m) & 1][0], fb[BSET][pp][1], acc[m][pp], 0, 0, 0); \
            acc[m][pp] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[((A0) + m) & 1][0], fb[BSET][pp][0], acc[m][pp], 0, 0, 0); \
        }                                                                                       \
    }
            CB_LD_B(0, 0)
            CB_LD_A(0, 0, 0)
            if constexpr (TAP3) {
                // 3 x 3 kernels (three taps per row, known at compile time): the tap loop is written out, so that no
                // run-time branch sits between the fragment reads and the MFMAs.  In the loop form hipcc's wait-count pass
                // puts `s_waitcnt lgkmcnt(0)` directly behind every prefetch (`ds_read x2; s_waitcnt lgkmcnt(0); v_mfma x6`:
                // it waits for the reads it has JUST issued); written out, the waits sit 4-6 MFMAs behind the reads
                // (+1...+3 % per shape).  Hand-counted waits with the reads as asm statements were tried and are 2.4 x
                // SLOWER: with an LDS-DMA in flight (the next step's weights) hipcc drains vmcnt in front of every asm
                // statement that might touch LDS.
                CB_TAP(0, 0, 0, true)
                CB_TAP(1, (MB & 1), 1, true)
                CB_TAP(0, 0, 2, false)
            } else {
                int tb = 0;
                for (; tb + 2 <= ntx; tb += 2) {
                    CB_TAP(0, 0, tb, true)
                    CB_TAP(1, (MB & 1), tb + 1, (tb + 2 < ntx))
                }
                if (tb < ntx) { CB_TAP(0, 0, tb, false) }
            }
#undef CB_LD_B
#undef CB_LD_A
#undef CB_TAP
            }
        }
        CB_STAMP(2)
        __syncthreads();
        CB_STAMP(3)
        if (++a == nt) { a = 0; ++ci; }
      }
#undef CB_STAMP
      if constexpr (NTY == 3) {
        if (ktail) {
            // Packed K tail: the last chunk holds r = Cin % 16 <= 3 real channels.  Its nine taps x r channels are contracted
            // in ceil(9 r / 16) MFMA steps whose K index runs over (tap, channel): position q = 16 j + 8 khalf + i of step j
            // is tap q / r, channel q % r.  The weights arrive in that order (tap slots 0 .. nj-1 of tap row 0 of the chunk,
            // requested by the last iteration of the loop above); the pixel operand is gathered from the chunk's staged
            // tile, 2 bytes per (tap, channel) -- once per tile, against 9 - nj tap steps of 6 MB MFMAs saved.  Outside the
            // loop on purpose: inside it hipcc hoists the gather's address arithmetic over the whole loop and spills.
            if (!(ablate & 2)) {
                const int s = nsteps - 1;
                const int r_ = p.ktail, nq_ = 9 * r_, nj_ = (nq_ + 15) >> 4;
                const unsigned short* ih = reinterpret_cast<const unsigned short*>(in_buf + ((nchunks - 1) & 1) * in_sz);
                const uint4* wtl = w_buf + (s & 1) * w_sz + (khalf * 2) * NT + (lane & 31);
                for (int j = 0; j < nj_; ++j) {
                    bf16x8 gb[PP][2];
#pragma unroll
                    for (int pp = 0; pp < PP; ++pp) {
                        unsigned hw[4], lw[4];
#pragma unroll
                        for (int i = 0; i < 8; ++i) {
                            const int q = 16 * j + 8 * khalf + i;
                            const int qc = min(q, nq_ - 1);
                            const int t = qc / r_, c = qc - t * r_;
                            const int tyy = t / 3, txx = t - 3 * tyy;
                            const int e = (bofs[pp] + tyy * IWS + txx) * 8 + c;   // bf16 index inside the [pixel][8] plane
                            unsigned hv = ih[e], lv = ih[plane * 8 + e];
                            if (q >= nq_) { hv = 0; lv = 0; }
                            if (i & 1) { hw[i >> 1] |= hv << 16; lw[i >> 1] |= lv << 16; }
                            else { hw[i >> 1] = hv; lw[i >> 1] = lv; }
                        }
                        gb[pp][0] = __builtin_bit_cast(bf16x8, make_uint4(hw[0], hw[1], hw[2], hw[3]));
                        gb[pp][1] = __builtin_bit_cast(bf16x8, make_uint4(lw[0], lw[1], lw[2], lw[3]));
                    }
#pragma unroll
                    for (int m = 0; m < MB; ++m) {
                        const bf16x8 ah = __builtin_bit_cast(bf16x8, wtl[j * 4 * NT + m * 32]);
                        const bf16x8 al = __builtin_bit_cast(bf16x8, wtl[j * 4 * NT + NT + m * 32]);
#pragma unroll
                        for (int pp = 0; pp < PP; ++pp) {
                            acc[m][pp] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, gb[pp][0], acc[m][pp], 0, 0, 0);
                            acc[m][pp] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, gb[pp][1], acc[m][pp], 0, 0, 0);
                            acc[m][pp] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, gb[pp][0], acc[m][pp], 0, 0, 0);
                        }
                    }
                }
            }
            __syncthreads();
        }
      }
    }
    if (ablate & 16384) return;   // timing experiments: no epilogue (nothing is written)
    // ---- epilogue ------------------------------------------------------------------------------
    // Dense convolutions: the accumulators (one pixel column per lane, 16 couts in registers) go
    // through LDS so that ALL 8 waves write 16-byte pieces along x (a lane then owns 4 consecutive
    // pixels of one channel) instead of 128 scalar stores per consumer lane: the store tail was
    // issue-bound.  32 couts x 256 pixels per pass, MB passes.
    if (GEO3 || !p.transposed) {
        float* stage = reinterpret_cast<float*>(smem4);
        constexpr int NIT = (8 * NPIX + 255 + NP) / (256 + NP);   // 16-byte pieces of a 32-channel pass per thread
        constexpr bool LEANK = MB == 1 || WPC == 2;   // kernels capped at 128 registers: the fast path below or the plain piece-by-piece one
        // addends of the epilogue: 0 none, 1 pre-activation image, 2 residual, 3 accumulate, 4 more than one of them
        const int add_kind = conv_add_kind(p);
        const bool fast_epi = add_kind <= 3 && (tx + 1) * TW <= ax.V && (ty + 1) * TH <= ay.V && n0 + NT <= p.Cout && (p.Wout & 3) == 0 &&
                              (reinterpret_cast<uintptr_t>(p.y) & 15) == 0 && (p.out_sN & 3) == 0 && (p.out_sC & 3) == 0 &&
                              (add_kind != 1 || ((reinterpret_cast<uintptr_t>(p.pre) & 15) == 0 && (p.pre_sN & 3) == 0 && (p.pre_sC & 3) == 0)) &&
                              (add_kind != 2 || ((reinterpret_cast<uintptr_t>(p.res) & 15) == 0 && (p.res_sN & 3) == 0 && (p.res_sC & 3) == 0));
#pragma unroll
        for (int m = 0; m < MB; ++m) {
            const bool p8_out = (PP == 2 || P8IN) && p.out_p8;
            if (!producer) {
                if (p8_out) {
                    // packed output: the pass is staged [pixel][32 channels] (36 floats apart: conflict-free 16-byte accesses) -- a
                    // lane writes its four groups of 4 consecutive channels as ds_write_b128, a reader fetches the 8 channels of its
                    // pixel as two ds_read_b128 (channel-major staging cost the packed epilogue 8 ds_read_b32 per item: the
                    // 128-channel producers ran 20 % slower than with fp32 output)
#pragma unroll
                    for (int pp = 0; pp < PP; ++pp) {
                        float* sp = stage + ((rw * PP + pp) * 32 + (lane & 31)) * 36 + 4 * khalf;
#pragma unroll
                        for (int g = 0; g < 4; ++g) {
                            const f32x4 q4 = {acc[m][pp][4 * g], acc[m][pp][4 * g + 1], acc[m][pp][4 * g + 2], acc[m][pp][4 * g + 3]};
                            *reinterpret_cast<f32x4*>(sp + 8 * g) = q4;
                        }
                    }
                } else {
#pragma unroll
                    for (int pp = 0; pp < PP; ++pp)
#pragma unroll
                        for (int r = 0; r < 16; ++r)
                            stage[((r & 3) + 8 * (r >> 2) + 4 * khalf) * NPIX + (rw * PP + pp) * 32 + (lane & 31)] = acc[m][pp][r];
                }
            }
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");   // LDS only: the previous pass's stores stay in flight
            if (p8_out) {   // (the 512-pixel tile writes packed output only in its packed-input form)
                // packed output: a thread takes one pixel x 8 output channels of the staged 32 x 256 block, applies the
                // epilogue, splits to hi / lo and writes two 16-byte units (lanes = consecutive pixels: coalesced)
                uint4* y4 = reinterpret_cast<uint4*>(p.y);
                const int gout = (p.Cout + 7) >> 3;
                const long hw_out = (long)p.Hout * p.Wout;
                // (bias from LDS; with a pre-activation image its values for all items are fetched first, WITHOUT one the code path holds
                //  no vector-memory load at all: a load that is merely conditional still makes hipcc wait, vmcnt(0), where its value
                //  would be used -- i.e. for the stores of the item before; see the fp32 epilogue below)
                constexpr int NI8 = 4 * NPIX / 512;   // (pixel, 8 channels) items of a 32-channel pass per thread
#define CB_P8_ITEM(i)                                                                          \
        const int item_ = tid + 512 * (i);                                                     \
        const int gq_ = item_ / NPIX, px_ = item_ - gq_ * NPIX;                                \
        const int co0_ = n0 + m * 32 + gq_ * 8;                                                \
        const int prow_ = px_ / TW, pcol_ = px_ - prow_ * TW;                                  \
        const int vy_ = ty * TH + prow_, vx_ = tx * TW + pcol_;                                \
        const bool ok_ = co0_ < p.Cout && vy_ < ay.V && vx_ < ax.V;                            \
        const long opix_ = ok_ ? (long)vy_ * p.Wout + vx_ : 0;
#define CB_P8_FINISH(PRE)                                                                      \
    _Pragma("unroll") for (int i = 0; i < NI8; ++i) {                                          \
        CB_P8_ITEM(i)                                                                          \
        if (ok_) {                                                                             \
            float v[8];                                                                        \
            const f32x4 s0 = *reinterpret_cast<const f32x4*>(stage + px_ * 36 + gq_ * 8);      \
            const f32x4 s1 = *reinterpret_cast<const f32x4*>(stage + px_ * 36 + gq_ * 8 + 4);  \
            _Pragma("unroll") for (int c = 0; c < 8; ++c) {                                    \
                float t = c < 4 ? s0[c] : s1[c - 4];                                           \
                t += (PRE);                                                                    \
                t += bias_s[m * 32 + gq_ * 8 + c];                                             \
                if (p.act == CCVS_ACT_LRELU) t = lrelu01(t);                                   \
                v[c] = t * p.out_scale;                                                        \
            }                                                                                  \
            uint4 hi, lo;                                                                      \
            split8(v, hi, lo);                                                                 \
            uint4* dst = y4 + ((long)n * gout + (co0_ >> 3)) * 2 * hw_out + opix_;             \
            cb_store16(dst, hi);                                                               \
            cb_store16(dst + hw_out, lo);                                                      \
        }                                                                                      \
    }
                if (p.pre) {
                    float pv[NI8][8];
                    const float* pb = p.pre + (long)(n / p.pre_div) * p.pre_sN;
#pragma unroll
                    for (int i = 0; i < NI8; ++i) {
                        CB_P8_ITEM(i)
#pragma unroll
                        for (int c = 0; c < 8; ++c) pv[i][c] = pb[(long)min(co0_ + c, p.Cout - 1) * p.pre_sC + opix_];
                    }
                    CB_P8_FINISH(pv[i][c])
                } else {
                    CB_P8_FINISH(0.f)
                }
#undef CB_P8_ITEM
#undef CB_P8_FINISH
                if (m + 1 < MB) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
                continue;
            }
            if constexpr (P8IN && PP == 4) {   // this instantiation writes packed output only (the launcher sees to it): its fp32
                continue;                        // epilogue, compiled in as well, keeps hipcc from unrolling the pass loop
            } else {
            // No wait on vector memory inside the store loop: `s_waitcnt vmcnt` counts stores too, so a wait for a load issued
            // after a store -- the bias value, the residual of the next piece -- also waits until that store has been
            // acknowledged by memory (~0.6 us).  With the loads of every piece interleaved with its store the tile's 131 KB left
            // the CU one round trip at a time: 9.9 us per 128-channel tile (CCVS_CONV_ABLATE runs), a quarter of the time a
            // 49->128 tile takes.  The bias values are fetched once in front of the first pass; a layer without addends
            // (most of them) issues no load at all here, the others fetch the addends of ALL pieces of a pass first.
            auto piece = [&](int i, int& co, long& opix, int& nv, int& col, int& px) -> bool {
                const int idx4 = tid + (256 + NP) * i;
                col = idx4 / (NPIX / 4);
                px = (idx4 % (NPIX / 4)) * 4;
                co = n0 + m * 32 + col;
                const int prow = px / TW, pcol = px - prow * TW;
                const int vy = ty * TH + prow, vx = tx * TW + pcol;
                opix = (long)vy * p.Wout + vx;
                nv = min(4, ax.V - vx);
                return idx4 < 8 * NPIX && co < p.Cout && vy < ay.V && vx < ax.V;
            };
            if (fast_epi) {
                // The whole tile inside the image, every output channel real, every row 16-byte aligned (the layers that matter):
                // straight-line code -- hipcc can then COUNT its waits (vmcnt(n) for the addend of piece i leaves the stores of
                // the pieces before it in flight; behind a per-piece branch it falls back to vmcnt(0))
                const float* abase = add_kind == 1 ? p.pre + (long)(n / p.pre_div) * p.pre_sN : add_kind == 2 ? p.res + (long)n * p.res_sN : p.y + (long)n * p.out_sN;
                const long a_sC = add_kind == 1 ? p.pre_sC : add_kind == 2 ? p.res_sC : p.out_sC;
                float* ybase = p.y + (long)n * p.out_sN;
                // (offsets are recomputed where they are used: kept in arrays across the two phases they cost the 32-channel kernels,
                //  capped at 128 registers, a spill)
#define CB_EPI_OFFS(i)                                                                               \
        const int idx4_ = tid + (256 + NP) * (i);                                                    \
        const int col_ = idx4_ / (NPIX / 4), px_ = (idx4_ % (NPIX / 4)) * 4;                         \
        const int prow_ = px_ / TW, pcol_ = px_ - prow_ * TW;                                        \
        const long opix_ = (long)(ty * TH + prow_) * p.Wout + tx * TW + pcol_;                       \
        const int co_ = n0 + m * 32 + col_;
#define CB_EPI_FINISH(ADD1, ADD2, ADD3)                                                              \
    _Pragma("unroll") for (int i = 0; i < NIT; ++i) {                                                \
        CB_EPI_OFFS(i)                                                                               \
        const float4 a4 = *reinterpret_cast<const float4*>(stage + col_ * NPIX + px_);              \
        float v[4] = {a4.x, a4.y, a4.z, a4.w};                                                       \
        const float bv = bias_s[m * 32 + col_];                                                      \
        _Pragma("unroll") for (int j = 0; j < 4; ++j) {                                              \
            float t = (v[j] + (ADD1)) + bv;                                                          \
            if (p.act == CCVS_ACT_LRELU) t = lrelu01(t);                                             \
            t = (t + (ADD2)) * p.out_scale;                                                          \
            v[j] = t + (ADD3);                                                                       \
        }                                                                                            \
        cb_store16(ybase + (long)co_ * p.out_sC + opix_, f32x4{v[0], v[1], v[2], v[3]});           \
    }
                if (add_kind == 0) {
                    CB_EPI_FINISH(0.f, 0.f, 0.f)
                } else {
                    f32x4 ad[NIT];
#pragma unroll
                    for (int i = 0; i < NIT; ++i) {
                        CB_EPI_OFFS(i)
                        ad[i] = *reinterpret_cast<const f32x4*>(abase + (long)co_ * a_sC + opix_);
                    }
                    if (add_kind == 1) { CB_EPI_FINISH(ad[i][j], 0.f, 0.f) }
                    else if (add_kind == 2) { CB_EPI_FINISH(0.f, ad[i][j], 0.f) }
                    else { CB_EPI_FINISH(0.f, 0.f, ad[i][j]) }
                }
#undef CB_EPI_OFFS
#undef CB_EPI_FINISH
            } else if (!LEANK && add_kind == 0) {
                // no addend (most layers): a code path of its own WITHOUT any vector-memory load, so that hipcc has no reason to
                // put a wait between the stores (a conditional load makes it wait with vmcnt(0) at the first use of the value)
#pragma unroll
                for (int i = 0; i < NIT; ++i) {
                    int co, nv, col, px;
                    long opix;
                    if (!piece(i, co, opix, nv, col, px)) continue;
                    const float4 a4 = *reinterpret_cast<const float4*>(stage + col * NPIX + px);
                    float v[4] = {a4.x, a4.y, a4.z, a4.w};
                    const float bv = bias_s[m * 32 + col];
                    float* dst = p.y + (long)n * p.out_sN + (long)co * p.out_sC + opix;
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        float t = (v[j] + 0.f) + bv;
                        if (p.act == CCVS_ACT_LRELU) t = lrelu01(t);
                        v[j] = (t + 0.f) * p.out_scale + 0.f;
                    }
                    if (nv == 4 && (reinterpret_cast<uintptr_t>(dst) & 15) == 0) {
                        cb_store16(dst, f32x4{v[0], v[1], v[2], v[3]});
                    } else {
                        _Pragma("unroll") for (int j = 0; j < 4; ++j) if (j < nv) dst[j] = v[j];
                    }
                }
            } else if (!LEANK && add_kind <= 3) {
                // exactly one addend: its pieces are fetched first, then every piece is finished and stored
                float ad[NIT][4];
#pragma unroll
                for (int i = 0; i < NIT; ++i) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) ad[i][j] = 0.f;
                    int co, nv, col, px;
                    long opix;
                    if (!piece(i, co, opix, nv, col, px)) continue;
                    const float* src = add_kind == 1 ? p.pre + (long)(n / p.pre_div) * p.pre_sN + (long)co * p.pre_sC + opix
                                     : add_kind == 2 ? p.res + (long)n * p.res_sN + (long)co * p.res_sC + opix
                                                     : p.y + (long)n * p.out_sN + (long)co * p.out_sC + opix;
                    if (nv == 4 && (reinterpret_cast<uintptr_t>(src) & 15) == 0) {
                        const float4 t = *reinterpret_cast<const float4*>(src);
                        ad[i][0] = t.x; ad[i][1] = t.y; ad[i][2] = t.z; ad[i][3] = t.w;
                    } else {
                        _Pragma("unroll") for (int j = 0; j < 4; ++j) if (j < nv) ad[i][j] = src[j];
                    }
                }
#pragma unroll
                for (int i = 0; i < NIT; ++i) {
                    int co, nv, col, px;
                    long opix;
                    if (!piece(i, co, opix, nv, col, px)) continue;
                    const float4 a4 = *reinterpret_cast<const float4*>(stage + col * NPIX + px);
                    float v[4] = {a4.x, a4.y, a4.z, a4.w};
                    const float bv = bias_s[m * 32 + col];
                    float* dst = p.y + (long)n * p.out_sN + (long)co * p.out_sC + opix;
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        float t = (v[j] + (add_kind == 1 ? ad[i][j] : 0.f)) + bv;
                        if (p.act == CCVS_ACT_LRELU) t = lrelu01(t);
                        t = (t + (add_kind == 2 ? ad[i][j] : 0.f)) * p.out_scale;
                        v[j] = t + (add_kind == 3 ? ad[i][j] : 0.f);
                    }
                    if (nv == 4 && (reinterpret_cast<uintptr_t>(dst) & 15) == 0) {
                        cb_store16(dst, f32x4{v[0], v[1], v[2], v[3]});
                    } else {
                        _Pragma("unroll") for (int j = 0; j < 4; ++j) if (j < nv) dst[j] = v[j];
                    }
                }
            } else {
                // piece by piece (several addends at once -- no layer of the models does this -- and the ragged tiles of the
                // register-capped kernels): the form of rounds 1-3
#pragma unroll
                for (int i = 0; i < NIT; ++i) {
                    int co, nv, col, px;
                    long opix;
                    if (!piece(i, co, opix, nv, col, px)) continue;
                    const float4 a4 = *reinterpret_cast<const float4*>(stage + col * NPIX + px);
                    float v[4] = {a4.x, a4.y, a4.z, a4.w};
                    const float bv = bias_s[m * 32 + col];
                    float* dst = p.y + (long)n * p.out_sN + (long)co * p.out_sC + opix;
                    const float* rsrc = p.res ? p.res + (long)n * p.res_sN + (long)co * p.res_sC + opix : nullptr;
                    const float* psrc = p.pre ? p.pre + (long)(n / p.pre_div) * p.pre_sN + (long)co * p.pre_sC + opix : nullptr;
                    const bool vec = (nv == 4) && ((reinterpret_cast<uintptr_t>(dst) & 15) == 0) &&
                                     (!rsrc || (reinterpret_cast<uintptr_t>(rsrc) & 15) == 0) &&
                                     (!psrc || (reinterpret_cast<uintptr_t>(psrc) & 15) == 0);
                    float rv[4] = {0.f, 0.f, 0.f, 0.f}, ov[4] = {0.f, 0.f, 0.f, 0.f}, pv[4] = {0.f, 0.f, 0.f, 0.f};
                    if (vec) {
                        if (psrc) { const float4 t = *reinterpret_cast<const float4*>(psrc); pv[0] = t.x; pv[1] = t.y; pv[2] = t.z; pv[3] = t.w; }
                        if (rsrc) { const float4 t = *reinterpret_cast<const float4*>(rsrc); rv[0] = t.x; rv[1] = t.y; rv[2] = t.z; rv[3] = t.w; }
                        if (p.accumulate) { const float4 t = *reinterpret_cast<const float4*>(dst); ov[0] = t.x; ov[1] = t.y; ov[2] = t.z; ov[3] = t.w; }
                    } else {
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            if (j < nv) {
                                if (psrc) pv[j] = psrc[j];
                                if (rsrc) rv[j] = rsrc[j];
                                if (p.accumulate) ov[j] = dst[j];
                            }
                        }
                    }
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        float t = (v[j] + pv[j]) + bv;
                        if (p.act == CCVS_ACT_LRELU) t = lrelu01(t);
                        t = (t + rv[j]) * p.out_scale;
                        v[j] = t + ov[j];
                    }
                    if (vec) {
                        cb_store16(dst, f32x4{v[0], v[1], v[2], v[3]});
                    } else {
                        _Pragma("unroll") for (int j = 0; j < 4; ++j) if (j < nv) dst[j] = v[j];
                    }
                }
            }
            if (m + 1 < MB) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");   // (LDS only: the stores stay in flight)
            }
        }
    } else if (!producer) {
#pragma unroll
    for (int pp = 0; pp < PP; ++pp) {
        const int pj = (rw * PP + pp) * 32 + (lane & 31);
        const int prow = pj / TW, pcol = pj - prow * TW;
        const int vy = ty * TH + prow, vx = tx * TW + pcol;
        if (vy >= ay.V || vx >= ax.V) continue;
        const int oy = vy * ay.os + ay.oo, ox = vx * ax.os + ax.oo;
        const long opix = (long)oy * p.Wout + ox;
#pragma unroll
        for (int m = 0; m < MB; ++m) {
            // (this path is taken by experiments only -- transposed layers run the synchronous kernel: the form of rounds 1-3)
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
}

// dynamic LDS limit of a kernel; a refusal (static + dynamic LDS beyond the CU's 160 KB) must not pass silently: every
// launch above the 64 KB default would then fail with "invalid argument"
#define CB_SET_LDS(KERNEL, BYTES)                                                                                           \
    do {                                                                                                                    \
        const hipError_t e_ = hipFuncSetAttribute((const void*)KERNEL, hipFuncAttributeMaxDynamicSharedMemorySize, BYTES);   \
        if (e_ != hipSuccess) fprintf(stderr, "ccvs_conv2d_bf16x3: LDS limit of %s refused: %s\n", #KERNEL, hipGetErrorString(e_)); \
    } while (0)

static int xcd_aware_p8() {
    static const int v = getenv("CCVS_CONV_XCD") ? atoi(getenv("CCVS_CONV_XCD")) : 1;
    return v;
}

// workgroups of kernel `fn` that fit one CU (HIP occupancy query, cached per kernel and LDS size)
static int conv_occupancy(const void* fn, int threads, size_t smem_bytes) {
    struct Entry { const void* fn; size_t smem; int occ; };
    static Entry cache[64];
    static int n_cache = 0;
    for (int i = 0; i < n_cache; ++i)
        if (cache[i].fn == fn && cache[i].smem == smem_bytes) return cache[i].occ;
    int occ = 1;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, fn, threads, smem_bytes) != hipSuccess || occ < 1) occ = 1;
    static const int occ_max = getenv("CCVS_CONV_CHUNK_OCC") ? atoi(getenv("CCVS_CONV_CHUNK_OCC")) : 0;  // experiments: cap workgroups per CU
    if (occ_max > 0 && occ > occ_max) occ = occ_max;
    if (n_cache < 64) cache[n_cache++] = Entry{fn, smem_bytes, occ};
    return occ;
}

#define CONV_PT_NOT_TAKEN (-12345)
template <int MB>
static int conv_pt_try(const ConvK& k, const void* wsplit, const void* wktail, int CinG, int gz, hipStream_t st, int pt_on);   // conv2d_bf16_pt.h

template <int TW, int MB>
static int launch_conv_bf16(const ConvK& k_in, const void* wsplit, const void* wktail, int CinG, int halo_h, int halo_w, int ntx_max, int gz, hipStream_t st, int wpc2) {
    constexpr int NT = 32 * MB;
    int plane = halo_h * halo_w;
    // the packed-output epilogue stages a pass [pixel][36 floats]: the allocation must cover it (256- / 512-pixel tiles)
    const size_t p8s2 = k_in.out_p8 ? (size_t)256 * 36 * 4 : 0, p8s4 = k_in.out_p8 ? (size_t)512 * 36 * 4 : 0;
    if (plane > 256 * CB_MAX_E) {
        ccvs_set_error("ccvs_conv2d_bf16x3: halo tile %dx%d too large", halo_h, halo_w);
        return CCVS_ERR_ARG;
    }
    static bool attr_set = false;
    if (!attr_set) {
        CB_SET_LDS((conv2d_bf16x3_kernel<TW, MB, 4>), 159 * 1024);
        CB_SET_LDS((conv2d_bf16x3_kernel<TW, MB, 8>), 159 * 1024);
        CB_SET_LDS((conv2d_bf16x3_pc_kernel<TW, MB, 0>), 159 * 1024);   // (+ the static bias block)
        CB_SET_LDS((conv2d_bf16x3_pc_kernel<TW, MB, -2>), 159 * 1024);   // (+ the static bias block)
        CB_SET_LDS((conv2d_bf16x3_pc_kernel<TW, MB, -8>), 159 * 1024);   // (+ the static bias block)
        CB_SET_LDS((conv2d_bf16x3_pc_kernel<TW, MB, 1>), 159 * 1024);   // (+ the static bias block)
        CB_SET_LDS((conv2d_bf16x3_pc_kernel<TW, MB, 3>), 159 * 1024);   // (+ the static bias block)
        attr_set = true;
    }
    static const int ablate_env = getenv("CCVS_CONV_ABLATE") ? atoi(getenv("CCVS_CONV_ABLATE")) : 0;  // timing experiments only (1: no staging, 2: no MFMA, 4: scalar staging, 128: no weight DMA, 256: no activation staging)
    static const int conv_prio = getenv("CCVS_CONV_PRIO") ? atoi(getenv("CCVS_CONV_PRIO")) : 0;   // s_setprio of the producer / consumer kernels (0-3)
    static const int conv_stagger = getenv("CCVS_CONV_STAGGER") ? atoi(getenv("CCVS_CONV_STAGGER")) : 0;   // experiments: phase stagger of the first workgroups (units of ~3.8 us)
    const int ablate = (ablate_env & 0xffff) | ((conv_prio & 3) << 16) | ((conv_stagger & 63) << 18);
    ConvK k = k_in;
#ifdef CB_STAMPS
    k.dbg = getenv("CCVS_CONV_DBG") ? (long long*)strtoull(getenv("CCVS_CONV_DBG"), nullptr, 0) : nullptr;
    k.dbg_block = getenv("CCVS_CONV_DBG_BLOCK") ? atoi(getenv("CCVS_CONV_DBG_BLOCK")) : 0;
#endif
    if constexpr (TW == 32 && (MB == 4 || MB == 2)) {
        // persistent tiles (conv2d_bf16_pt.h): resident workgroups walk the tiles, a tile's prologue runs under the step loop of the tile before
        const int pt_on = k.pt;   // bit 0: fp32-input 128-channel layers, bit 1: packed-input layers (set per launch by the dispatcher, conv2d_bf16.hip)
        if (pt_on) {
            const int r = conv_pt_try<MB>(k, wsplit, wktail, CinG, gz, st, pt_on);
            if (r != CONV_PT_NOT_TAKEN) return r;
        }
    }
    const dim3 grid3(k.tiles_x * k.tiles_y, k.CoutPad / NT, gz);
    // cu_limit > 0: the tiles go out as consecutive 1-D chunks of cu_limit x (workgroups of this instantiation that fit one
    // CU) workgroups -- launches on one stream run one after the other, so the convolution never holds more than cu_limit
    // CUs and work on another stream (the token loop of the next batch) always finds the remaining ones free.
    long n_chunk = 1, cap = 0;
    const long total = (long)grid3.x * grid3.y * grid3.z;
    static const int xcd_aware = getenv("CCVS_CONV_XCD") ? atoi(getenv("CCVS_CONV_XCD")) : 1;
    auto plan = [&](const void* fn, int threads, size_t smem_bytes) {
        k.nwork = 0; k.work0 = 0;
        k.gx = (int)grid3.x; k.gy = (int)grid3.y;
        k.xcd_chunk = (xcd_aware && total % 8 == 0 && total >= 64) ? (int)(total / 8) : 0;
        n_chunk = 1;
        if (k.cu_limit <= 0) return;
        cap = (long)k.cu_limit * conv_occupancy(fn, threads, smem_bytes);
        if (total <= cap) return;  // fits as it is
        cap -= cap % 8;            // chunks of whole rounds over the XCDs
        if (cap < 8) cap = 8;
        k.nwork = (int)total;
        n_chunk = (total + cap - 1) / cap;
    };
    auto chunk_grid = [&](long c) -> dim3 {
        if (k.nwork == 0) return grid3;
        k.work0 = (int)(c * cap);
        const long n = (total - c * cap < cap) ? total - c * cap : cap;
        k.xcd_chunk = (xcd_aware && n % 8 == 0 && n >= 64) ? (int)(n / 8) : 0;   // within the chunk
        return dim3((unsigned)n);
    };
#define CB_LAUNCH(KERNEL, THREADS, SMEM, ...)                                                    \
    do {                                                                                          \
        plan((const void*)KERNEL, THREADS, SMEM);                                                 \
        for (long c_ = 0; c_ < n_chunk; ++c_) {                                                   \
            const dim3 g_ = chunk_grid(c_);                                                       \
            hipLaunchKernelGGL((KERNEL), g_, dim3(THREADS), SMEM, st, k, __VA_ARGS__);            \
        }                                                                                         \
    } while (0)
    if (k.in_p8) {  // packed input: LDS-DMA staging (validated by the caller: stride 1, not transposed, Cin % 8 == 0)
        const size_t smem_p = (size_t)(2 * 4 * plane + 2 * ntx_max * 4 * NT) * 16;
        if (4 * plane > 8 * 256 || ntx_max > (MB == 1 ? 9 : 3) || smem_p > 156 * 1024) {
            ccvs_set_error("ccvs_conv2d_bf16x3: packed input with a %dx%d halo tile / %d taps per row is not supported", halo_h, halo_w, ntx_max);
            return CCVS_ERR_ARG;
        }
        if constexpr (MB != 4) {   // (128 output channels per workgroup: the written-out form spills; no layer of the models needs it)
            static bool attr_p = false;
            if (!attr_p) {
                CB_SET_LDS((conv2d_bf16x3_pc_kernel<TW, MB, -83>), 159 * 1024);
                attr_p = true;
            }
        }
        if constexpr (TW == 32 && MB == 2) {   // 64 output channels: the 512-pixel tile (see below), packed input
            static const int pp4p = (getenv("CCVS_CONV_PP4") ? atoi(getenv("CCVS_CONV_PP4")) : 1) && !(getenv("CCVS_CONV_P8_WPC2") && atoi(getenv("CCVS_CONV_P8_WPC2")));
            const int th4 = 16, halo_h4 = (th4 - 1) + k.kh, plane4 = halo_h4 * halo_w;
            const size_t smem_4 = (size_t)(2 * 4 * plane4 + 2 * ntx_max * 4 * NT) * 16;
            if (pp4p && k.out_p8 && k.kh == 3 && k.kw == 3 && k.pad == 1 && k.cu_limit <= 0 && k.Hout >= 2 * th4 && 4 * plane4 <= 10 * 256 && smem_4 <= 156 * 1024 &&
                smem_4 >= (size_t)32 * 512 * 4) {
                k.tiles_y = cdiv(k.Hout, th4);
                const dim3 grid4(k.tiles_x * k.tiles_y, k.CoutPad / NT, gz);
                const long total4 = (long)grid4.x * grid4.y * grid4.z;
                k.nwork = 0; k.work0 = 0; k.gx = (int)grid4.x; k.gy = (int)grid4.y;
                k.xcd_chunk = (xcd_aware_p8() && total4 % 8 == 0 && total4 >= 64) ? (int)(total4 / 8) : 0;
                static bool attr4p = false;
                if (!attr4p) {
                    CB_SET_LDS((conv2d_bf16x3_pc_kernel<TW, MB, -83, 4>), 159 * 1024);
                    attr4p = true;
                }
                hipLaunchKernelGGL((conv2d_bf16x3_pc_kernel<TW, MB, -83, 4>), grid4, dim3(512), (smem_4 > p8s4 ? smem_4 : p8s4), st, k, (const uint4*)wsplit, CinG, ntx_max, ablate);
                CCVS_CHECK_LAUNCH("ccvs_conv2d_bf16x3");
                return CCVS_OK;
            }
        }
        bool done3 = false;
        if constexpr (TW == 32 && MB == 2) {
            // two workgroups per CU on packed input (experiment, CCVS_CONV_P8_WPC2): staging is LDS-DMA only, so a second resident
            // workgroup costs no conversion work -- one tile's prologue / epilogue beside the other's step loop
            static const int p8w2 = getenv("CCVS_CONV_P8_WPC2") ? atoi(getenv("CCVS_CONV_P8_WPC2")) : 0;
            const size_t smem_2 = smem_p > p8s2 ? smem_p : p8s2;
            if (p8w2 && k.kh == 3 && k.kw == 3 && k.pad == 1 && smem_2 <= 79 * 1024 && k.cu_limit <= 0) {
                static bool attr2p = false;
                if (!attr2p) {
                    CB_SET_LDS((conv2d_bf16x3_pc_kernel<TW, MB, -83, 2, 2>), 159 * 1024);
                    attr2p = true;
                }
                CB_LAUNCH((conv2d_bf16x3_pc_kernel<TW, MB, -83, 2, 2>), 512, smem_2, (const uint4*)wsplit, CinG, ntx_max, ablate);
                CCVS_CHECK_LAUNCH("ccvs_conv2d_bf16x3");
                return CCVS_OK;
            }
        }
        if constexpr (MB != 4) {
            if (k.kh == 3 && k.kw == 3 && k.pad == 1) {
                CB_LAUNCH((conv2d_bf16x3_pc_kernel<TW, MB, -83>), 512, (smem_p > p8s2 ? smem_p : p8s2), (const uint4*)wsplit, CinG, ntx_max, ablate);
                done3 = true;
            }
        }
        if (!done3) CB_LAUNCH((conv2d_bf16x3_pc_kernel<TW, MB, -8>), 512, (smem_p > p8s2 ? smem_p : p8s2), (const uint4*)wsplit, CinG, ntx_max, ablate);
        CCVS_CHECK_LAUNCH("ccvs_conv2d_bf16x3");
        return CCVS_OK;
    }
    // aligned float4 staging: dense stride-1 rows on 16-byte boundaries, one item per staging thread
    const int xsh = ((-k.pad % 4) + 4) % 4, nq = (xsh + halo_w + 3) / 4;
    const bool vec_ok = !k.transposed && k.stride == 1 && k.Win % 4 == 0 && k.in_sC % 4 == 0 && k.in_sN % 4 == 0 &&
                        (reinterpret_cast<uintptr_t>(k.x) & 15) == 0 && TW >= 16 && halo_h * ((nq + 1) / 2) * 4 <= 256 && ntx_max <= (MB == 1 ? 9 : 3) &&
                        (k.kh == 1 || (k.kh == 3 && k.kw == 3 && k.pad == 1)) && !(ablate & 4);   // (3 x 3: the GEO3 instantiations assume padding 1)
    // packed K tail (ccvs_conv_desc.w_ktail): read by the vectorised 3 x 3 instantiations only
    static const int ktail_on = getenv("CCVS_CONV_KTAIL") ? atoi(getenv("CCVS_CONV_KTAIL")) : 1;
    const int ktail_r = k.Cin % CB_CC;
    const bool kt = ktail_on && wktail && vec_ok && k.kh == 3 && k.kw == 3 && ktail_r >= 1 && ktail_r <= 3 && k.Cin > CB_CC;
    if constexpr (TW == 32 && MB == 2) {
        // the 512-pixel tile (PP = 4, 16 x 32): 64 output channels, VEC staging, at least two tile rows of work.  Measured per
        // shape on a BAIR decode (tools/conv_shape_census.py): 128->64 3x3 at 256^2 253 -> 282 TFLOP/s; the 32-channel layers
        // (MB = 1: four workgroups per CU hide each other's latency with the 256-pixel tile, one with this one) LOSE -- 64->32
        // 204 -> 156, the 1 x 9 heads 159 -> 120 -- and stay on the 256-pixel tile.
        static const int pp4 = getenv("CCVS_CONV_PP4") ? atoi(getenv("CCVS_CONV_PP4")) : 1;
        const int th4 = 16, halo_h4 = (th4 - 1) + k.kh;
        const size_t smem_4 = (size_t)(2 * 4 * halo_h4 * (nq * 4 + 1) + 2 * ntx_max * 4 * NT) * 16;
        if (pp4 && !wpc2 && vec_ok && !k.out_p8 && k.Hout >= 2 * th4 && halo_h4 * ((nq + 1) / 2) * 4 <= 512 && smem_4 <= 156 * 1024 && smem_4 >= (size_t)32 * 512 * 4) {
            k.tiles_y = cdiv(k.Hout, th4);
            const dim3 grid4(k.tiles_x * k.tiles_y, k.CoutPad / NT, gz);
            const long total4 = (long)grid4.x * grid4.y * grid4.z;
            // (one launch over the whole chip; with a CU budget the 256-pixel form below runs in chunks)
            if (k.cu_limit <= 0) {
                k.nwork = 0; k.work0 = 0; k.gx = (int)grid4.x; k.gy = (int)grid4.y;
                k.xcd_chunk = (xcd_aware && total4 % 8 == 0 && total4 >= 64) ? (int)(total4 / 8) : 0;
                static bool attr4 = false;
                if (!attr4) {
                    CB_SET_LDS((conv2d_bf16x3_pc_kernel<TW, MB, 1, 4>), 159 * 1024);   // (+ the static bias block)
                    CB_SET_LDS((conv2d_bf16x3_pc_kernel<TW, MB, 3, 4>), 159 * 1024);   // (+ the static bias block)
                    attr4 = true;
                }
                k.ktail = kt ? ktail_r : 0;
                if (k.kh == 3) hipLaunchKernelGGL((conv2d_bf16x3_pc_kernel<TW, MB, 3, 4>), grid4, dim3(512), smem_4, st, k, (const uint4*)(kt ? wktail : wsplit), CinG, ntx_max, ablate);
                else hipLaunchKernelGGL((conv2d_bf16x3_pc_kernel<TW, MB, 1, 4>), grid4, dim3(512), smem_4, st, k, (const uint4*)wsplit, CinG, ntx_max, ablate);
                CCVS_CHECK_LAUNCH("ccvs_conv2d_bf16x3");
                return CCVS_OK;
            }
            k.tiles_y = k_in.tiles_y;
        }
    }
    if (vec_ok) {
        const size_t smem_v = (size_t)(2 * 4 * halo_h * (nq * 4 + 1) + 2 * ntx_max * 4 * NT) * 16;
        if constexpr (TW == 32 && MB == 2) {
            if (wpc2 && k.kh == 3 && k.kw == 3 && smem_v <= 80 * 1024 && k.cu_limit <= 0) {   // two workgroups per CU (see the kernel)
                static bool attr2 = false;
                if (!attr2) {
                    CB_SET_LDS((conv2d_bf16x3_pc_kernel<TW, MB, 3, 2, 2>), 159 * 1024);   // (+ the static bias block)
                    attr2 = true;
                }
                k.ktail = kt ? ktail_r : 0;
                CB_LAUNCH((conv2d_bf16x3_pc_kernel<TW, MB, 3, 2, 2>), 512, (smem_v > p8s2 ? smem_v : p8s2), (const uint4*)(kt ? wktail : wsplit), CinG, ntx_max, ablate);
                k.ktail = 0;
                CCVS_CHECK_LAUNCH("ccvs_conv2d_bf16x3");
                return CCVS_OK;
            }
        }
        if (smem_v <= 156 * 1024) {
            if (k.kh == 3) {
                k.ktail = kt ? ktail_r : 0;
                CB_LAUNCH((conv2d_bf16x3_pc_kernel<TW, MB, 3>), 512, (smem_v > p8s2 ? smem_v : p8s2), (const uint4*)(kt ? wktail : wsplit), CinG, ntx_max, ablate);
                k.ktail = 0;
            } else CB_LAUNCH((conv2d_bf16x3_pc_kernel<TW, MB, 1>), 512, (smem_v > p8s2 ? smem_v : p8s2), (const uint4*)wsplit, CinG, ntx_max, ablate);
            CCVS_CHECK_LAUNCH("ccvs_conv2d_bf16x3");
            return CCVS_OK;
        }
    }
    const size_t smem_pc = (size_t)(2 * 4 * plane + 2 * ntx_max * 4 * NT) * 16;
    const int nt_min = (k.transposed ? 1 : k.kh);               // fewest tap rows of any parity class
    const int passes = (plane + 255) / 256;
    const bool regs_ok = (ntx_max <= (MB == 1 ? 9 : 3)) && (passes <= nt_min);
    // transposed layers (1-2 taps per row and parity class: little MFMA work per staged tile) are faster on the 8-wave synchronous
    // kernel than on the two-pass producer / consumer form: 117 vs 77 TFLOP/s on 128->128 at 128^2 (CCVS_CONV_ABLATE=1024 keeps the old routing)
    const bool regs_ok2 = (ntx_max <= (MB == 1 ? 9 : 3)) && (passes <= 2 * nt_min) && !(ablate & 32) && (!k.transposed || (ablate & 1024));
    // (8 staging waves for <= 64 output channels were measured slower: the steps are latency- not staging-bound)
    if (smem_pc <= 156 * 1024 && regs_ok) {  // double-buffered producer / consumer form, scalar staging
        CB_LAUNCH((conv2d_bf16x3_pc_kernel<TW, MB, 0>), 512, (smem_pc > p8s2 ? smem_pc : p8s2), (const uint4*)wsplit, CinG, ntx_max, ablate);
        CCVS_CHECK_LAUNCH("ccvs_conv2d_bf16x3");
        return CCVS_OK;
    }
    if (smem_pc <= 156 * 1024 && regs_ok2) {
        CB_LAUNCH((conv2d_bf16x3_pc_kernel<TW, MB, -2>), 512, (smem_pc > p8s2 ? smem_pc : p8s2), (const uint4*)wsplit, CinG, ntx_max, ablate);
        CCVS_CHECK_LAUNCH("ccvs_conv2d_bf16x3");
        return CCVS_OK;
    }
    if (k.out_p8) {
        ccvs_set_error("ccvs_conv2d_bf16x3: packed output is not available for this shape (synchronous kernel)");
        return CCVS_ERR_ARG;
    }
    const size_t smem = (size_t)(4 * plane + ntx_max * 4 * NT) * 16;
    if (smem > 159 * 1024) {
        ccvs_set_error("ccvs_conv2d_bf16x3: %zu bytes of LDS needed", smem);
        return CCVS_ERR_ARG;
    }
    static const int sync_waves = getenv("CCVS_CONV_SYNC_WAVES") ? atoi(getenv("CCVS_CONV_SYNC_WAVES")) : 8;
    if (sync_waves == 8) CB_LAUNCH((conv2d_bf16x3_kernel<TW, MB, 8>), 512, smem, (const uint4*)wsplit, CinG);
    else CB_LAUNCH((conv2d_bf16x3_kernel<TW, MB, 4>), 256, smem, (const uint4*)wsplit, CinG);
    CCVS_CHECK_LAUNCH("ccvs_conv2d_bf16x3");
    return CCVS_OK;
}

