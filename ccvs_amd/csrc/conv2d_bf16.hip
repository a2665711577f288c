// Entry point of the split-bf16 convolution (kernels: conv2d_bf16_kernels.h).  The (tile width, channel block)
// instantiations are compiled as separate translation units (conv2d_bf16_inst.hip, one object per pair: they build in
// parallel); this file validates the call, plans the tiling and dispatches.
#include "common.h"
#include "conv_common.h"
#include <stdlib.h>
#include <string.h>
#include <stdio.h>

#define CB_DECL(TWv, MBv) int ccvs_conv_bf16_launch_##TWv##_##MBv(const ConvK& k, const void* wsplit, const void* wktail, int CinG, int halo_h, int halo_w, int ntx_max, int gz, hipStream_t st, int wpc2);
CB_DECL(32, 4) CB_DECL(32, 2) CB_DECL(32, 1) CB_DECL(16, 4) CB_DECL(16, 2) CB_DECL(16, 1) CB_DECL(8, 4) CB_DECL(8, 2) CB_DECL(8, 1)
#undef CB_DECL

// How many bytes per lane the ACTIVATION fetches of a convolution kernel of this library read at a time, by the kernel's name as
// rocprofv3 prints it (`conv2d_bf16x3_pc_kernel<32, 2, -83, 4, 1>`): 16 (aligned dwordx4 rows / LDS-DMA), 4 (dword by dword: the
// scalar staging modes, the synchronous split-bf16 kernel and the fp32-MFMA kernel), -1 for a name this library does not know --
// the caller must then stop instead of guessing.  Pure host code (no GPU call): tools/pmc_widths.py and its CPU test use it to
// apply gfx950's FETCH_SIZE correction (a 16-byte-per-lane stream is tallied at half its bytes) per instantiation.
extern "C" int ccvs_conv_fetch_bytes_per_lane(const char* kernel_name) {
    if (!kernel_name) return -1;
    const char* pc = strstr(kernel_name, "conv2d_bf16x3_pc_kernel<");
    if (pc) {
        int tw = 0, mb = 0, nty = 0;
        if (sscanf(pc + strlen("conv2d_bf16x3_pc_kernel<"), " %d , %d , %d", &tw, &mb, &nty) != 3) return -1;
        const int b = conv_nty_fetch_bytes(nty);
        return b ? b : -1;
    }
    if (strstr(kernel_name, "conv2d_bf16x3_pt_kernel<")) return 16;   // persistent tiles (conv2d_bf16_pt.h): aligned dwordx4 rows or LDS-DMA, nothing else
    if (strstr(kernel_name, "conv2d_bf16x3_kernel<") || strstr(kernel_name, "conv2d_mfma_kernel")) return 4;
    return -1;
}

// Persistent tiles (conv2d_bf16_pt.h) by CONTEXT: alone on the chip the resident form is 6 % faster over a decode's convolutions (a single
// generate call: +1...1.5 %), beside the token loops of other batches the chip is power-managed and the denser kernel costs everyone clock
// (profiles/r06_conv_pt_bench_ab.txt) -- so the host switches it: 1 by default, 0 while a pipelined run has several batches in flight
// (helpers/pipeline.py).  CCVS_CONV_PT in the environment overrides both.
static int g_conv_pt_mode = 1;
extern "C" int ccvs_conv_persistent_tiles(int32_t mode) {
    const int prev = __atomic_exchange_n(&g_conv_pt_mode, mode < 0 ? __atomic_load_n(&g_conv_pt_mode, __ATOMIC_RELAXED) : (int)(mode & 3), __ATOMIC_RELAXED);
    return prev;
}

extern "C" int ccvs_conv2d_bf16x3(const float* x, const void* w_split, const float* bias, const float* residual, float* y,
                                  const ccvs_conv_desc* d, void* stream) {
    CCVS_REQUIRE(x && w_split && y && d, "ccvs_conv2d_bf16x3: null pointer");
    CCVS_REQUIRE(d->N > 0 && d->Cin > 0 && d->Cout > 0 && d->Hin > 0 && d->Win > 0, "ccvs_conv2d_bf16x3: empty tensor");
    CCVS_REQUIRE((d->kh == d->kw || d->kw == 1 || d->kh == 1) && d->kh >= 1 && d->kh <= 9 && d->kw >= 1 && d->kw <= 9,
                 "ccvs_conv2d_bf16x3: kernel %dx%d unsupported", d->kh, d->kw);
    CCVS_REQUIRE(d->CoutPad % 32 == 0 && d->CoutPad >= d->Cout, "ccvs_conv2d_bf16x3: CoutPad %d invalid for Cout %d", d->CoutPad, d->Cout);
    int Hout, Wout;
    if (d->transposed) {
        CCVS_REQUIRE(d->kh == 3 && d->stride == 2 && d->pad == 0, "ccvs_conv2d_bf16x3: transposed supports k3 s2 p0 only");
        Hout = 2 * d->Hin + d->kh - 2;
        Wout = 2 * d->Win + d->kw - 2;
    } else {
        CCVS_REQUIRE(d->stride == 1 || d->stride == 2, "ccvs_conv2d_bf16x3: stride %d unsupported", d->stride);
        Hout = (d->Hin + 2 * d->pad - d->kh) / d->stride + 1;
        Wout = (d->Win + 2 * d->pad - d->kw) / d->stride + 1;
    }
    CCVS_REQUIRE(Hout == d->Hout && Wout == d->Wout, "ccvs_conv2d_bf16x3: output %dx%d expected, got %dx%d", Hout, Wout, d->Hout, d->Wout);

    ConvK k;
    k.x = x; k.w = nullptr; k.bias = bias; k.res = residual; k.y = y;
    k.N = d->N; k.Cin = d->Cin; k.Hin = d->Hin; k.Win = d->Win; k.in_sN = d->in_sN; k.in_sC = d->in_sC;
    k.Cout = d->Cout; k.CoutPad = d->CoutPad; k.Hout = d->Hout; k.Wout = d->Wout;
    k.out_sN = d->out_sN; k.out_sC = d->out_sC; k.res_sN = d->res_sN; k.res_sC = d->res_sC;
    k.kh = d->kh; k.kw = d->kw; k.stride = d->stride; k.pad = d->pad; k.transposed = d->transposed ? 1 : 0;
    k.act = d->act; k.accumulate = d->accumulate; k.out_scale = d->out_scale;
    k.pre = d->pre; k.pre_sN = d->pre_sN; k.pre_sC = d->pre_sC; k.pre_div = d->pre_div > 0 ? d->pre_div : 1;
    k.in_p8 = d->in_p8 ? 1 : 0; k.out_p8 = d->out_p8 ? 1 : 0;
    static const int pt_env = getenv("CCVS_CONV_PT") ? atoi(getenv("CCVS_CONV_PT")) : -1;
    k.pt = pt_env >= 0 ? pt_env : __atomic_load_n(&g_conv_pt_mode, __ATOMIC_RELAXED);
    k.ktail = 0; k.nwork = 0; k.gx = k.gy = 1; k.work0 = 0; k.xcd_chunk = 0; k.cu_limit = d->cu_limit > 0 ? d->cu_limit : ccvs_cu_limit_of(stream);
    static const int pre_order = getenv("CCVS_CONV_PRE_ORDER") ? atoi(getenv("CCVS_CONV_PRE_ORDER")) : 1;   // 0: image-major tiles for every launch
    k.zi = (pre_order && d->pre && k.pre_div > 1 && !d->transposed && d->N % k.pre_div == 0) ? k.pre_div : 0;
    if (k.in_p8) CCVS_REQUIRE(!d->transposed && d->stride == 1 && d->Cin % 8 == 0, "ccvs_conv2d_bf16x3: packed input needs stride 1, Cin %% 8 == 0");
    if (k.out_p8) CCVS_REQUIRE(!d->transposed && d->Cout % 8 == 0 && !d->accumulate && !residual, "ccvs_conv2d_bf16x3: packed output needs Cout %% 8 == 0, no residual / accumulate");
    const int CinG = 2 * ((d->Cin + 15) / 16);  // 8-channel groups, Cin padded to 16

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
    CCVS_REQUIRE(gz <= 65535, "ccvs_conv2d_bf16x3: batch %d too large for one launch", d->N);
    hipStream_t st = (hipStream_t)stream;
    int mb = (d->CoutPad % 128 == 0) ? 4 : ((d->CoutPad % 64 == 0) ? 2 : 1);
    // Small launches (the 8 x 8 ... 32 x 32 levels of the encoder / decoder at 16 images: one or four pixel tiles per image):
    // with 128 output channels per workgroup they are 16-64 workgroups walking the whole K depth alone on a quarter of the
    // chip.  Narrower channel blocks give the same tiles to 2-4 x as many workgroups (same arithmetic per output, bit-identical).
    static const int small_split = getenv("CCVS_CONV_SMALL_SPLIT") ? atoi(getenv("CCVS_CONV_SMALL_SPLIT")) : 256;   // workgroups aimed at; 0: off
    while (mb > 1 && (long)k.tiles_x * k.tiles_y * gz * (d->CoutPad / (32 * mb)) < small_split) mb >>= 1;
    // Few input channels per output byte (the 49- and 99-channel layers in front of 128 outputs): 64 output channels per
    // workgroup and two workgroups per CU, so that one tile's prologue / epilogue runs beside the other's K loop
    // (conv2d_bf16_kernels.h, WPC).  CCVS_CONV_WPC2 = largest Cin that takes this form (0: off).
    static const int wpc2_cin = getenv("CCVS_CONV_WPC2") ? atoi(getenv("CCVS_CONV_WPC2")) : 64;   // 49->128: +6...7 %; 99->128: none (tools/conv_one.py)
    int wpc2 = 0;
    static const int wpc2_p8out = getenv("CCVS_CONV_WPC2_P8OUT") ? atoi(getenv("CCVS_CONV_WPC2_P8OUT")) : 0;   // experiment: ... also when the layer writes packed output
    if (wpc2_cin > 0 && mb >= 2 && TW == 32 && d->Cin <= wpc2_cin && d->kh == 3 && d->kw == 3 && d->stride == 1 && !d->transposed && !d->in_p8 && (!d->out_p8 || wpc2_p8out)) {
        mb = 2;
        wpc2 = 1;
    }
    // experiment (CCVS_CONV_P8_WPC2): packed-input 3 x 3 layers as 64-channel workgroups, two per CU (see launch_conv_bf16)
    static const int p8_wpc2 = getenv("CCVS_CONV_P8_WPC2") ? atoi(getenv("CCVS_CONV_P8_WPC2")) : 0;
    if (p8_wpc2 && mb == 4 && TW == 32 && d->in_p8 && d->kh == 3 && d->kw == 3 && d->stride == 1 && d->pad == 1 && k.cu_limit <= 0) mb = 2;
#define CB_DISPATCH(TWv)                                                                                        \
    if (mb == 4) return ccvs_conv_bf16_launch_##TWv##_4(k, w_split, d->w_ktail, CinG, halo_h, halo_w, ntx_max, gz, st, wpc2);           \
    if (mb == 2) return ccvs_conv_bf16_launch_##TWv##_2(k, w_split, d->w_ktail, CinG, halo_h, halo_w, ntx_max, gz, st, wpc2);           \
    return ccvs_conv_bf16_launch_##TWv##_1(k, w_split, d->w_ktail, CinG, halo_h, halo_w, ntx_max, gz, st, wpc2);
    if (TW == 32) { CB_DISPATCH(32) }
    if (TW == 16) { CB_DISPATCH(16) }
    CB_DISPATCH(8)
#undef CB_DISPATCH
}
