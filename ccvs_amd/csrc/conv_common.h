// Shared by the fp32-MFMA and the split-bf16 convolution kernels: the kernel argument block and the
// per-axis tap table that turns padding, stride and the stride-2 transposed form into one loop.
#pragma once
#include "common.h"

struct ConvK {
    const float* x;
    const float* w;
    const float* bias;
    const float* res;
    float* y;
    int N, Cin, Hin, Win;
    long in_sN, in_sC;
    int Cout, CoutPad, Hout, Wout;
    long out_sN, out_sC, res_sN, res_sC;
    int kh, kw, stride, pad, transposed;
    int act, accumulate;
    float out_scale;
    const float* pre;
    long pre_sN, pre_sC;
    int pre_div;
    int tiles_x, tiles_y;
    int in_p8, out_p8;  // split-bf16 packed activations (ccvs_hip.h: ccvs_conv_desc)
    // CU-limited form (ccvs_conv_desc.cu_limit > 0): the nwork = gx * gy * gz tiles (x fastest, like the 3-D grid's dispatch
    // order) are launched as consecutive 1-D chunks of at most cu_limit x occupancy workgroups; a chunk starts at tile work0
    int nwork, gx, gy, work0;
    int xcd_chunk;  // XCD-aware order of the tiles of one launch (see CONV_TILE_COORDS)
    int cu_limit;  // host side only: CUs this launch may occupy (0 = classic 3-D grid over all of them)
    int zi;        // > 1: the zi images that share one `pre` image run back to back per tile (see CONV_TILE_COORDS)
    int ktail;     // r = Cin % 16 in {1, 2, 3} and the weights are in the packed-tail form (ccvs_conv_desc.w_ktail); else 0
    int pt;        // host side only: persistent tiles for this launch (conv2d_bf16_pt.h): bit 0 fp32-input 128-channel layers, bit 1 packed-input layers
#ifdef CB_STAMPS   // debug build only (make EXTRA=-DCB_STAMPS; tools/r05/conv_stamps.py): s_memtime stamps of one MFMA wave's steps
    long long* dbg;
    int dbg_block;
#endif
};

// Staging mode of a producer / consumer instantiation (template parameter NTY of conv2d_bf16x3_pc_kernel, described at the kernel):
// which modes fetch their activations 16 bytes per lane (aligned global_load_dwordx4 or LDS-DMA) and which dword by dword.  One
// definition for the kernel and for ccvs_conv_fetch_bytes_per_lane (the HBM-counter scripts under tools/ ask the LIBRARY how an
// instantiation reads instead of keeping a list of their own: on gfx950 FETCH_SIZE tallies a 16-byte-per-lane stream at half its
// bytes, MI355X_MICROARCH.md "HBM").
constexpr bool conv_nty_vec(int nty) { return nty == 1 || nty == 3; }        // fp32 rows by dwordx4, weights by LDS-DMA
constexpr bool conv_nty_p8(int nty) { return nty == -8 || nty == -83; }      // packed split-bf16 input: everything by LDS-DMA
constexpr bool conv_nty_scalar(int nty) { return nty == 0 || nty == -2; }    // halo elements dword by dword
constexpr int conv_nty_fetch_bytes(int nty) { return (conv_nty_vec(nty) || conv_nty_p8(nty)) ? 16 : (conv_nty_scalar(nty) ? 4 : 0); }

// Tile coordinates of a workgroup.  Workgroups are handed to the 8 XCDs round-robin in dispatch order (linear id % 8), and
// each XCD has its own L2: with the identity mapping the 8 neighbours of a tile -- whose halo rows and columns it shares --
// are all fetched through OTHER L2s.  The linear id is therefore re-mapped so that XCD j walks the j-th contiguous eighth
// of the tile sequence (x fastest within an image): neighbouring tiles then meet in the same L2, at about the same time.
// (xcd_chunk = tiles / 8 when that divides, else 0 = identity; correctness does not depend on the placement.)
// zi > 1 (a launch whose epilogue adds `pre[n / pre_div]`, zi = pre_div images per `pre` image): the sequence is
// (image group, tile, image of the group) with the image fastest, so the zi workgroups that read one tile of `pre` run back to
// back on one XCD and 14 of 15 of those reads hit its L2 -- image-major, they were a whole image of tiles (tens of MB) apart and
// every one of them went to HBM: 8 GB of the 22 GB the first Subpixel convolution of a 256 x 256 frame moved.
#define CONV_TILE_COORDS(p, bx, by, bz)                                         \
    int bx, by, bz;                                                             \
    {                                                                           \
        int w_ = (p).nwork > 0 ? (int)blockIdx.x                                \
                               : (int)(blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z)); \
        if ((p).xcd_chunk > 0) w_ = (w_ & 7) * (p).xcd_chunk + (w_ >> 3);       \
        w_ += (p).work0;                                                        \
        if ((p).zi > 1) {                                                       \
            const int per_ = (p).gx * (p).gy * (p).zi;                          \
            const int g_ = w_ / per_, q_ = w_ - g_ * per_;                      \
            const int t_ = q_ / (p).zi;                                         \
            bz = g_ * (p).zi + (q_ - t_ * (p).zi);                              \
            bx = t_ % (p).gx;                                                   \
            by = t_ / (p).gx;                                                   \
        } else {                                                                \
            bx = w_ % (p).gx;                                                   \
            const int r_ = w_ / (p).gx;                                         \
            by = r_ % (p).gy;                                                   \
            bz = r_ / (p).gy;                                                   \
        }                                                                       \
    }

struct AxisTaps {
    int nt;       // number of taps along this axis
    int d0, dd;   // input offset of tap a: d0 + a*dd
    int w0, dw;   // weight index of tap a along this axis: w0 + a*dw
    int s;        // virtual -> input stride
    int os, oo;   // virtual -> output: o = v*os + oo
    int V;        // virtual extent
    int lo, ext;  // min offset, halo extent (hi - lo)
};

__device__ __forceinline__ AxisTaps axis_taps(int k, int stride, int pad, int transposed, int parity, int out_extent) {
    AxisTaps t;
    if (transposed) {
        // conv_transpose2d stride 2 pad 0:  o = 2*i + kk.  Output parity class `parity`
        // uses taps kk = parity, parity+2, ... reading input i = v - a.
        t.nt = (k - parity + 1) / 2;
        t.d0 = 0; t.dd = -1; t.w0 = parity; t.dw = 2; t.s = 1; t.os = 2; t.oo = parity;
        t.V = (out_extent - parity + 1) / 2;
        t.lo = -(t.nt - 1); t.ext = t.nt - 1;
    } else {
        t.nt = k; t.d0 = -pad; t.dd = 1; t.w0 = 0; t.dw = 1; t.s = stride; t.os = 1; t.oo = 0;
        t.V = out_extent; t.lo = -pad; t.ext = k - 1;
    }
    return t;
}

// the tap table of a 3-tap axis with stride 1 and padding 1 (every field but the extent a constant)
__device__ __forceinline__ AxisTaps axis_taps_k3(int out_extent) {
    AxisTaps t;
    t.nt = 3; t.d0 = -1; t.dd = 1; t.w0 = 0; t.dw = 1; t.s = 1; t.os = 1; t.oo = 0;
    t.V = out_extent; t.lo = -1; t.ext = 2;
    return t;
}
