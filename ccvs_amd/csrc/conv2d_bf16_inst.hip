// One (tile width CB_TW, channel block CB_MB) instantiation of the split-bf16 convolution kernels and their launcher
// (compiled once per pair by the Makefile: -DCB_TW=.. -DCB_MB=..).
#include "conv2d_bf16_kernels.h"
#include "conv2d_bf16_pt.h"

#define CB_CAT_(a, b, c) ccvs_conv_bf16_launch_##a##_##b
#define CB_CAT(a, b) CB_CAT_(a, b, 0)
int CB_CAT(CB_TW, CB_MB)(const ConvK& k, const void* wsplit, const void* wktail, int CinG, int halo_h, int halo_w, int ntx_max, int gz, hipStream_t st, int wpc2) {
    return launch_conv_bf16<CB_TW, CB_MB>(k, wsplit, wktail, CinG, halo_h, halo_w, ntx_max, gz, st, wpc2);
}
