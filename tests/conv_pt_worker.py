"""Worker of tests/test_ops_gpu.py::test_conv_persistent_tiles_are_bit_identical: the 3 x 3 layers the persistent-tile kernel takes
(conv2d_bf16_pt.h) under whatever CCVS_CONV_PT the parent set (the library reads the switch once per process), results to the .npz
named on the command line.  Every launch has >= 2 x 256 tiles (fewer keep the producer / consumer kernel) and an odd tile count per
workgroup somewhere (the two register sets of the staging stream change roles from tile to tile)."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ccvs_amd import ops  # noqa: E402

ops.CONV_PRECISION = "bf16x3"
g = torch.Generator().manual_seed(23)
out = {}


def rnd(*shape, scale=1.0):
    return (torch.randn(*shape, generator=g) * scale).cuda()


def layer(cin, cout):
    w = rnd(cout, cin, 3, 3, scale=(9 * cin) ** -0.5)
    return ops.pack_conv_weight(w), rnd(cout)


# fp32 input, 128 output channels: 99 (K tail r = 3), 49 (r = 1), 195, 96 (no tail), 128; 64 x 64 images (16 tiles of 8 x 32, 8 of 16 x 32 each), 72 images = 1152 / 576 tiles: 4.5 / 2.25 per workgroup
for cin in (99, 49, 195, 96, 128):
    x = rnd(72, cin, 64, 64)
    wp, b = layer(cin, 128)
    out[f"f32_{cin}"] = ops.conv2d(x, wp, b, 128, 3, pad=1, act=True)
    y8 = ops.conv2d(x, wp, b, 128, 3, pad=1, act=True, out_p8=True)                       # packed output
    out[f"p8out_{cin}"] = y8.data
    pre = rnd(24, 128, 64, 64)
    out[f"pre_{cin}"] = ops.conv2d(x, wp, b, 128, 3, pad=1, act=True, pre=pre, pre_div=3)  # shared pre-activation image: the (group, tile, image) order
    out[f"pre8_{cin}"] = ops.conv2d(x, wp, b, 128, 3, pad=1, act=True, pre=pre, pre_div=3, out_p8=True).data
    res = rnd(72, 128, 64, 64)
    out[f"res_{cin}"] = ops.conv2d(x, wp, b, 128, 3, pad=1, act=True, residual=res, out_scale=0.5 ** 0.5)
    acc = rnd(72, 128, 64, 64)
    ops.conv2d(x, wp, b, 128, 3, pad=1, act=False, out=acc, accumulate=True)
    out[f"acc_{cin}"] = acc
    # a chain through packed activations: 128 -> 64 (512-pixel tile, packed in and out), and 128 -> 128 packed in
    wp2, b2 = layer(128, 64)
    z8 = ops.conv2d(y8, wp2, b2, 64, 3, pad=1, act=True, out_p8=True)
    out[f"chain64_{cin}"] = z8.data
    wp3, b3 = layer(128, 128)
    out[f"chain128_{cin}"] = ops.conv2d(y8, wp3, b3, 128, 3, pad=1, act=True)
# 256 output channels (two channel blocks per tile position), 200 input channels, 32-row images of 64 columns
x = rnd(40, 200, 32, 64)
wp, b = layer(200, 256)
out["f32_200_256"] = ops.conv2d(x, wp, b, 256, 3, pad=1, act=True)
# shapes the persistent form must refuse (ragged width, 24 rows with the 16-row tile, too few tiles): the old kernels, trivially equal
x = rnd(36, 99, 64, 48)
wp, b = layer(99, 128)
out["ragged"] = ops.conv2d(x, wp, b, 128, 3, pad=1, act=True)
x = rnd(4, 99, 64, 64)
out["few"] = ops.conv2d(x, wp, b, 128, 3, pad=1, act=True)
torch.cuda.synchronize()
np.savez(sys.argv[1], **{k: v.cpu().numpy() for k, v in out.items()})
