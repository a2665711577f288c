#!/usr/bin/env python3
"""Per-shape timing of the convolution kernel on the shapes of one BAIR decode step
(batch 16, k = 15 context frames -> N = 240 image pairs).  GPU box only.

    python tools/conv_bench.py [--n 240] [--reps 3]
"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ccvs_amd import ops  # noqa: E402

# (name, Cin, Cout, k, stride, transposed, H, N-scale)  N-scale: "pairs" (B*k) or "frames" (B)
SHAPES = [
    ("match0 49->128 3x3 @256", 49, 128, 3, 1, False, 256, "pairs"),
    ("match1 128->64 3x3 @256", 128, 64, 3, 1, False, 256, "pairs"),
    ("match2 64->32 3x3 @256", 64, 32, 3, 1, False, 256, "pairs"),
    ("subpx0 195->128 3x3 @256", 195, 128, 3, 1, False, 256, "pairs"),
    ("heads 32->3 9x9 @256", 32, 3, 9, 1, False, 256, "pairs"),
    ("proj 96->24 1x1 @256", 96, 24, 1, 1, False, 256, "pairs"),
    ("subpx0 195->128 3x3 @128", 195, 128, 3, 1, False, 128, "pairs"),
    ("heads 32->3 9x9 @128", 32, 3, 9, 1, False, 128, "pairs"),
    ("subpx0 387->128 3x3 @64", 387, 128, 3, 1, False, 64, "pairs"),
    ("heads 32->3 5x5 @64", 32, 3, 5, 1, False, 64, "pairs"),
    ("subpx0 771->128 3x3 @16", 771, 128, 3, 1, False, 16, "pairs"),
    ("trunk 128->128 3x3 @256", 128, 128, 3, 1, False, 256, "frames"),
    ("trunk up 128->128 3x3T @128", 128, 128, 3, 2, True, 128, "frames"),
    ("trunk dn 128->128 3x3s2 @257", 128, 128, 3, 2, False, 257, "frames"),
    ("trunk 512->512 3x3 @8", 512, 512, 3, 1, False, 8, "frames"),
    ("rgb 128->3 1x1 @256", 128, 3, 1, 1, False, 256, "frames"),
]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--pairs", type=int, default=240)
    ap.add_argument("--frames", type=int, default=16)
    ap.add_argument("--reps", type=int, default=3)
    ap.add_argument("--precision", type=str, default=None)
    args = ap.parse_args()
    if args.precision:
        ops.CONV_PRECISION = args.precision
    print("conv precision:", ops.CONV_PRECISION)
    torch.manual_seed(0)
    total_t = total_f = 0.0
    print(f"{'shape':34s} {'N':>4s} {'GFLOP':>9s} {'ms':>9s} {'TFLOP/s':>8s}")
    for name, cin, cout, k, stride, tr, h, scale in SHAPES:
        n = args.pairs if scale == "pairs" else args.frames
        x = torch.randn(n, cin, h, h, device="cuda")
        w = torch.randn(cout, cin, k, k, device="cuda")
        b = torch.randn(cout, device="cuda")
        pad = 0 if (tr or stride == 2) else k // 2
        if name.startswith("heads"):   # the model's fused flow/occ heads: 3k-output MFMA convolution + tap sum (ops.conv_heads)
            wp = ops.pack_head_weights(w[:2], w[2:3])
            y = torch.zeros(n, 3, h, h, device="cuda")
            run = lambda: ops.conv_heads(x, wp, b, y, accumulate=True)
        else:
            wp = ops.pack_conv_weight(w)
            y = ops.conv2d(x, wp, b, cout, k, stride=stride, pad=pad, transposed=tr, act=True)
            run = lambda: ops.conv2d(x, wp, b, cout, k, stride=stride, pad=pad, transposed=tr, act=True, out=y)
        run()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(args.reps):
            run()
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / args.reps
        px = h * h if tr else y.shape[2] * y.shape[3]
        gflop = 2.0 * n * cout * cin * k * k * px / 1e9
        total_t += ms
        total_f += gflop
        print(f"{name:34s} {n:4d} {gflop:9.1f} {ms:9.3f} {gflop / ms:8.1f}")
        del x, y
    print(f"{'sum':34s} {'':4s} {total_f:9.1f} {total_t:9.3f} {total_f / total_t:8.1f}")


if __name__ == "__main__":
    main()
