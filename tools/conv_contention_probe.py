#!/usr/bin/env python3
"""What does a convolution launch lose to when the token loops run beside it?  rocprofv3 --pmc serialises dispatches, so
the mechanism is isolated by experiment instead: the decoder's main convolution shapes (and two HBM-bound kernels) are
timed alone and beside ONE kind of background load at a time (tools/micro/hogs.hip: workgroups with the footprint of the
decode kernels -- 256 threads, <= 48 VGPRs -- on a high-priority stream, like the token loops):
   idle (occupancy only) | HBM stream, default policy (evicts L2 / Infinity Cache) | HBM stream, non-temporal |
   L2-resident re-reads (L2 bandwidth, nothing evicted) | VALU spin (issue slots) | LDS spin.
Prints ms per launch, the slow-down against the run alone and what the hog moved meanwhile.
   python tools/conv_contention_probe.py [images]          (needs tools/micro/libhogs.so)"""
import ctypes
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from ccvs_amd import ops  # noqa: E402

hogs = ctypes.CDLL(os.path.join(ROOT, "tools", "micro", "libhogs.so"))
hogs.hog_launch.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_double, ctypes.c_void_p, ctypes.c_long, ctypes.c_void_p, ctypes.c_void_p,
                            ctypes.c_void_p]
N = int(sys.argv[1]) if len(sys.argv) > 1 else 120
torch.manual_seed(0)
big = torch.empty(6 << 30, dtype=torch.uint8, device="cuda")
big.view(torch.float32)[: 1 << 20].zero_()
count = torch.zeros(1, dtype=torch.int64, device="cuda")
sink = torch.zeros(4, device="cuda")
s_hog = torch.cuda.Stream(priority=-1)


def make_conv(cin, cout, k, kw, h):
    x = torch.randn(N, cin, h, h, device="cuda")
    w = torch.randn(cout, cin, k, kw, device="cuda")
    b = torch.randn(cout, device="cuda")
    wp = ops.pack_conv_weight(w)
    pad = kw // 2 if k == 1 else k // 2
    y = ops.conv2d(x, wp, b, cout, k, pad=pad, act=True)
    flops = 2.0 * N * cout * cin * k * kw * y.shape[2] * y.shape[3]
    return (lambda: ops.conv2d(x, wp, b, cout, k, pad=pad, act=True, out=y)), flops, 0.0


def make_backwarp(c, h):
    x = torch.randn(N, c, h, h, device="cuda")
    flow = torch.randn(N, 2, h, h, device="cuda") * 1.5
    y = torch.empty_like(x)
    return (lambda: ops.backwarp(x, flow, 1.0, out=y)), 0.0, 8.0 * x.numel()


def make_blur(c, h):
    x = torch.randn(N, c, h, h, device="cuda")
    return (lambda: ops.upfirdn2d(x, pad=(2, 1))), 0.0, 8.0 * x.numel()


cases = [("195->128 3x3 @256", make_conv(195, 128, 3, 3, 256)), ("128->64 3x3 @256", make_conv(128, 64, 3, 3, 256)),
         ("64->32 3x3 @256", make_conv(64, 32, 3, 3, 256)), ("32->27 1x9 @256", make_conv(32, 27, 1, 9, 256)),
         ("backwarp 96ch @256", make_backwarp(96, 256)), ("blur 128ch @256", make_blur(128, 256))]
# (mode, workgroups, label, bytes per counted operation per thread)
hog_kinds = [(None, 0, "alone", 0), (0, 256, "idle x256", 0), (0, 1024, "idle x1024", 0),
             (1, 256, "HBM stream x256", 16), (1, 1024, "HBM stream x1024", 16), (2, 256, "HBM stream nt x256", 16), (2, 1024, "HBM stream nt x1024", 16),
             (3, 256, "L2 re-read x256", 16), (3, 1024, "L2 re-read x1024", 16), (4, 256, "VALU spin x256", 0), (4, 1024, "VALU spin x1024", 0),
             (5, 256, "LDS spin x256", 16), (5, 1024, "LDS spin x1024", 16)]
REPS = 4
if len(sys.argv) > 2:     # second argument: comma-separated hog labels to keep (the run alone always stays)
    keep = sys.argv[2].split(",")
    hog_kinds = [h for h in hog_kinds if h[0] is None or h[2] in keep]
if len(sys.argv) > 3:     # third: case-name prefixes to keep
    cases = [c for c in cases if any(c[0].startswith(p) for p in sys.argv[3].split(","))]


def timed(fn):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(REPS):
        fn()
    e1.record()
    e1.synchronize()
    return e0.elapsed_time(e1) / REPS


print(f"{N} images per launch; hog workgroups: 256 threads, <= 48 VGPRs, high-priority stream", flush=True)
for name, (fn, flops, nbytes) in cases:
    fn()
    torch.cuda.synchronize()
    base = None
    for mode, n_wg, label, bpo in hog_kinds:
        torch.cuda.synchronize()
        if mode is not None:
            count.zero_()
            torch.cuda.synchronize()
            secs = max(0.05, 2.2 * REPS * (base or 10.0) * 1e-3 * 3)
            t_h = time.time()
            rc = hogs.hog_launch(mode, n_wg, secs, ctypes.c_void_p(big.data_ptr()), big.numel(), ctypes.c_void_p(count.data_ptr()),
                                 ctypes.c_void_p(sink.data_ptr()), ctypes.c_void_p(s_hog.cuda_stream))
            assert rc == 0, rc
            time.sleep(0.01)
        ms = timed(fn)
        rate = ""
        if mode is not None:
            torch.cuda.synchronize()           # the hog runs out its time
            ops_per_thread = int(count.item())
            if bpo:
                rate = f"   hog moved {ops_per_thread * 256 * bpo / secs / 1e12:5.2f} TB/s"
        else:
            base = ms
        perf = f"{flops / ms / 1e9:6.1f} TF/s" if flops else f"{nbytes / ms / 1e9:6.2f} TB/s"
        print(f"{name:20s} {label:22s} {ms:8.3f} ms  x{ms / base:5.2f}  {perf}{rate}", flush=True)
