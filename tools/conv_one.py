#!/usr/bin/env python3
"""One convolution shape, a few launches -- the target of rocprofv3 --pmc runs.
    python tools/conv_one.py Cin Cout k H N [precision] [p8] [pre=K]     (p8: packed split-bf16 input and output; pre=K: with a
    pre-activation addend [N / K, Cout, H, H] shared by K consecutive images, as the first Subpixel convolution has)"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ccvs_amd import ops  # noqa: E402

cin, cout, k, h, n = [int(v) for v in sys.argv[1:6]]
pre_div = max([int(a[4:]) for a in sys.argv[6:] if a.startswith("pre=")] + [0])    # pre=15: a pre-activation addend shared by 15 images (Subpixel conv0)
sys.argv = [a for a in sys.argv if not a.startswith("pre=")]
p8 = "p8" in sys.argv[6:]
p8out = "p8out" in sys.argv[6:]      # fp32 input, packed output (the first convolution of a Matching / Subpixel stack)
if len(sys.argv) > 6 and sys.argv[6] not in ("p8", "p8out"):
    ops.CONV_PRECISION = sys.argv[6]
torch.manual_seed(0)
x = torch.randn(n, cin, h, h, device="cuda")
w = torch.randn(cout, cin, k, k, device="cuda")
b = torch.randn(cout, device="cuda")
wp = ops.pack_conv_weight(w)
if p8:   # a packed input: the output of an identity 1x1 convolution
    eye = torch.eye(cin, device="cuda").view(cin, cin, 1, 1) * (cin ** 0.5)
    x = ops.conv2d(x, ops.pack_conv_weight(eye), None, cin, 1, out_p8=True)
kw = dict(out_p8=True) if (p8 or p8out) else {}
kw_pre = dict(pre=torch.randn(n // pre_div, cout, h, h, device="cuda"), pre_div=pre_div) if pre_div else {}
kw.update(kw_pre)
y = ops.conv2d(x, wp, b, cout, k, pad=k // 2, act=True, **kw)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(3):
    ops.conv2d(x, wp, b, cout, k, pad=k // 2, act=True, **(kw if (p8 or p8out) else dict(kw_pre, out=y)))
e1.record()
torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 3
print(f"{cin}->{cout} k{k} @{h} N={n}: {ms:.3f} ms, {2.0 * n * cout * cin * k * k * h * h / ms / 1e9:.1f} TFLOP/s")
if os.environ.get("CCVS_CONV_ONE_SUM"):   # two position-dependent checksums of the output's words: A / B runs of two builds or switches must print the same pair
    words = (y.data if hasattr(y, "data") and not torch.is_tensor(y) else y).reshape(-1).view(torch.int32)
    s0 = s1 = 0
    step = 1 << 26
    for i in range(0, words.numel(), step):
        c = words[i:i + step].to(torch.int64)
        s0 += int(c.sum())
        s1 += int((c * (torch.arange(c.numel(), device=c.device) % 8191 + 1)).sum())
    print(f"   checksum {s0 & 0xffffffffffff:012x} {s1 & 0xffffffffffff:012x}")
