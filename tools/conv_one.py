#!/usr/bin/env python3
"""One convolution shape, a few launches -- the target of rocprofv3 --pmc runs.
    python tools/conv_one.py Cin Cout k H N [precision]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ccvs_amd import ops  # noqa: E402

cin, cout, k, h, n = [int(v) for v in sys.argv[1:6]]
if len(sys.argv) > 6:
    ops.CONV_PRECISION = sys.argv[6]
torch.manual_seed(0)
x = torch.randn(n, cin, h, h, device="cuda")
w = torch.randn(cout, cin, k, k, device="cuda")
b = torch.randn(cout, device="cuda")
wp = ops.pack_conv_weight(w)
y = ops.conv2d(x, wp, b, cout, k, pad=k // 2, act=True)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(3):
    ops.conv2d(x, wp, b, cout, k, pad=k // 2, act=True, out=y)
e1.record()
torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 3
print(f"{cin}->{cout} k{k} @{h} N={n}: {ms:.3f} ms, {2.0 * n * cout * cin * k * k * h * h / ms / 1e9:.1f} TFLOP/s")
