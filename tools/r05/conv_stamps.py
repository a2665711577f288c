#!/usr/bin/env python3
"""s_memtime stamps of ONE MFMA wave (wave 0 of one workgroup) over the steps of a convolution tile, debug build of the library
(make EXTRA=-DCB_STAMPS -> tools/r05/libstamps.so).  Per step: T0 loop top, T1 after the weight-DMA issue, T2 after the step's matrix
instructions have been issued, T3 behind the step barrier.
    python tools/r05/conv_stamps.py Cin Cout k H N [p8] [block]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["CCVS_LIB"] = os.path.join(ROOT, "tools", "r05", "libstamps.so")
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from ccvs_amd import ops  # noqa: E402

cin, cout, k, h, n = [int(v) for v in sys.argv[1:6]]
p8 = "p8" in sys.argv[6:]
blocks = [int(v) for v in sys.argv[6:] if v.isdigit()] or [3000]
torch.manual_seed(0)
x = torch.randn(n, cin, h, h, device="cuda")
w = torch.randn(cout, cin, k, k, device="cuda")
b = torch.randn(cout, device="cuda")
wp = ops.pack_conv_weight(w)
if p8:
    eye = torch.eye(cin, device="cuda").view(cin, cin, 1, 1) * (cin ** 0.5)
    x = ops.conv2d(x, ops.pack_conv_weight(eye), None, cin, 1, out_p8=True)
kw = dict(out_p8=True) if p8 else {}
ops.conv2d(x, wp, b, cout, k, pad=k // 2, act=True, **kw)
torch.cuda.synchronize()
for blk in blocks:
    dbg = torch.zeros(128, dtype=torch.int64, device="cuda")
    os.environ["CCVS_CONV_DBG"] = str(dbg.data_ptr())
    os.environ["CCVS_CONV_DBG_BLOCK"] = str(blk)
    ops.conv2d(x, wp, b, cout, k, pad=k // 2, act=True, **kw)
    torch.cuda.synchronize()
    os.environ.pop("CCVS_CONV_DBG")
    t = dbg.cpu().view(32, 4)[:24]
    rows = [r for r in t.tolist() if r[0] > 0]
    print(f"{cin}->{cout} k{k} @{h} N={n} {'p8 ' if p8 else ''}workgroup {blk}: {len(rows)} steps stamped (cycles of the 100 MHz? constant clock -> shown as raw ticks)")
    prev3 = None
    for i, (t0, t1, t2, t3) in enumerate(rows):
        gap = t0 - prev3 if prev3 else 0
        print(f"  step {i:2d}: loop top -> DMA issued {t1 - t0:5d} | reads + {('72' )} MFMAs issued {t2 - t1:5d} | barrier {t3 - t2:5d} | to next top {gap:4d} | step {t3 - t0:5d}")
        prev3 = t3
    if rows:
        tot = rows[-1][3] - rows[0][0]
        print(f"  {len(rows)} steps in {tot} ticks = {tot / len(rows):.0f} per step")
