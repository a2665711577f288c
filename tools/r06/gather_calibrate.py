#!/usr/bin/env python3
"""Calibrate gfx950's FETCH_SIZE for the ACCESS PATTERN of the gather kernels (backwarp4: per lane four pixels, the two x-taps of a row
as one 8-byte load from a gathered address): MI355X_MICROARCH.md calibrates 16-byte-per-lane streams (tallied at half their
bytes) and says other widths are uncalibrated.  With a flow of small random sub-pixel shifts every input byte of the source planes is
needed exactly once per launch through HBM (the planes are far larger than L2 + Infinity Cache when N x C x H x W x 4 >> 256 MB),
so  factor = (bytes of x + bytes of flow) / (FETCH_SIZE x 1024)  is the correction for this pattern.
    rocprofv3 --kernel-trace --pmc FETCH_SIZE ... -- python3 tools/r06/gather_calibrate.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from ccvs_amd import ops  # noqa: E402

n, c, h, w = 48, 96, 256, 256           # 1.2 GB of source planes: every launch streams them from HBM
g = torch.Generator().manual_seed(0)
x = torch.randn(n, c, h, w, generator=g).cuda()
flow = (torch.rand(n, 2, h, w, generator=g) * 1.5 - 0.75).cuda()      # sub-pixel shifts: the taps stay inside the neighbouring pixels
out = torch.empty_like(x)
for _ in range(3):
    ops.backwarp(x, flow, 1.0, out=out)
torch.cuda.synchronize()
print(f"ALG_READ_BYTES {x.numel() * 4 + flow.numel() * 4}  ALG_WRITE_BYTES {out.numel() * 4}")
