#!/usr/bin/env python3
"""nn.Linear timing against the number of rows M (decode M = batch ... prefill M = batch x tokens), eager launches."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ccvs_amd import ops  # noqa: E402

torch.manual_seed(0)
for M in (16, 64, 256, 1024, 4096):
    for n, k, ln in ((3072, 1024, True), (1024, 4096, False)):
        w = torch.randn(n, k, device="cuda") * 0.02
        b = torch.randn(n, device="cuda")
        x = torch.randn(M, k, device="cuda")
        pk = ops.pack_ln_linear(w, b, torch.ones(k, device="cuda"), torch.zeros(k, device="cuda")) if ln else None
        out = torch.empty(M, n, device="cuda")
        f = (lambda: ops.gemm_ln(x, *pk, out=out)) if ln else (lambda: ops.gemm_nt(x, w, b, out=out))
        f()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            f()
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / 20 * 1e3
        print(f"M={M:5d} N={n} K={k}: {us:8.1f} us  {2 * M * n * k / us / 1e6:7.2f} TFLOP/s")
