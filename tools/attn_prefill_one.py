#!/usr/bin/env python3
"""The MFMA prefill attention on one BAIR-size problem (B x H x T x D = 16 x 16 x 1024 x 64, causal): time, TFLOP/s, and the
target of the --pmc run of tools/pmc_attention.sh.   python tools/attn_prefill_one.py [B] [T] [reps]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ccvs_amd import ops  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
T = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 20
H, D = 16, 64
torch.manual_seed(0)
q = torch.randn(B, T, H * D, device="cuda")
kc, vc = torch.randn(B, H, T, D, device="cuda"), torch.randn(B, H, T, D, device="cuda")
ops.attention(q, kc, vc, 0)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(reps):
    ops.attention(q, kc, vc, 0)
e1.record()
e1.synchronize()
ms = e0.elapsed_time(e1) / reps
flops = 4.0 * B * H * D * T * (T + 1) / 2          # QK^T and PV over the causal half
print(f"attention_prefill_kernel<{D}> B={B} H={H} T={T}: {ms:.3f} ms, {flops / ms / 1e9:.1f} TFLOP/s algorithmic (fp32 MFMA peak 157.3)")
