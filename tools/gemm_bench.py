#!/usr/bin/env python3
"""Decode-regime GEMM timing (M = 16 rows) for the four GPT shapes + attention, as a hipGraph of
24 back-to-back launches (what one decode step issues).  GPU box only."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ccvs_amd import ops  # noqa: E402


def timed_graph(fn, reps=5):
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        fn()
    torch.cuda.current_stream().wait_stream(s)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        fn()
    g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        g.replay()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def main():
    torch.manual_seed(0)
    M, C, L = 16, 1024, 24
    x = torch.randn(M, C, device="cuda")
    for name, n, k, ln in [("qkv 3072x1024 (ln)", 3072, 1024, True), ("proj 1024x1024", 1024, 1024, False),
                           ("fc 4096x1024 (ln,gelu)", 4096, 1024, True), ("fc2 1024x4096", 1024, 4096, False)]:
        ws = [torch.randn(n, k, device="cuda") * 0.02 for _ in range(L)]   # distinct weights: no cache reuse
        bs = torch.randn(n, device="cuda")
        xin = torch.randn(M, k, device="cuda")
        packs = [ops.pack_ln_linear(w, bs, torch.ones(k, device="cuda"), torch.zeros(k, device="cuda")) for w in ws] if ln else None
        out = torch.empty(M, n, device="cuda")

        def run():
            for i in range(L):
                if ln:
                    ops.gemm_ln(xin, *packs[i], out=out)
                else:
                    ops.gemm_nt(xin, ws[i], bs, out=out)
        ms = timed_graph(run)
        mb = n * k * 4 / 1e6
        print(f"{name:26s} {mb:6.1f} MB  {1e3 * ms / L:7.2f} us/launch  {mb / (ms / L) / 1e3:6.2f} TB/s")
    for T in (128, 512, 1023):
        H, D = 16, 64
        kc = [torch.randn(M, H, 1024, D, device="cuda") for _ in range(L)]
        vc = [torch.randn(M, H, 1024, D, device="cuda") for _ in range(L)]
        q = torch.randn(M, 1, C, device="cuda")

        def run():
            for i in range(L):
                ops.attention(q, kc[i], vc[i], T)
        ms = timed_graph(run)
        mb = 2 * M * H * (T + 1) * D * 4 / 1e6
        print(f"attention decode T={T:4d}     {mb:6.1f} MB  {1e3 * ms / L:7.2f} us/launch  {mb / (ms / L) / 1e3:6.2f} TB/s")


if __name__ == "__main__":
    main()
