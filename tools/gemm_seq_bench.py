#!/usr/bin/env python3
"""Whole-sequence GEMMs of the GPT (prefill / teacher-forced forward / re-prefill: `CCVS_GEMM_SEQ`): TFLOP/s of the four
layer shapes at the row counts the configurations use -- BAIR token-group prefill (3 x 16 x 64 = 3072 rows), Kinetics prefill
(64 x 320 = 20480), Drums re-prefill (4 x 1218 = 4872) -- and accuracy against torch fp32.
    python tools/gemm_seq_bench.py            (CCVS_GEMM_SEQ_DENSE=0: the row-blocked weight-stream kernel of rounds 2-3)"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ccvs_amd import ops  # noqa: E402


def timed(fn, reps=5):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def main():
    torch.manual_seed(0)
    C = 1024
    SEQ = ops.GEMM_SEQ
    print("dense" if os.environ.get("CCVS_GEMM_SEQ_DENSE", "1") != "0" else "row-blocked", "sequence form")
    for M in ([int(v) for v in sys.argv[1:]] or [1024, 3072, 4872, 20480]):
        total_ms, total_fl = 0.0, 0.0
        for name, n, k, ln, epi in [("qkv (ln)", 3 * C, C, True, ops.EPI_NONE), ("proj +res", C, C, False, ops.EPI_RESIDUAL),
                                    ("fc (ln, gelu)", 4 * C, C, True, ops.EPI_GELU), ("fc2 +res", C, 4 * C, False, ops.EPI_RESIDUAL)]:
            w = torch.randn(n, k, device="cuda") * 0.02
            b = torch.randn(n, device="cuda")
            x = torch.randn(M, k, device="cuda")
            res = torch.randn(M, n, device="cuda")
            out = torch.empty(M, n, device="cuda")
            if ln:
                gamma, beta = 1 + 0.1 * torch.randn(k, device="cuda"), 0.1 * torch.randn(k, device="cuda")
                pk = ops.pack_ln_linear(w, b, gamma, beta)
                run = lambda: ops.gemm_ln(x, *pk, epilogue=epi | SEQ, out=out)
                want = torch.nn.functional.layer_norm(x, (k,), gamma, beta) @ w.t() + b
                if epi == ops.EPI_GELU:
                    want = torch.nn.functional.gelu(want)
            else:
                run = lambda: ops.gemm_nt(x, w, b, epi | SEQ, residual=res, out=out)
                want = x @ w.t() + b + res
            ms = timed(run)
            err = (out - want).abs().max().item()
            fl = 2.0 * M * n * k
            total_ms += ms
            total_fl += fl
            print(f"M={M:6d} {name:14s} N={n:5d} K={k:5d}  {ms:7.3f} ms  {fl / ms / 1e9:6.1f} TFLOP/s  max|err| vs torch {err:.1e}")
        print(f"M={M:6d} one layer's four GEMMs: {total_ms:7.3f} ms, {total_fl / total_ms / 1e9:6.1f} TFLOP/s (fp32 matrix peak 157.3)")


if __name__ == "__main__":
    main()
