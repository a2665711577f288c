#!/usr/bin/env python3
"""Achieved HBM GB/s of the memory-bound kernels at BAIR level-5 / level-4 shapes (GPU box only)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ccvs_amd import ops  # noqa: E402


def timeit(fn, reps=3):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def row(name, ms, nbytes):
    print(f"{name:44s} {ms:8.3f} ms  {nbytes / 1e9:7.2f} GB  {nbytes / ms / 1e9:7.2f} TB/s")


def main():
    torch.manual_seed(0)
    n, c, h = 120, 96, 256
    x = torch.randn(n, c, h, h, device="cuda")
    flow = torch.randn(n, 2, h, h, device="cuda") * float(os.environ.get("FLOW_SCALE", "0.1"))
    if os.environ.get("FLOW_SMOOTH", "0") == "1":   # a smooth field of the same spread (a real motion field; per-pixel noise is the worst case for the gathers)
        flow = torch.nn.functional.interpolate(torch.randn(n, 2, h // 16, h // 16, device="cuda"), size=(h, h), mode="bicubic") * float(os.environ.get("FLOW_SCALE", "0.1"))
        flow = flow.contiguous()
    occ = torch.randn(n, 1, h, h, device="cuda")
    out = torch.empty_like(x)
    row("backwarp 120x96x256^2", timeit(lambda: ops.backwarp(x, flow, 32.0, out=out)), 2 * x.numel() * 4)
    ctx15 = [x.view(8, 15, c, h, h)[:, j] for j in range(15)]
    fo3 = torch.cat([flow, occ], dim=1).contiguous()
    row("backwarp (ctx list, fp32 out) 120x96x256^2", timeit(lambda: ops.backwarp(ctx15, flow, 32.0, out=out)), 2 * x.numel() * 4)
    row("backwarp_p8 (packed out, +flow|occ group)", timeit(lambda: ops.backwarp_p8(ctx15, fo3, 32.0)), (2 * x.numel() + 11 * n * h * h) * 4)
    dec = torch.randn(8, 128, h, h, device="cuda")
    row("warp_fuse_blend k=15 8x96x256^2", timeit(lambda: ops.warp_fuse_blend(dec[:, :96], x, flow, occ, 32.0, 15)), x.numel() * 4)
    xb = torch.randn(16, 128, 257, 257, device="cuda")
    row("blur4x4 16x128x257^2 (+lrelu,res)", timeit(lambda: ops.upfirdn2d(xb, pad=(1, 1), gain=4.0, act=True)), 2 * xb.numel() * 4)
    xs = torch.randn(16, 128, 128, 128, device="cuda")
    row("upfirdn up=2 16x128x128^2", timeit(lambda: ops.upfirdn2d(xs, up=2, pad=(2, 1), gain=4.0)), 5 * xs.numel() * 4)
    xd = torch.randn(16, 128, 256, 256, device="cuda")
    row("upfirdn down=2 16x128x256^2", timeit(lambda: ops.upfirdn2d(xd, down=2, pad=(1, 1))), 1.25 * xd.numel() * 4)
    cr = torch.randn(n, 49, 128, 128, device="cuda")
    wc = torch.randn(49, 1, 4, 4, device="cuda")
    row("dwconvT 120x49x128^2 -> 256^2", timeit(lambda: ops.dwconvT4x4s2(cr, wc)), 5 * cr.numel() * 4)
    pa = torch.randn(8, 24, h, h, device="cuda")
    pb = torch.randn(n, 24, h, h, device="cuda")
    row("correlation s2 120x24x256^2", timeit(lambda: ops.correlation7x7(pa, pb, 2, first_div=15, lrelu=True)),
        pb.numel() * 4 + n * 49 * 128 * 128 * 4)
    ctxs = [torch.randn(8, c, h, h, device="cuda") for _ in range(15)]
    wproj = torch.randn(24, c, 1, 1, device="cuda")
    w_t, cpad = ops.pack_proj_weight(wproj)
    bproj = torch.randn(24, device="cuda")
    row("warp + 1x1 proj 96->24, 120 pairs @256^2", timeit(lambda: ops.backwarp_proj(ctxs, flow, 32.0, w_t, cpad, bproj, 24, act=True)),
        n * c * h * h * 4 + n * 24 * h * h * 4)
    del ctxs
    t = torch.randn(n, 27, h, h + 8, device="cuda")
    fo = torch.randn(n, 3, h, h, device="cuda")
    import ctypes
    from ccvs_amd import lib
    L = lib.load()
    def tsa():
        lib.check(L.ccvs_tap_shift_add(ctypes.c_void_p(t.data_ptr()), None, ctypes.c_void_p(fo.data_ptr()), fo.stride(0), n, 9, h, h, 1, 0,
                                       ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)), "tsa")
    row("tap_shift_add k=9 120x27x256x264", timeit(tsa), t.numel() * 4 + 2 * fo.numel() * 4)
    src = [torch.randn(8, 1, c, h, h, device="cuda") for _ in range(15)]
    row("torch.stack 15 x [8,96,256^2]", timeit(lambda: torch.stack([s.reshape(8, c, h, h) for s in src], dim=1)), 2 * 15 * 8 * c * h * h * 4)
    sp = torch.empty(n, 195, h, h, device="cuda")
    d8 = torch.randn(8, 128, h, h, device="cuda")
    row("sp_in broadcast copy_", timeit(lambda: sp.view(8, 15, 195, h, h)[:, :, :96].copy_(d8[:, :96].unsqueeze(1))), (8 + n) * 96 * h * h * 4)


if __name__ == "__main__":
    main()
