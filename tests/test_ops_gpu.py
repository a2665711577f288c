"""-m gpu: every HIP kernel, through the C ABI, against the CPU oracle and the golden
fixtures captured from the reference (tests/golden/ops.npz)."""
import math
import os

import numpy as np
import pytest
import torch

from oracle import ccvs_oracle as O

pytestmark = pytest.mark.gpu

TOL = 2e-4  # abs, fp32 kernels vs the fp32 CPU oracle on O(1) activations


@pytest.fixture(scope="module")
def ops():
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    from ccvs_amd import ops as _ops
    return _ops


@pytest.fixture(scope="module")
def gold(golden_dir):
    return np.load(os.path.join(golden_dir, "ops.npz"))


def dev(a):
    return torch.as_tensor(np.asarray(a)).cuda()


def close(a, b, tol=TOL):
    a, b = a.detach().float().cpu(), b.detach().float().cpu()
    assert a.shape == b.shape, (a.shape, b.shape)
    d = (a - b).abs().max().item() if a.numel() else 0.0
    assert d <= tol, f"max abs diff {d:.3e} > {tol:.1e}"


# ------------------------------------------------------------------ blur / upfirdn2d
@pytest.mark.parametrize("name,kw", [("down3", dict(pad=(2, 2))), ("down1", dict(pad=(1, 1))),
                                     ("up3", dict(pad=(1, 1), gain=4.0)), ("up1", dict(pad=(2, 2), gain=4.0)),
                                     ("upsample2", dict(pad=(2, 1), up=2, gain=4.0))])
def test_upfirdn2d_golden(ops, gold, name, kw):
    y = ops.upfirdn2d(dev(gold["blur/x"]), **kw)
    close(y, torch.from_numpy(gold[f"blur/{name}"]), 1e-5)


@pytest.mark.parametrize("shape", [(1, 1, 4, 4), (2, 3, 33, 17), (1, 2, 64, 130)])
@pytest.mark.parametrize("up,down,pad", [(1, 1, (2, 2)), (1, 1, (1, 1)), (1, 2, (1, 1)), (2, 1, (2, 1)), (1, 2, (2, 2))])
def test_upfirdn2d_oracle(ops, shape, up, down, pad):
    torch.manual_seed(0)
    x = torch.randn(*shape)
    res = None
    ref = O.upfirdn2d(x, O.make_fir_kernel(gain=2.0), up=up, down=down, pad=pad)
    res = torch.randn_like(ref)
    want = (torch.nn.functional.leaky_relu(ref, 0.1) + res) * 0.5
    got = ops.upfirdn2d(x.cuda(), up=up, down=down, pad=pad, gain=2.0, act=True, residual=res.cuda(), out_scale=0.5)
    close(got, want, 1e-5)


def test_dwconvT(ops):
    torch.manual_seed(0)
    for c, h, w in [(2, 5, 7), (49, 16, 16), (1, 1, 1), (3, 9, 6), (3, 6, 4), (5, 7, 36)]:   # one / two / four input pixels per thread (W odd, % 2, % 4)
        x, wt = torch.randn(3, c, h, w), torch.randn(c, 1, 4, 4)
        close(ops.dwconvT4x4s2(x.cuda(), wt.cuda()), O.dw_convT_x2(x, wt), 1e-5)
    # into the [flow | occ] tail of a wider tensor (batch stride free), as InterBlock.forward_fused does
    x, wt = torch.randn(2, 3, 8, 12), torch.randn(3, 1, 4, 4)
    buf = torch.zeros(2, 7, 16, 24).cuda()
    ops.dwconvT4x4s2(x.cuda(), wt.cuda(), out=buf[:, 4:])
    close(buf[:, 4:], O.dw_convT_x2(x, wt), 1e-5)
    assert torch.equal(buf[:, :4].cpu(), torch.zeros(2, 4, 16, 24))


# ------------------------------------------------------------------ conv
def _conv_case(ops, gold, name, **kw):
    pre = f"conv/{name}/m."
    sd = {k[len(f"conv/{name}/"):]: torch.from_numpy(gold[k]) for k in gold.files if k.startswith(pre)}
    x = torch.from_numpy(gold[f"conv/{name}/x"])
    return sd, x, torch.from_numpy(gold[f"conv/{name}/y"])


def test_conv_plain_golden(ops, gold):
    for name, k in [("plain3", 3), ("plain1", 1), ("head9", 9), ("head5", 5)]:
        sd, x, want = _conv_case(ops, gold, name)
        w, b = sd["m.0.weight"], sd["m.0.bias"]
        act = name.startswith("plain")
        got = ops.conv2d(x.cuda(), ops.pack_conv_weight(w.cuda()), b.cuda(), w.shape[0], k, pad=k // 2, act=act)
        close(got, want)


def test_conv_down_golden(ops, gold):
    # Blur(pad 2,2) -> 3x3 stride 2 ; Blur(pad 1,1) -> 1x1 stride 2 (skip_autoencoder.py:71-79)
    sd, x, want = _conv_case(ops, gold, "down3")
    w, b = sd["m.1.weight"], sd["m.1.bias"]
    xb = ops.upfirdn2d(x.cuda(), pad=(2, 2))
    close(ops.conv2d(xb, ops.pack_conv_weight(w.cuda()), b.cuda(), w.shape[0], 3, stride=2, act=True), want)
    sd, x, want = _conv_case(ops, gold, "down1")
    w = sd["m.1.weight"]
    xb = ops.upfirdn2d(x.cuda(), pad=(1, 1))
    close(ops.conv2d(xb, ops.pack_conv_weight(w.cuda()), None, w.shape[0], 1, stride=2), want)


def test_conv_up_golden(ops, gold):
    # conv_transpose 3x3 s2 (+bias) -> Blur x4 pad (1,1) -> lrelu ; 1x1 transposed == 1x1 conv + upfirdn(up=2)
    sd, x, want = _conv_case(ops, gold, "up3")
    w, b = sd["m.0.weight"], sd["m.0.bias"]
    t = ops.conv2d(x.cuda(), ops.pack_conv_weight(w.cuda()), b.cuda(), w.shape[0], 3, stride=2, transposed=True)
    close(ops.upfirdn2d(t, pad=(1, 1), gain=4.0, act=True), want)
    sd, x, want = _conv_case(ops, gold, "up1")
    w = sd["m.0.weight"]
    t = ops.conv2d(x.cuda(), ops.pack_conv_weight(w.cuda()), None, w.shape[0], 1)
    close(ops.upfirdn2d(t, up=2, pad=(2, 1), gain=4.0), want)


@pytest.mark.parametrize("cin,cout,k,stride,hw", [(3, 8, 1, 1, (5, 7)), (49, 128, 3, 1, (40, 72)), (195, 128, 3, 1, (16, 16)),
                                                  (16, 70, 3, 2, (35, 37)), (32, 3, 9, 1, (20, 33)), (8, 8, 3, 1, (8, 8)),
                                                  (10, 33, 5, 1, (9, 64)), (64, 64, 1, 2, (31, 31))])
def test_conv_oracle_shapes(ops, cin, cout, k, stride, hw):
    torch.manual_seed(cin * 7 + cout)
    x = torch.randn(2, cin, *hw)
    w = torch.randn(cout, cin, k, k)
    b = torch.randn(cout)
    pad = k // 2 if stride == 1 else 0
    want = torch.nn.functional.leaky_relu(O.equal_conv2d(x, w, b, stride=stride, padding=pad), 0.1)
    got = ops.conv2d(x.cuda(), ops.pack_conv_weight(w.cuda()), b.cuda(), cout, k, stride=stride, pad=pad, act=True)
    close(got, want)


@pytest.mark.parametrize("cin,cout,k,kw,hw", [(128, 64, 3, 3, (48, 64)), (64, 32, 3, 3, (40, 36)), (32, 27, 1, 9, (33, 64)), (99, 64, 3, 3, (37, 100)),
                                              (128, 64, 3, 3, (256, 256)), (17, 5, 3, 3, (32, 40))])
def test_conv_tall_tile_shapes(ops, cin, cout, k, kw, hw):
    """33 .. 64 output channels on dense stride-1 rows of >= 32 lines run the 512-pixel tile (PP = 4: four pixel blocks per MFMA wave,
    two activation items per staging thread): ragged heights / widths, a channel tail, the 1 x 9 head shape, pre-activation
    addend + residual + accumulate through its epilogue; and the same bits with a CU budget (which takes the 256-pixel form)."""
    torch.manual_seed(cin + 3 * cout)
    n = 1 if hw[0] > 100 else 3
    x = torch.randn(n, cin, *hw)
    w = torch.randn(cout, cin, k, kw)
    b = torch.randn(cout)
    res = torch.randn(n, cout, *hw)
    conv = torch.nn.functional.conv2d(x, w * (1 / math.sqrt(cin * k * kw)), bias=b, padding=(k // 2, kw // 2))
    want = (torch.nn.functional.leaky_relu(conv, 0.1) + res) / math.sqrt(2)
    pk = ops.pack_conv_weight(w.cuda())
    got = ops.conv2d(x.cuda(), pk, b.cuda(), cout, k, pad=kw // 2 if k == 1 else k // 2, act=True, residual=res.cuda(), out_scale=1 / math.sqrt(2)) \
        if k == kw else None
    if got is None:   # the 1 x k head form: padding on x only
        got = ops.conv2d(x.cuda(), pk, b.cuda(), cout, k, pad=kw // 2, act=True)
        hp = got.shape[2]
        got = (got[:, :, (hp - hw[0]) // 2:(hp - hw[0]) // 2 + hw[0]] + res.cuda()) / math.sqrt(2)
    close(got, want, 2e-4)
    if k == kw:
        ops.CONV_CU_LIMIT = 7
        try:
            budget = ops.conv2d(x.cuda(), pk, b.cuda(), cout, k, pad=k // 2, act=True, residual=res.cuda(), out_scale=1 / math.sqrt(2))
        finally:
            ops.CONV_CU_LIMIT = 0
        assert torch.equal(budget, got), "the 256- and the 512-pixel tile must round identically"
        base = torch.randn(n, cout, *hw).cuda()
        acc = base.clone()
        ops.conv2d(x.cuda(), pk, b.cuda(), cout, k, pad=k // 2, out=acc, accumulate=True)
        close(acc, base.cpu() + conv, 2e-4)


@pytest.mark.parametrize("cin,cout,hw,n,div", [(99, 128, (64, 64), 6, 3), (35, 64, (40, 64), 10, 5), (19, 32, (32, 32), 4, 2), (195, 128, (16, 16), 30, 15)])
def test_conv_shared_pre_image_tile_order(ops, cin, cout, hw, n, div):
    """The first Subpixel convolution adds `pre[n // pre_div]` in its epilogue (skip_autoencoder.py:224: the `dec` block of the
    input is the same for the k contexts of a frame).  Launches of that kind walk their tiles (image group, tile, image): every
    image and tile must still be computed exactly once -- against torch, and with a CU budget (chunked 1-D launches over the same
    sequence)."""
    torch.manual_seed(cin + n)
    x = torch.randn(n, cin, *hw)
    w = torch.randn(cout, cin, 3, 3)
    b = torch.randn(cout)
    pre = torch.randn(n // div, cout, *hw)
    want = torch.nn.functional.conv2d(x, w * (1 / math.sqrt(cin * 9)), bias=b, padding=1) + pre.repeat_interleave(div, dim=0)
    want = torch.nn.functional.leaky_relu(want, 0.1)
    pk = ops.pack_conv_weight(w.cuda())
    got = ops.conv2d(x.cuda(), pk, b.cuda(), cout, 3, pad=1, act=True, pre=pre.cuda(), pre_div=div)
    close(got, want, 2e-4)
    ops.CONV_CU_LIMIT = 5
    try:
        budget = ops.conv2d(x.cuda(), pk, b.cuda(), cout, 3, pad=1, act=True, pre=pre.cuda(), pre_div=div)
    finally:
        ops.CONV_CU_LIMIT = 0
    assert torch.equal(budget, got)


@pytest.mark.parametrize("k,hw,precision", [(9, (32, 48), "bf16x3"), (9, (19, 37), "bf16x3"), (5, (16, 24), "bf16x3"), (3, (9, 8), "bf16x3"),
                                            (9, (16, 36), "f32"), (5, (13, 18), "f32")])
def test_flow_occ_heads_vs_torch(ops, k, hw, precision):
    """flow_head (2 outputs) + occ_head (1) of Matching / Subpixel (skip_autoencoder.py:176-177, 225-226) as ONE convolution to 3k
    maps plus the tap sum (`ccvs_tap_shift_add`: four pixels per lane when W % 4 == 0, one otherwise; vertical taps behind the
    split-bf16 1 x k kernel, horizontal ones behind the fp32 k x 1 kernel): against the two k x k convolutions, written into a
    channel-slice view, plain and accumulating."""
    torch.manual_seed(k * 100 + hw[1])
    n, c = 3, 32
    feat = torch.randn(n, c, *hw)
    fw, ow = torch.randn(2, c, k, k), torch.randn(1, c, k, k)
    b3 = torch.randn(3)
    scale = 1 / math.sqrt(c * k * k)
    want = torch.nn.functional.conv2d(feat, torch.cat([fw, ow]) * scale, bias=b3, padding=k // 2)
    old = ops.CONV_PRECISION
    ops.CONV_PRECISION = precision
    try:
        pk = ops.pack_head_weights(fw.cuda(), ow.cuda())
        buf = torch.zeros(n, 7, *hw).cuda()              # [.. | flow 2 | occ 1] tail of a wider tensor, as in InterBlock.forward_fused
        out = buf[:, 4:]
        ops.conv_heads(feat.cuda(), pk, b3.cuda(), out, accumulate=False)
        close(out, want, 2e-4)
        assert torch.equal(buf[:, :4].cpu(), torch.zeros(n, 4, *hw))
        base = torch.randn(n, 3, *hw)
        out.copy_(base)
        ops.conv_heads(feat.cuda(), pk, b3.cuda(), out, accumulate=True)
        close(out, base + want, 2e-4)
    finally:
        ops.CONV_PRECISION = old


def test_conv_fp32_mfma_variant(ops):
    """The exact-fp32 kernel (v_mfma_f32_32x32x2_f32) stays available next to the split-bf16 default."""
    torch.manual_seed(11)
    for cin, cout, k, stride, tr, hw in [(49, 128, 3, 1, False, (40, 72)), (32, 3, 9, 1, False, (20, 33)), (16, 70, 3, 2, False, (35, 37)),
                                         (7, 70, 3, 2, True, (17, 33))]:
        x, w, b = torch.randn(2, cin, *hw), torch.randn(cout, cin, k, k), torch.randn(cout)
        pad = 0 if (tr or stride == 2) else k // 2
        want = O.equal_conv2d(x, w, b, stride=stride, padding=pad, transpose=tr)
        got = ops.conv2d(x.cuda(), ops.pack_conv_weight(w.cuda(), "f32"), b.cuda(), cout, k, stride=stride, pad=pad, transposed=tr)
        close(got, want, 1e-4)
        got3 = ops.conv2d(x.cuda(), ops.pack_conv_weight(w.cuda(), "bf16x3"), b.cuda(), cout, k, stride=stride, pad=pad, transposed=tr)
        close(got3, want, 1e-4)
        # the split-bf16 result sits within ~1e-5 relative of the exact-fp32 kernel
        assert (got3 - got).abs().max().item() <= 2e-5 * max(1.0, want.abs().max().item())


def test_conv_transposed_oracle(ops):
    torch.manual_seed(5)
    for cin, cout, hw in [(16, 24, (8, 8)), (7, 70, (17, 33)), (32, 32, (1, 1))]:
        x, w, b = torch.randn(2, cin, *hw), torch.randn(cout, cin, 3, 3), torch.randn(cout)
        want = O.equal_conv2d(x, w, b, stride=2, padding=0, transpose=True)
        got = ops.conv2d(x.cuda(), ops.pack_conv_weight(w.cuda()), b.cuda(), cout, 3, stride=2, transposed=True)
        close(got, want)


def test_conv_views_residual_accumulate(ops):
    """channel-slice input view, strided output view, residual (a+b)/sqrt2, y += conv."""
    torch.manual_seed(9)
    big = torch.randn(2, 20, 12, 16)
    x = big[:, :12]
    w, b = torch.randn(6, 12, 3, 3), torch.randn(6)
    res = torch.randn(2, 6, 12, 16)
    want = (torch.nn.functional.leaky_relu(O.equal_conv2d(x, w, b, padding=1), 0.1) + res) / math.sqrt(2)
    outbig = torch.zeros(2, 10, 12, 16).cuda()
    ops.conv2d(big.cuda()[:, :12], ops.pack_conv_weight(w.cuda()), b.cuda(), 6, 3, pad=1, act=True, residual=res.cuda(),
               out_scale=1 / math.sqrt(2), out=outbig[:, 2:8])
    close(outbig[:, 2:8], want)
    assert outbig[:, :2].abs().max().item() == 0 and outbig[:, 8:].abs().max().item() == 0
    base = torch.randn(2, 6, 12, 16)
    acc = base.clone().cuda()
    ops.conv2d(x.cuda(), ops.pack_conv_weight(w.cuda()), b.cuda(), 6, 3, pad=1, out=acc, accumulate=True)
    close(acc, base + O.equal_conv2d(x, w, b, padding=1))


def test_res_blocks_golden(ops, gold):
    from ccvs_amd.models.skip_vid_generator.models import skip_autoencoder as sae
    for name, kw in [("down", dict(downsample=True)), ("up", dict(upsample=True))]:
        m = sae.ResBlock(16, 24, **kw).cuda()
        sd = {k[len(f"res/{name}/m."):]: torch.from_numpy(gold[k]) for k in gold.files if k.startswith(f"res/{name}/m.")}
        m.load_state_dict(sd, strict=False)
        y = m(dev(gold[f"res/{name}/x"]))
        close(y, torch.from_numpy(gold[f"res/{name}/y"]))


# ------------------------------------------------------------------ correlation / warp
@pytest.mark.parametrize("s", [1, 2])
def test_correlation_golden(ops, gold, s):
    got = ops.correlation7x7(dev(gold["corr/a"]), dev(gold["corr/b"]), s)
    close(got, torch.from_numpy(gold[f"corr/s{s}"]), 1e-5)


@pytest.mark.parametrize("s", [1, 2])
def test_correlation_properties(ops, s):
    torch.manual_seed(3)
    a, b = torch.randn(4, 24, 40, 70), torch.randn(4, 24, 40, 70)
    got = ops.correlation7x7(a.cuda(), b.cuda(), s).cpu()
    close(got, O.correlation(a, b, s), 1e-5)
    # zero-displacement channel 24 == mean_c A*B at the strided positions
    close(got[:, 24], (a * b).mean(1)[:, ::s, ::s], 1e-5)
    # first_div broadcast: one `first` serves k pairs; fused leaky relu
    got2 = ops.correlation7x7(a[:2].cuda(), b.cuda(), s, first_div=2, lrelu=True).cpu()
    a_rep = a[:2].unsqueeze(1).repeat(1, 2, 1, 1, 1).view(4, 24, 40, 70)
    close(got2, torch.nn.functional.leaky_relu(O.correlation(a_rep, b, s), 0.1), 1e-5)
    # zero padding at the border: displacement (-3,-3) at pixel (0,0) reads outside
    assert got[:, 0, 0, 0].abs().max().item() == 0


def test_backwarp_golden(ops, gold):
    got = ops.backwarp(dev(gold["warp/x"]), dev(gold["warp/flow"]), 1.0)
    close(got, torch.from_numpy(gold["warp/y"]), 1e-5)


@pytest.mark.parametrize("w", [47, 48, 4])      # W % 4 == 0: four pixels per lane, column pairs re-based at both borders
def test_backwarp_oracle(ops, w):
    torch.manual_seed(2)
    x, flow = torch.randn(3, 20, 33, w), torch.randn(3, 2, 33, w) * 4
    flow[0, :, 0, 0] = 1e4  # far outside -> zeros
    flow[1, 0, :, 0] = -0.25        # x0 = -1: only the right tap is inside
    flow[1, 0, :, w - 1] = 0.25     # x0 = W - 1: only the left tap is inside
    flow[2, 0, 5, :] = -float(w)    # a whole row sampled left of the image
    want = O.backwarp(x, flow * 2.0, O.backwarp_grid(33, w))
    close(ops.backwarp(x.cuda(), flow.cuda(), 2.0), want, 1e-4)
    ch = torch.randn(3, 24, 33, w)  # a channel-sliced destination (not 16-byte aligned rows for w = 47)
    out = ch.clone().cuda()
    ops.backwarp(x.cuda(), flow.cuda(), 2.0, out=out[:, 2:22])
    close(out[:, 2:22], want, 1e-4)
    close(out[:, :2], ch[:, :2], 0)
    close(out[:, 22:], ch[:, 22:], 0)


@pytest.mark.parametrize("k", [1, 3])
def test_warp_fuse_blend(ops, k):
    torch.manual_seed(4)
    n, c, h, w = 2, 18, 16, 24
    dec_full = torch.randn(n, c + 6, h, w)
    ctx = torch.randn(n * k, c, h, w)
    flows, occs = torch.randn(n * k, 2, h, w), torch.randn(n * k, 1, h, w)
    inp = dec_full[:, :c]
    grid = O.backwarp_grid(h, w)
    warped = O.backwarp(ctx, flows * 4.0, grid)
    if k > 1:
        confs = (1 - torch.sigmoid(occs)).view(-1, k, 1, h, w) + 1e-6
        wi = (warped.view(-1, k, c, h, w) * confs).sum(1) / confs.sum(1)
        occ = (occs.view(-1, k, 1, h, w) * confs).sum(1) / confs.sum(1)
    else:
        wi, occ = warped, occs
    m = torch.sigmoid(occ)
    want = m * inp + (1 - m) * wi
    d = dec_full.clone().cuda()
    ops.warp_fuse_blend(d[:, :c], ctx.cuda(), flows.cuda(), occs.cuda(), 4.0, k)
    close(d[:, :c], want, 1e-4)
    close(d[:, c:], dec_full[:, c:], 0)


_WARP_TILED_SCRIPT = r"""
import sys, torch
sys.path.insert(0, sys.argv[1])
from ccvs_amd import ops
torch.manual_seed(11)
out = []
for (n, k, c, h, w, scale) in ((2, 3, 10, 32, 64, 2.0), (1, 5, 7, 48, 128, 6.0), (2, 2, 4, 16, 64, 40.0), (1, 15, 8, 64, 256, 3.0)):
    dec = torch.randn(n, c + 3, h, w).cuda()
    ctxs = [torch.randn(n, c, h, w).cuda() for _ in range(k)]
    flow = (torch.randn(n * k, 2, h, w) * scale).cuda()
    flow[0, :, :4, :8] = 300.0            # far outside the image: zero weights, clamped coordinates
    occ = torch.randn(n * k, 1, h, w).cuda()
    ops.warp_fuse_blend(dec[:, :c], ctxs, flow, occ, 0.5, k)
    out.append(dec.cpu())
    out.append(ops.backwarp(ctxs, flow, 0.5).cpu())
    if c % 4 == 0:
        w_t, cout_pad = ops.pack_proj_weight(torch.randn(24, c, 1, 1).cuda())
        out.append(ops.backwarp_proj(ctxs, flow, 0.5, w_t, cout_pad, torch.randn(24).cuda(), 24, act=True).cpu())
torch.save(out, sys.argv[2])
"""


def test_warp_kernels_tiled_pixel_order(ops, tmp_path):
    """The four-pixel warp kernels (back-warp, warp + projection, fusion / blend) with a workgroup = a 64 x 16 tile (W % 64 == 0,
    H % 16 == 0: CCVS_WARP_TILED, the default) against 1024 consecutive pixels (=0): bit for bit -- small, moderate and huge
    flows, k = 2 ... 15, ragged channel chunks; the fusion / blend and the back-warp of the first case also against the oracle."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    outs = {}
    for mode in ("1", "0"):
        path = str(tmp_path / f"warp_{mode}.pt")
        env = dict(os.environ, CCVS_WARP_TILED=mode)
        res = subprocess.run([sys.executable, "-c", _WARP_TILED_SCRIPT, root, path], capture_output=True, text=True, timeout=600, env=env)
        assert res.returncode == 0, res.stderr[-3000:]
        outs[mode] = torch.load(path)
    assert len(outs["1"]) == 10
    for a, b in zip(outs["1"], outs["0"]):
        assert torch.equal(a, b), "a result depends on the pixel order of the workgroups"
    # the first case against the oracle
    torch.manual_seed(11)
    n, k, c, h, w, scale = 2, 3, 10, 32, 64, 2.0
    dec = torch.randn(n, c + 3, h, w)
    ctxs = [torch.randn(n, c, h, w) for _ in range(k)]
    flow = torch.randn(n * k, 2, h, w) * scale
    flow[0, :, :4, :8] = 300.0
    occ = torch.randn(n * k, 1, h, w)
    stacked = torch.stack(ctxs, dim=1).reshape(n * k, c, h, w)
    warped = O.backwarp(stacked, flow * 0.5, O.backwarp_grid(h, w))
    confs = (1 - torch.sigmoid(occ)).view(-1, k, 1, h, w) + 1e-6
    wi = (warped.view(-1, k, c, h, w) * confs).sum(1) / confs.sum(1)
    oc = (occ.view(-1, k, 1, h, w) * confs).sum(1) / confs.sum(1)
    m = torch.sigmoid(oc)
    want = m * dec[:, :c] + (1 - m) * wi
    close(outs["1"][0][:, :c], want, 1e-4)
    close(outs["1"][0][:, c:], dec[:, c:], 0)
    close(outs["1"][1], warped, 1e-4)


# ------------------------------------------------------------------ VQ
@pytest.mark.parametrize("name", ["default", "randn"])
def test_vq_golden(ops, gold, name):
    z = dev(gold[f"vq/{name}/z"])
    cb = dev(gold[f"vq/{name}/codebook"])
    idx = ops.vq_argmin(z, cb.t().contiguous(), (cb ** 2).sum(1)).cpu()
    want = torch.from_numpy(gold[f"vq/{name}/idx"])
    bad = (idx != want).nonzero().flatten()
    if len(bad):  # audit against the reference's own top-2 gap (default init is tie-prone)
        zf = torch.from_numpy(gold[f"vq/{name}/z"]).permute(0, 2, 3, 1).reshape(-1, cb.shape[1])
        c = cb.cpu()
        d = (zf ** 2).sum(1, keepdim=True) + (c ** 2).sum(1) - 2 * zf @ c.t()
        for r in bad.tolist():
            gap = (d[r, idx[r]] - d[r, want[r]]).abs().item()
            assert gap <= 4 * torch.finfo(torch.float32).eps * d[r].abs().max().item(), f"row {r}: gap {gap}"
        assert name == "default", "scaled-normal codebook must be bit-exact"
    emb = ops.embed_gather(want.cuda(), cb, 3, 64).view(3, -1, 8, 8)
    close(emb, torch.from_numpy(gold[f"vq/{name}/embed"]).permute(0, 3, 1, 2), 0)


def test_embed_gather_every_code(ops):
    """ccvs_embed_gather alone (VectorQuantizer.embed_code + the transposes of QVidModel.decode, quantize.py:76-83): every code
    value of the Kinetics codebook (n_e = 16384) in a ragged [N, HW] layout equals the oracle's row gather bit for bit, and
    codes outside [0, n_e) read the nearest valid row (the reference would raise; a device kernel must not fault)."""
    g = torch.Generator().manual_seed(12)
    n_e, c, hw = 16384, 24, 37
    cb = torch.randn(n_e, c, generator=g)
    n = -(-n_e // hw) + 2
    code = torch.cat([torch.randperm(n_e, generator=g), torch.randint(0, n_e, (n * hw - n_e,), generator=g)])
    z = ops.embed_gather(code.cuda(), cb.cuda(), n, hw).cpu()                       # [n, C, hw]
    want = O.embed_code(code.view(n, hw), cb).permute(0, 2, 1)                        # [n, hw, C] -> [n, C, hw]
    assert torch.equal(z, want)
    bad = code.clone()
    bad[::7] = n_e + 5
    bad[3::11] = -4
    zb = ops.embed_gather(bad.cuda(), cb.cuda(), n, hw).cpu()
    assert torch.equal(zb, O.embed_code(bad.clamp(0, n_e - 1).view(n, hw), cb).permute(0, 2, 1))


def test_vq_bair_shape(ops):
    torch.manual_seed(1)
    z = torch.randn(32, 512, 8, 8) * 0.3
    cb = torch.randn(1024, 512) * 0.3
    idx = ops.vq_argmin(z.cuda(), cb.cuda().t().contiguous(), (cb ** 2).sum(1).cuda()).cpu()
    want = O.vq_indices(z, cb)
    assert torch.equal(idx, want)
    # ties keep the lowest index: duplicate codebook rows
    cb2 = cb.clone()
    cb2[777] = cb2[5]
    idx2 = ops.vq_argmin(z.cuda(), cb2.cuda().t().contiguous(), (cb2 ** 2).sum(1).cuda()).cpu()
    assert not (idx2 == 777).any()


# ------------------------------------------------------------------ transformer pieces
def test_layernorm_gemm(ops):
    torch.manual_seed(0)
    x, g, b = torch.randn(37, 1024), torch.randn(1024), torch.randn(1024)
    close(ops.layernorm(x.cuda(), g.cuda(), b.cuda()), torch.nn.functional.layer_norm(x, (1024,), g, b), 1e-5)
    for m, n, k in [(16, 1024, 1024), (16, 3072, 1024), (16, 1024, 4096), (5, 50, 64), (100, 200, 32), (64, 4096, 1024)]:
        x, w, bias, res = torch.randn(m, k), torch.randn(n, k) * 0.05, torch.randn(n), torch.randn(m, n)
        close(ops.gemm_nt(x.cuda(), w.cuda(), bias.cuda()), x @ w.t() + bias, 1e-4)
        close(ops.gemm_nt(x.cuda(), w.cuda(), bias.cuda(), epilogue=ops.EPI_GELU), torch.nn.functional.gelu(x @ w.t() + bias), 1e-4)
        close(ops.gemm_nt(x.cuda(), w.cuda(), None, epilogue=ops.EPI_RESIDUAL, residual=res.cuda()), x @ w.t() + res, 1e-4)


def test_gemm_ln_fold(ops):
    """LayerNorm folded into the following Linear (and the QKV variant scattering K/V into the cache)."""
    torch.manual_seed(1)
    F = torch.nn.functional
    for m, n, k in [(16, 4096, 1024), (16, 1024, 1024), (37, 96, 64), (5, 50, 32)]:
        x = torch.randn(m, k) * 0.7 + 0.3          # non-zero row means
        gam, bet = torch.randn(k) * 0.3 + 1.0, torch.randn(k) * 0.2
        w, b = torch.randn(n, k) * 0.05, torch.randn(n)
        packed = [t.cuda() for t in ops.pack_ln_linear(w, b, gam, bet)]
        want = F.layer_norm(x, (k,), gam, bet) @ w.t() + b
        close(ops.gemm_ln(x.cuda(), *packed), want, 2e-4)
        close(ops.gemm_ln(x.cuda(), *packed, epilogue=ops.EPI_GELU), F.gelu(want), 2e-4)
    B, Tq, C, H, Tmax, pos0 = 3, 5, 128, 2, 16, 4
    x = torch.randn(B * Tq, C) + 0.1
    gam, bet = torch.randn(C) * 0.3 + 1.0, torch.randn(C) * 0.2
    w, b = torch.randn(3 * C, C) * 0.05, torch.randn(3 * C)
    packed = [t.cuda() for t in ops.pack_ln_linear(w, b, gam, bet)]
    want = (F.layer_norm(x, (C,), gam, bet) @ w.t() + b).view(B, Tq, 3, H, C // H)
    kc, vc = torch.zeros(B, H, Tmax, C // H).cuda(), torch.zeros(B, H, Tmax, C // H).cuda()
    pos_dev = torch.tensor([3], dtype=torch.int32).cuda()   # effective position = pos0 + 3
    q = ops.gemm_ln_qkv(x.cuda(), *packed, kc, vc, B, Tq, pos0, pos_dev)
    close(q.view(B, Tq, H, C // H), want[:, :, 0], 2e-4)
    close(kc[:, :, 7:12], want[:, :, 1].transpose(1, 2), 2e-4)
    close(vc[:, :, 7:12], want[:, :, 2].transpose(1, 2), 2e-4)
    assert kc[:, :, :7].abs().max().item() == 0 and kc[:, :, 12:].abs().max().item() == 0


def test_attention_cache(ops):
    torch.manual_seed(0)
    B, H, D, T = 3, 4, 64, 37
    qkv = torch.randn(B, T, 3 * H * D)
    q, k, v = qkv[..., :H * D], qkv[..., H * D:2 * H * D], qkv[..., 2 * H * D:]
    qh, kh, vh = [t.view(B, T, H, D).transpose(1, 2) for t in (q, k, v)]
    att = (qh @ kh.transpose(-2, -1)) / math.sqrt(D)
    att = att.masked_fill(~torch.tril(torch.ones(T, T, dtype=torch.bool)), float("-inf")).softmax(-1)
    want = (att @ vh).transpose(1, 2).reshape(B, T, H * D)
    kc, vc = torch.zeros(B, H, 64, D).cuda(), torch.zeros(B, H, 64, D).cuda()
    g = qkv.cuda()
    gq, gk, gv = g[..., :H * D], g[..., H * D:2 * H * D], g[..., 2 * H * D:]
    # prefill 30 tokens then decode 7 one by one
    ops.kv_append(gk[:, :30], gv[:, :30], kc, vc, 0)
    outs = [ops.attention(gq[:, :30], kc, vc, 0)]
    for t in range(30, T):
        step = g[:, t:t + 1].contiguous()
        sq, sk, sv = step[..., :H * D], step[..., H * D:2 * H * D], step[..., 2 * H * D:]
        ops.kv_append(sk, sv, kc, vc, t)
        outs.append(ops.attention(sq, kc, vc, t))
    close(torch.cat(outs, 1), want, 1e-5)


def test_sample_topk(ops, gold):
    logits = torch.from_numpy(gold["topk/logits"])[:, -1]
    got = ops.sample_topk(logits.cuda(), 7, 0.7).cpu()
    assert torch.equal(got, torch.from_numpy(gold["topk/icode"]).view(-1))
    # sampled: identical to torch.multinomial on the same generator stream
    torch.manual_seed(0)
    lg = torch.randn(16, 1024)
    probs = torch.softmax(O.top_k_logits(lg / 0.9, 100), -1)
    g1, g2 = torch.Generator().manual_seed(123), torch.Generator().manual_seed(123)
    want = torch.multinomial(probs, 1, generator=g1).view(-1)
    noise = torch.empty(16, 1024).exponential_(1, generator=g2)
    got = ops.sample_topk(lg.cuda(), 100, 0.9, noise=noise.cuda()).cpu()
    assert torch.equal(got, want)
    # picked tokens are always inside the top-k set
    topk = torch.topk(lg, 100, dim=1)[1]
    assert all(int(got[i]) in set(topk[i].tolist()) for i in range(16))


def test_gpt_embed_pack(ops):
    torch.manual_seed(0)
    tok, pos = torch.randn(50, 32), torch.randn(20, 32)
    idx = torch.randint(0, 50, (3, 7))
    off = torch.tensor([0, 5, 9], dtype=torch.int32)
    prow = (off.long().view(3, 1) + 2 + torch.arange(7).view(1, 7)).view(-1)
    close(ops.gpt_embed(idx.cuda(), tok.cuda(), pos.cuda(), 2, off.cuda()), tok[idx.view(-1)] + pos[prow], 0)
    wide = torch.randint(0, 50, (3, 11))  # a [B,1] column view of a wider code buffer (decode step)
    close(ops.gpt_embed(wide.cuda()[:, 4:5], tok.cuda(), pos.cuda(), 6), tok[wide[:, 4]] + pos[6], 0)
    vid = torch.randn(2, 3, 3, 9, 11) * 1.5
    assert torch.equal(ops.pack_u8(vid.cuda()).cpu(), O.pack_u8(vid))


@pytest.mark.parametrize("V,n,top_k", [(1024, 4, 100), (16384, 3, 50), (200, 5, None)])
def test_sample_topn_vs_oracle(ops, V, n, top_k):
    """ccvs_sample_topn (the n proposals of beam search + their log p, get_icode with n > 1, transformer_model.py:395-409):
    greedy = torch.topk of the top-k softmax; sampled = torch.multinomial(probs, n) without replacement under the same seed
    (the Exp(1) stream it consumes); log p within float rounding; n = 1 agrees with the single-pick kernel."""
    g = torch.Generator().manual_seed(21)
    B = 6
    logits = torch.randn(B, 1, V, generator=g) * 2
    lg = logits[:, -1].contiguous().cuda()
    for temperature in (1.0, 0.7):
        idx, logp = ops.sample_topn(lg, top_k, temperature, n)
        want_idx, probs = O.get_icode(logits, temperature, top_k, False, n=n)
        assert torch.equal(idx.cpu(), want_idx)
        close(logp, torch.log(torch.gather(probs, 1, want_idx)), 1e-5)
        torch.manual_seed(33)
        want_idx, probs = O.get_icode(logits, temperature, top_k, True, n=n)
        torch.manual_seed(33)
        noise = torch.empty(B, V).exponential_(1)
        idx, logp = ops.sample_topn(lg, top_k, temperature, n, noise=noise.cuda())
        assert torch.equal(idx.cpu(), want_idx), (V, n, top_k, temperature)
        close(logp, torch.log(torch.gather(probs, 1, want_idx)), 1e-5)
        one, _ = ops.sample_topn(lg, top_k, temperature, 1, noise=noise.cuda())
        assert torch.equal(one[:, 0], ops.sample_topk(lg, top_k, temperature, noise=noise.cuda()))


def test_gpt_decode_step_matches_per_op_engine(ops):
    """ccvs_gpt_decode_step (hipGraph replay and eager) == prefill + step() + argmax through the per-op entry points, bit for bit."""
    from ccvs_amd.models.skip_vid_generator.models import mingpt
    torch.manual_seed(3)
    for n_embd, n_head in ((256, 4), (128, 8), (64, 4)):
        net = mingpt.GPT(vocab_size=200, block_size=700, num_blocks=44, n_layer=3, n_head=n_head, n_embd=n_embd,
                         emb_mode="temporal", shape=(4, 4)).cuda()
        for p in net.parameters():  # non-trivial LayerNorm / bias values
            p.data.add_(0.05 * torch.randn_like(p))
        code = torch.randint(0, 200, (16, 16), device="cuda")
        n_new = 600 if n_embd == 256 else 100   # 600: several double-buffered K/V batches in the attention kernel
        net.drop_engine_state()
        got_graph = net.generate(code, n_new, sample=False, top_k=10).cpu()
        got_eager = net.generate(code, n_new, sample=False, top_k=10, use_graph=False).cpu()
        net.begin(16, 16 + n_new)
        logits = net.prefill(code)
        seq = [code]
        for i in range(n_new):
            tok = logits.argmax(dim=-1, keepdim=True)
            seq.append(tok)
            if i < n_new - 1:
                logits = net.step(tok)
        want = torch.cat(seq, dim=1).cpu()
        assert torch.equal(got_graph, want) and torch.equal(got_eager, want)


def test_gpt_decode_device_rng_sampling(ops):
    """In-kernel Philox noise: argmax(p / Exp(1)) draws from p -- uniform logits give uniform tokens; the key comes from
    torch's generator (same seed -> same tokens, new seed -> new tokens); top-k restricts the support."""
    from ccvs_amd.models.skip_vid_generator.models import mingpt
    torch.manual_seed(5)
    net = mingpt.GPT(vocab_size=16, block_size=512, num_blocks=32, n_layer=1, n_head=2, n_embd=32, emb_mode="temporal", shape=(4, 4)).cuda()
    with torch.no_grad():
        net.head.weight.zero_()  # logits == 0: uniform over the vocabulary
    code = torch.zeros(16, 1, dtype=torch.int64, device="cuda")
    torch.manual_seed(11)
    a = net.generate(code, 400, sample=True, top_k=None).cpu()[:, 1:]
    torch.manual_seed(11)
    b = net.generate(code, 400, sample=True, top_k=None).cpu()[:, 1:]
    c = net.generate(code, 400, sample=True, top_k=None).cpu()[:, 1:]
    assert torch.equal(a[:, 1:], b[:, 1:]) and not torch.equal(a[:, 1:], c[:, 1:])  # column 0 is the prefill pick (torch noise)
    counts = torch.bincount(a.reshape(-1), minlength=16).float()
    assert counts.sum() == 16 * 400
    assert (counts - 400).abs().max() < 5 * (400 * 15 / 16) ** 0.5, counts  # 5 sigma of a binomial bin
    assert (a[0] != a[1]).float().mean() > 0.8 and (a[:, 1:] != a[:, :-1]).float().mean() > 0.8  # rows / steps independent
    # skewed distribution + top-k: only the k most likely tokens appear, in the right proportions
    lg = torch.log(torch.tensor([8.0, 4.0, 2.0, 1.0] + [0.5] * 12, device="cuda"))
    with torch.no_grad():  # in-place on the parameters themselves: bumps their version, so the folded weights are re-packed
        net.head.weight.zero_()
        net.head.weight[:, 0] = lg
        net.ln_f.weight.zero_()  # LN output == bias = e0: logits = lg exactly
        net.ln_f.bias.zero_()
        net.ln_f.bias[0] = 1.0
    d = net.generate(code, 400, sample=True, top_k=3).cpu()[:, 2:]
    cnt = torch.bincount(d.reshape(-1), minlength=16).float()
    assert cnt[3:].sum() == 0
    frac = cnt[:3] / cnt.sum()
    assert (frac - torch.tensor([8.0, 4.0, 2.0]) / 14.0).abs().max() < 0.03, frac


@pytest.mark.parametrize("D", [16, 32, 64])
def test_attention_decode_long_cache(ops, D):
    """Decode-form attention against caches of 1 .. several register batches (odd / even batch counts, ragged tails)."""
    torch.manual_seed(D)
    B, H, Tmax = 2, 3, 2200
    kc, vc = torch.randn(B, H, Tmax, D).cuda(), torch.randn(B, H, Tmax, D).cuda()
    batch = 8 * (64 // (D // 4)) * 8
    lens = sorted({1, 2, batch - 1, batch, batch + 1, 2 * batch, 2 * batch + 7, min(3 * batch + 5, Tmax)})
    for L in lens:
        q = torch.randn(B, 1, H * D).cuda()
        got = ops.attention(q, kc, vc, L - 1)   # query at position L-1 sees keys 0..L-1
        qh = q.view(B, H, 1, D)
        att = ((qh @ kc[:, :, :L].transpose(-2, -1)) / math.sqrt(D)).softmax(-1)
        want = (att @ vc[:, :, :L]).view(B, 1, H * D)
        close(got, want, 2e-5)
        pos_dev = torch.tensor([L - 1], dtype=torch.int32).cuda()  # device-resident position, as in the captured decode step
        close(ops.attention(q, kc, vc, 0, pos_dev), want, 2e-5)


def test_warp_kernels_with_context_list(ops):
    """ccvs_backwarp_ctx / ccvs_warp_fuse_blend_ctx read the k contexts in place (ring slots, any order, plus a tensor
    outside the ring) and equal the stacked-copy forms bit for bit."""
    torch.manual_seed(2)
    n, k, c, h, w = 3, 4, 20, 12, 16
    ring = torch.randn(n, 6, c, h, w).cuda()
    extra = torch.randn(n, 1, c, h, w).cuda()
    slots = [4, 1, 5]
    ctxs = [ring[:, s_] for s_ in slots] + [extra[:, 0]]
    stacked = torch.stack(ctxs, dim=1).reshape(n * k, c, h, w).contiguous()
    flow = (torch.randn(n * k, 2, h, w) * 3).cuda()
    occ = torch.randn(n * k, 1, h, w).cuda()
    assert torch.equal(ops.backwarp(ctxs, flow, 0.5), ops.backwarp(stacked, flow, 0.5))
    dec = torch.randn(n, c + 5, h, w).cuda()
    a, b = dec.clone(), dec.clone()
    ops.warp_fuse_blend(a[:, :c], ctxs, flow, occ, 0.5, k)
    ops.warp_fuse_blend(b[:, :c], stacked, flow, occ, 0.5, k)
    assert torch.equal(a, b)


def test_conv_packed_activation_chain(ops):
    """conv -> conv -> conv with split-bf16 packed (P8) intermediates == the same chain with fp32 intermediates, bit for bit
    (3x3, 1x1 and 1xk kernels, ragged image edges, Cout 128 / 64 / 32 tiles, channel counts that are odd multiples of 8)."""
    if ops.CONV_PRECISION != "bf16x3":
        pytest.skip("packed activations are a split-bf16 format")
    torch.manual_seed(5)
    for n, h, w in ((3, 40, 36), (2, 8, 8), (1, 70, 33)):
        x = torch.randn(n, 19, h, w).cuda()
        specs = [(19, 128, 3, 3), (128, 72, 3, 3), (72, 40, 1, 1), (40, 24, 1, 5), (24, 32, 3, 3)]
        ws = [torch.randn(co, ci, kh, kw).cuda() for ci, co, kh, kw in specs]
        bs = [torch.randn(co).cuda() for _, co, _, _ in specs]
        packs = [ops.pack_conv_weight(wt) for wt in ws]

        def run2(packed):
            t = x
            outs = []
            for i, (ci, co, kh, kw) in enumerate(specs):
                last = i == len(specs) - 1
                pad = kh // 2 if kh == kw else 2
                t = ops.conv2d(t, packs[i], bs[i], co, kh, pad=pad, act=not last, out_p8=packed and not last)
                outs.append(t)
            return outs
        a, b = run2(False), run2(True)
        for i in range(len(specs) - 1):
            # the packed tensor decodes to hi + lo of the fp32 value: equal to 2^-16 relative
            assert b[i].shape == tuple(a[i].shape)
            assert (b[i].float() - a[i]).abs().max().item() <= 2e-5 * a[i].abs().max().item() + 1e-6
        assert torch.equal(a[-1], b[-1])


def test_conv_packed_chain_of_the_interblocks(ops):
    """The conv -> conv chain of Matching / Subpixel with packed (P8) intermediates at the tile forms the decoder uses: 49 -> 128
    (128 channels per workgroup, packed output; and its two-workgroups-per-CU form), 128 -> 64 (the 512-pixel tile with packed input
    AND output, tap loop written out), 64 -> 32 (32-channel kernel, packed in / out), 1 x 9 head (packed input, fp32 output) -- equal to
    the fp32-intermediate chain bit for bit, with the shared pre-activation image in the first layer, ragged heights included."""
    if ops.CONV_PRECISION != "bf16x3":
        pytest.skip("packed activations are a split-bf16 format")
    torch.manual_seed(6)
    for n, h, w, div in ((4, 40, 64, 2), (2, 33, 96, 1), (3, 16, 32, 3)):
        x = torch.randn(n, 49, h, w).cuda()
        pre = torch.randn(n // div, 128, h, w).cuda()
        specs = [(49, 128, 3, 3), (128, 64, 3, 3), (64, 32, 3, 3), (32, 27, 1, 9)]
        ws = [torch.randn(co, ci, kh, kw).cuda() for ci, co, kh, kw in specs]
        bs = [torch.randn(co).cuda() for _, co, _, _ in specs]
        packs = [ops.pack_conv_weight(wt) for wt in ws]

        def run2(packed):
            t = ops.conv2d(x, packs[0], bs[0], 128, 3, pad=1, act=True, pre=pre, pre_div=div, out_p8=packed)
            t = ops.conv2d(t, packs[1], bs[1], 64, 3, pad=1, act=True, out_p8=packed)
            t = ops.conv2d(t, packs[2], bs[2], 32, 3, pad=1, act=True, out_p8=packed)
            return ops.conv2d(t, packs[3], bs[3], 27, 1, pad=4)
        assert torch.equal(run2(False), run2(True)), (n, h, w)


@pytest.mark.parametrize("n,k,c,h,w", [(2, 3, 24, 40, 64), (1, 2, 96, 33, 36), (2, 1, 16, 8, 8), (1, 2, 8, 5, 4)])
def test_backwarp_p8_feeds_the_first_subpixel_convolution(ops, n, k, c, h, w):
    """`ccvs_backwarp_p8_ctx`: the k contexts warped straight into the packed input [warped | flow | occ | 0 x 5] of Subpixel's first
    convolution.  The packed tensor decodes to the fp32 back-warp (hi + lo: 2^-16 relative), and the convolution on it equals the
    convolution on the fp32 tensor of the same s + 8 channels bit for bit -- far flows (zero padding) and ragged sizes included."""
    if ops.CONV_PRECISION != "bf16x3":
        pytest.skip("packed activations are a split-bf16 format")
    g = torch.Generator().manual_seed(n * 100 + c)
    ctxs = [torch.randn(n, c, h, w, generator=g).cuda() for _ in range(k)]
    fo = torch.randn(n * k, 3, h, w, generator=g).cuda()
    fo[:, :2] *= 3.0
    fo[0, :2, :2, :3] = 500.0          # far outside: zero padding
    mult = 2.0
    packed = ops.backwarp_p8(ctxs, fo, mult)
    assert packed.shape == (n * k, c + 8, h, w)
    want = torch.cat([ops.backwarp(ctxs, fo[:, :2].contiguous(), mult), fo, fo.new_zeros(n * k, 5, h, w)], dim=1)
    dec = packed.float()
    assert (dec - want).abs().max().item() <= 2e-5 * want.abs().max().item() + 1e-7
    assert torch.equal(dec[:, c + 3:], want[:, c + 3:])
    wt = torch.randn(128, c + 8, 3, 3, generator=g).cuda()
    wt[:, c + 3:] = 0
    b = torch.randn(128, generator=g).cuda()
    pre = torch.randn(n, 128, h, w, generator=g).cuda()
    pk = ops.pack_conv_weight(wt)
    a = ops.conv2d(want, pk, b, 128, 3, pad=1, act=True, pre=pre, pre_div=k)
    bb = ops.conv2d(packed, pk, b, 128, 3, pad=1, act=True, pre=pre, pre_div=k)
    assert torch.equal(a, bb)     # same samples as backwarp4_kernel, same split as the staging waves: the same operands
    # ... and against torch on the unpadded 99-channel form of the layer
    ref = torch.nn.functional.conv2d(want[:, :c + 3].cpu(), (wt[:, :c + 3] * (1 / math.sqrt((c + 8) * 9))).cpu(), bias=b.cpu(), padding=1)
    ref = torch.nn.functional.leaky_relu(ref + pre.cpu().repeat_interleave(k, dim=0), 0.1)
    assert (bb.cpu() - ref).abs().max().item() <= 2e-5 * ref.abs().max().item() + 2e-4


def test_gemm_row_blocked_matches_plain(ops):
    """Up to 256 rows the weight-stream kernel runs (one workgroup per 16 rows x 16 columns): every 16-row slice equals the
    plain M = 16 launch bit for bit, with and without the folded LayerNorm, GELU / residual epilogues and ragged M, N.  Beyond
    256 rows the prefill form runs (row-blocked, 8 K slices instead of 4: another summation order): equal to rounding."""
    torch.manual_seed(9)
    for m, n, k in ((128, 320, 256), (150, 200, 128), (260, 1024, 512), (64, 48, 64)):
        x = torch.randn(m, k).cuda()
        w = (torch.randn(n, k) * 0.05).cuda()
        b = torch.randn(n).cuda()
        res = torch.randn(m, n).cuda()
        gam, bet = (1 + 0.1 * torch.randn(k)).cuda(), (0.1 * torch.randn(k)).cuda()
        packed = ops.pack_ln_linear(w, b, gam, bet)
        full = [ops.gemm_nt(x, w, b), ops.gemm_nt(x, w, b, ops.EPI_RESIDUAL, residual=res), ops.gemm_ln(x, *packed, epilogue=ops.EPI_GELU)]
        for r0 in range(0, m, 16):
            xs = x[r0:r0 + 16].contiguous()
            part = [ops.gemm_nt(xs, w, b), ops.gemm_nt(xs, w, b, ops.EPI_RESIDUAL, residual=res[r0:r0 + 16].contiguous()),
                    ops.gemm_ln(xs, *packed, epilogue=ops.EPI_GELU)]
            for f, q in zip(full, part):
                if m <= 256:
                    assert torch.equal(f[r0:r0 + 16], q)
                else:
                    close(f[r0:r0 + 16], q, 2e-5)
        want = torch.nn.functional.gelu(torch.nn.functional.layer_norm(x, (k,), gam, bet) @ w.t() + b)
        close(full[2], want, 5e-4)


@pytest.mark.parametrize("D,B,H,Tq,pos0", [(64, 16, 16, 1024, 0), (64, 2, 3, 301, 0), (64, 2, 2, 97, 200), (32, 3, 2, 130, 5), (16, 2, 2, 33, 0),
                                           (16, 1, 2, 2, 62)])
def test_attention_prefill_mfma(ops, D, B, H, Tq, pos0):
    """Flash-style prefill attention on the fp32 matrix cores (attention_prefill_kernel) against softmax(QK^T / sqrt(d)
    masked) V of mingpt.py:67-77: the BAIR-size case T = 1024, D = 64, B = 16; ragged query counts (not multiples of the
    32-query wave block / 128-query workgroup); queries that start at a cache offset (extend); the unwritten tail of the
    cache filled with NaN must not leak into the result."""
    torch.manual_seed(D + Tq)
    Tmax = pos0 + Tq + 37
    L = pos0 + Tq
    kc = torch.full((B, H, Tmax, D), float("nan"))
    vc = torch.full((B, H, Tmax, D), float("nan"))
    kc[:, :, :L], vc[:, :, :L] = torch.randn(B, H, L, D), torch.randn(B, H, L, D)
    qfull = torch.randn(B, Tq, 3 * H * D)          # q is a strided view, as in the fused QKV output
    q = qfull[..., H * D:2 * H * D]
    got = ops.attention(q.cuda(), kc.cuda(), vc.cuda(), pos0)
    qh = q.reshape(B, Tq, H, D).transpose(1, 2).double()
    att = (qh @ kc[:, :, :L].double().transpose(-2, -1)) / math.sqrt(D)
    vis = torch.arange(L).view(1, L) <= (pos0 + torch.arange(Tq)).view(Tq, 1)
    att = att.masked_fill(~vis, float("-inf")).softmax(-1)
    want = (att @ vc[:, :, :L].double()).transpose(1, 2).reshape(B, Tq, H * D).float()
    assert torch.isfinite(got).all()
    close(got, want, 2e-5)
    pos_dev = torch.tensor([pos0], dtype=torch.int32).cuda()      # device-resident offset
    close(ops.attention(q.cuda(), kc.cuda(), vc.cuda(), 0, pos_dev), want, 2e-5)


@pytest.mark.parametrize("cin,cout,h,w,k", [(96, 24, 40, 72, 3), (192, 48, 16, 16, 2), (384, 96, 8, 8, 1), (40, 16, 9, 13, 2)])
def test_backwarp_proj_fused(ops, cin, cout, h, w, k):
    """ccvs_backwarp_proj_ctx == ConvLayer(C, C', 1)(backwarp(ctx, flow * mult)) of Matching (skip_autoencoder.py:186-190), with
    the contexts read in place from a list, out-of-range flows, and Cout below its padded instantiation."""
    g = torch.Generator().manual_seed(cin + k)
    nf = 3
    ctxs = [torch.randn(nf, cin, h, w, generator=g) for _ in range(k)]
    flow = torch.randn(nf * k, 2, h, w, generator=g) * 2.5
    flow[0, :, 0, 0] = 1e4                                   # far outside: zero padding
    wt = torch.randn(cout, cin, 1, 1, generator=g)
    bias = torch.randn(cout, generator=g)
    mult = 4.0
    stacked = torch.stack(ctxs, dim=1).reshape(nf * k, cin, h, w)
    warped = O.backwarp(stacked, flow * mult, O.backwarp_grid(h, w))
    want = torch.nn.functional.leaky_relu(O.equal_conv2d(warped, wt, bias), 0.1)
    w_t, cpad = ops.pack_proj_weight(wt.cuda())
    got = ops.backwarp_proj([c.cuda() for c in ctxs], flow.cuda(), mult, w_t, cpad, bias.cuda(), cout)
    close(got, want, 1e-4)   # sample positions differ from grid_sample's by an ulp of the (scaled) coordinates


def test_conv_persistent_tiles_are_bit_identical(tmp_path):
    """`CCVS_CONV_PT` (conv2d_bf16_pt.h: resident workgroups walk their tiles, a tile's prologue under the step loop of the tile before)
    against the producer / consumer kernel it replaces on the same inputs, by flipping the switch in two worker processes (the library
    reads it once): fp32 input with and without the packed K tail, packed input (256- and 512-pixel tiles), fp32 and packed output,
    every epilogue addend (pre-activation image with the shared-image tile order, residual + scale, accumulate), two channel blocks
    per position, workgroups with odd and even tile counts -- and shapes the persistent form must refuse.  Same arithmetic per
    output in the same order: the same bits.  (Reference of the layer: skip_autoencoder.py:53-59.)"""
    import subprocess
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    outs = []
    for flag in ("0", "3"):
        path = str(tmp_path / f"pt{flag}.npz")
        env = dict(os.environ, CCVS_CONV_PT=flag)
        subprocess.run([sys.executable, os.path.join(here, "conv_pt_worker.py"), path], env=env, check=True, timeout=900)
        outs.append(np.load(path))
    assert sorted(outs[0].files) == sorted(outs[1].files) and len(outs[0].files) == 5 * 8 + 3
    for key in outs[0].files:
        a, b = outs[0][key], outs[1][key]
        assert np.array_equal(a.view(np.uint32), b.view(np.uint32)), key
    assert np.isfinite(outs[0]["f32_99"]).all() and np.abs(outs[0]["f32_99"]).max() > 0.1


def test_l2_normalize_channels_and_the_encoder_with_normalize_out():
    """`opt.normalize_out` (skip_autoencoder.py:348-349: out / torch.norm(out, p=2, dim=1, keepdim=True)): the kernel alone on odd sizes
    against the torch expression, then the encoder with `--q_normalize_out` against the oracle's (oracle/ccvs_oracle.py:
    encoder_forward) -- latents of unit length per position, within 1e-5 (fp32; the sum over the channels runs in channel order
    here, in torch's reduction order there)."""
    from ccvs_amd import ops
    from ccvs_amd.tools.options import Options
    from ccvs_amd.models.skip_vid_generator.models import skip_autoencoder as sae
    from tests.test_capi_host import TINY_ARGV
    g = torch.Generator().manual_seed(3)
    for shape in ((5, 16, 7, 9), (1, 1, 1, 1), (3, 257, 8, 8), (2, 4, 33, 65)):
        x = torch.randn(*shape, generator=g) * 3
        want = x / torch.norm(x, p=2, dim=1, keepdim=True)
        got = ops.l2_normalize_channels_(x.cuda().contiguous()).cpu()
        assert (got - want).abs().max().item() < 1e-6, shape
    z = torch.zeros(1, 4, 2, 2)
    assert torch.isnan(ops.l2_normalize_channels_(z.cuda()).cpu()).all()      # 0 / 0, as in the reference
    opt = Options().parse(True, True, argv=TINY_ARGV + ["--q_normalize_out"])["qvid_generator"]
    assert opt.normalize_out
    torch.manual_seed(0)
    enc = sae.SkipGANEncoder(opt).cuda().eval()
    vid = torch.rand(2, 3, 3, 32, 32, generator=g) * 2 - 1
    with torch.no_grad():
        got, inter = enc(vid.cuda())
        sd = {k: v.detach().cpu() for k, v in enc.state_dict().items()}
        want, winter = O.encoder_forward(sd, opt, vid)
    assert got.shape == want.shape and (got.cpu() - want).abs().max().item() < 1e-5
    assert (got.cpu().reshape(-1, *got.shape[2:]).pow(2).sum(1).sqrt() - 1).abs().max().item() < 1e-5
    for a, b in zip(inter, winter):
        assert (a.cpu() - b).abs().max().item() < 1e-4
