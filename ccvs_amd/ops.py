"""Tensor-level wrappers over the C ABI (include/ccvs_hip.h).

PyTorch is plumbing here: it owns device memory and the stream; every call below hands raw
device pointers and sizes to libccvs_hip.so.  There is no fallback: a CPU tensor or a missing
library raises.
"""
import ctypes as C
import math

import torch

from . import lib as _lib

ACT_NONE, ACT_LRELU = 0, 1
EPI_NONE, EPI_GELU, EPI_RESIDUAL = 0, 1, 2
GEMM_SEQ = 0x100   # CCVS_GEMM_SEQ: whole-sequence call (OR-ed into the epilogue word)


class KernelTimer:
    """HIP-event bracket around individual launches of one kernel on the current stream
    (used by bench.py for the roofline of the dominant kernel; off by default)."""

    def __init__(self):
        self.records = []  # (name, algorithmic flops, start event, end event)
        self._open = None

    def begin(self, name, flops=0.0, nbytes=0.0, side=0.0, tag=None):
        e0 = torch.cuda.Event(enable_timing=True)
        e0.record()
        self._open = (name, flops, e0, nbytes, side, tag)

    def end(self):
        e1 = torch.cuda.Event(enable_timing=True)
        e1.record()
        name, flops, e0, nbytes, side, tag = self._open
        self.records.append((name, flops, e0, e1, nbytes, side, tag))

    def summary(self, name):
        """(launches, total algorithmic flops, total milliseconds) -- call after a device sync."""
        rec = [r for r in self.records if r[0] == name]
        return len(rec), sum(r[1] for r in rec), sum(r[2].elapsed_time(r[3]) for r in rec)

    def total_bytes(self, name):
        """Algorithmic bytes (inputs read once + outputs written once) of the launches called `name`."""
        return sum(r[4] for r in self.records if r[0] == name)

    def total_side_bytes(self, name):
        """The other operands of the same launches, each counted once: weights (split-bf16: 4 bytes per element), the residual,
        the pre-activation addend (`pre`: one image per `pre_div` outputs) and the old output of an accumulating launch."""
        return sum(r[5] for r in self.records if r[0] == name)


KERNEL_TIMER = None  # set to a KernelTimer() to time conv launches
# Packed split-bf16 intermediates in the conv -> conv chains of the InterBlocks (P8Act): 49|99 -> 128 -> 64 -> 32 -> heads.  Bit-identical
# to fp32 intermediates.  ON since round 4 (CCVS_CONV_P8=0: fp32): the consumers stage by LDS-DMA with no conversion (128->64 at
# 256^2 298 -> 382 TFLOP/s on the 512-pixel tile, 64->32 205 -> 250), the producers' packed epilogue no longer waits on vector memory
# between its stores (it cost them 20 % in round 3: that, not the format, made P8 a net loss then); a BAIR batch's convolutions 617 ->
# 570 ms alone, the default bench line 190.7 -> 200.4 frames/s on one box (profiles/r04_conv_p8_ab.txt).
CONV_P8 = __import__("os").environ.get("CCVS_CONV_P8", "1") == "1"
# ... and the back-warp in front of the first Subpixel convolution can write that convolution's packed input (`backwarp_p8`).  OFF:
# the convolution gains (99->128 at 256^2 268 -> 306 TFLOP/s, 13 ms per decode) what the packed back-warp loses (3.85 against 2.9 ms
# per 120 x 96 x 256^2 call in its best form of three -- one pixel per lane, four pixels per lane, four pixels + LDS transposition);
# the default line 197.0 against 197.1 frames/s (profiles/r04_p8_warp_ab.txt)
P8_WARP = __import__("os").environ.get("CCVS_P8_WARP", "0") == "1"


def conv_persistent_tiles(mode=-1):
    """Which 3 x 3 layers of the split-bf16 convolution run as resident workgroups (`ccvs_conv_persistent_tiles`: bit 0 fp32-input
    128-channel layers, bit 1 packed-input layers); process-wide, returns the previous mode (mode < 0: query).  The library's default
    is 1; `helpers/pipeline.py` sets 0 while a run has several batches in flight.  `CCVS_CONV_PT` in the environment overrides both."""
    return int(_lib.load().ccvs_conv_persistent_tiles(int(mode)))


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _p(t):
    return C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)


def _need_gpu(*ts):
    for t in ts:
        if t is not None and not t.is_cuda:
            raise _lib.CcvsError("ccvs_amd ops need device tensors: the HIP path has no CPU fallback")


def _rows_dense(t):
    return t.stride(-1) == 1 and t.stride(-2) == t.shape[-1]


def _as_rows_dense(t):
    return t if _rows_dense(t) else t.contiguous()


# ------------------------------------------------------------------ convolution
# Arithmetic of the convolution kernels:
#   "bf16x3": split-bf16, 3 products per fp32 product on v_mfma_f32_32x32x16_bf16 (default)
#   "f32"   : exact fp32 on v_mfma_f32_32x32x2_f32
CONV_PRECISION = "bf16x3"
# Matching: warp + 1x1 projection of the contexts in one kernel (`backwarp_proj`) instead of backwarp followed by the convolution.
FUSE_WARP_PROJ = __import__("os").environ.get("CCVS_FUSE_WARP_PROJ", "1") == "1"
# Explicit `ccvs_conv_desc.cu_limit` of the convolution launches (0 = the budget of the stream, see `stream_cu_limit`); tests.
CONV_CU_LIMIT = 0


class PackedConv:
    """Kernel-ready weights of one convolution (built once per parameter version).  `ktail`: the same weights with the
    last 16-channel chunk in the packed-tail order (`ccvs_conv_desc.w_ktail`), or None."""
    __slots__ = ("kind", "data", "cin", "cout", "cout_pad", "k", "kw", "ktail")

    def __init__(self, kind, data, cin, cout, cout_pad, k, kw=None, ktail=None):
        self.kind, self.data, self.cin, self.cout, self.cout_pad, self.k = kind, data, cin, cout, cout_pad, k
        self.kw = k if kw is None else kw
        self.ktail = ktail


class P8Act:
    """A split-bf16 packed activation (`ccvs_conv_desc.in_p8 / out_p8`): logical [n, c, h, w] fp32 values stored as
    [n][c/8][hi|lo][h][w] units of 8 bf16 -- what the convolution kernel stages in LDS, so a conv -> conv chain moves its
    intermediate tensors with no conversion work.  `data` is an opaque float32 buffer of n*c*h*w elements."""
    __slots__ = ("data", "n", "c", "h", "w")

    def __init__(self, data, n, c, h, w):
        self.data, self.n, self.c, self.h, self.w = data, n, c, h, w

    @property
    def shape(self):
        return (self.n, self.c, self.h, self.w)

    def float(self):
        """Decode to an fp32 [n,c,h,w] tensor (hi + lo) -- tests / debugging only."""
        u = self.data.view(torch.bfloat16).view(self.n, self.c // 8, 2, self.h, self.w, 8).float()
        v = u[:, :, 0] + u[:, :, 1]                                   # [n, g, h, w, 8]
        return v.permute(0, 1, 4, 2, 3).reshape(self.n, self.c, self.h, self.w).contiguous()


def pack_conv_weight(weight, precision=None, scale=None):
    """[Cout,Cin,k,k] parameter -> PackedConv, with the EqualConv2d scale 1/sqrt(Cin*k*k) multiplied in
    first (the same fp32 product as `weight * scale`, skip_autoencoder.py:44,55,58).
      f32   : [k*k][Cin][CoutPad] fp32
      bf16x3: [k*k][CinPad/8][hi|lo][CoutPad][8] bf16, w = hi + lo (both round-to-nearest-even)"""
    precision = precision or CONV_PRECISION
    cout, cin, kh, kw = weight.shape
    if scale is None:
        scale = 1 / math.sqrt(cin * kh * kw)
    w = (weight.detach().float() * scale).permute(2, 3, 1, 0).reshape(kh * kw, cin, cout)
    if precision == "f32":
        cpad = -(-cout // 64) * 64 if cout >= 64 else 32 * (-(-cout // 32))
        out = torch.zeros(kh * kw, cin, cpad, dtype=torch.float32, device=weight.device)
        out[:, :, :cout] = w
        return PackedConv("f32", out.contiguous(), cin, cout, cpad, kh, kw)
    if precision != "bf16x3":
        raise ValueError(f"unknown conv precision {precision!r}")
    cpad = 32 * (-(-cout // 32))
    cinp = 16 * (-(-cin // 16))
    full = torch.zeros(kh * kw, cinp, cpad, dtype=torch.float32, device=weight.device)
    full[:, :cin, :cout] = w
    hi = full.to(torch.bfloat16)
    lo = (full - hi.float()).to(torch.bfloat16)
    lay = lambda t: t.view(kh * kw, cinp // 8, 8, cpad).permute(0, 1, 3, 2)
    out = torch.stack([lay(hi), lay(lo)], dim=2).contiguous()  # [tap][cg][2][cpad][8]
    ktail = None
    r = cin % 16
    if kh == 3 and kw == 3 and 1 <= r <= 3 and cin > 16:
        # Packed K tail (include/ccvs_hip.h, ccvs_conv_desc.w_ktail): the last chunk's 9 taps x r channels as ceil(9r/16)
        # steps whose K position kk of step j holds (tap t, channel c) with 16 j + kk = t r + c; rest of the chunk zero.
        tail = torch.zeros(kh * kw, 16, cpad, dtype=torch.float32, device=weight.device)   # [tap slot j][kk][cout]
        q = torch.arange(9 * r, device=weight.device)
        tail.view(-1, cpad)[q] = full[q // r, cin - r + q % r]
        full_t = full.clone()
        full_t[:, cinp - 16:] = tail
        hi_t = full_t.to(torch.bfloat16)
        lo_t = (full_t - hi_t.float()).to(torch.bfloat16)
        ktail = torch.stack([lay(hi_t), lay(lo_t)], dim=2).contiguous()
    return PackedConv("bf16x3", out, cin, cout, cpad, kh, kw, ktail=ktail)


def conv2d(x, w_packed, bias, cout, k, stride=1, pad=0, transposed=False, act=False, residual=None,
           out_scale=1.0, out=None, accumulate=False, pre=None, pre_div=1, out_p8=False):
    """y = [y +] (act(conv(x) [+ pre[n // pre_div]] + bias) [+ residual]) * out_scale.
    x may be a P8Act (packed split-bf16 input); out_p8=True returns one (split-bf16 kernel only)."""
    in_p8 = isinstance(x, P8Act)
    if in_p8:
        _need_gpu(x.data, w_packed.data, bias, residual, out, pre)
        assert w_packed.kind == "bf16x3" and stride == 1 and not transposed
    else:
        _need_gpu(x, w_packed.data, bias, residual, out, pre)
        x = _as_rows_dense(x)
    n, cin, h, w = x.shape
    assert w_packed.k == k and w_packed.cin == cin and w_packed.cout == cout, (w_packed.k, w_packed.cin, w_packed.cout, k, cin, cout)
    kw = w_packed.kw
    if transposed:
        ho, wo = 2 * h + k - 2, 2 * w + k - 2
    else:
        ho, wo = (h + 2 * pad - k) // stride + 1, (w + 2 * pad - kw) // stride + 1
    dev = x.data.device if in_p8 else x.device
    if out_p8:
        assert w_packed.kind == "bf16x3" and cout % 8 == 0 and out is None and residual is None and not accumulate
        out = torch.empty(n * cout * ho * wo, dtype=torch.float32, device=dev)
    elif out is None:
        out = torch.empty(n, cout, ho, wo, dtype=torch.float32, device=dev)
    if not out_p8:
        assert out.shape == (n, cout, ho, wo) and _rows_dense(out), (out.shape, (n, cout, ho, wo))
    d = _lib.ConvDesc()
    d.N, d.Cin, d.Hin, d.Win = n, cin, h, w
    d.in_p8, d.out_p8 = (1 if in_p8 else 0), (1 if out_p8 else 0)
    if not in_p8:
        d.in_sN, d.in_sC = x.stride(0), x.stride(1)
    d.Cout, d.CoutPad, d.Hout, d.Wout = cout, w_packed.cout_pad, ho, wo
    if not out_p8:
        d.out_sN, d.out_sC = out.stride(0), out.stride(1)
    if residual is not None:
        residual = _as_rows_dense(residual)
        assert residual.shape == out.shape
        d.res_sN, d.res_sC = residual.stride(0), residual.stride(1)
    if pre is not None:
        pre = _as_rows_dense(pre)
        assert pre.shape == (n // pre_div, cout, ho, wo) and n % pre_div == 0, (pre.shape, out.shape, pre_div)
        d.pre, d.pre_sN, d.pre_sC, d.pre_div = pre.data_ptr(), pre.stride(0), pre.stride(1), pre_div
    d.kh, d.kw = k, kw
    d.stride, d.pad, d.transposed = stride, pad, 1 if transposed else 0
    d.act = ACT_LRELU if act else ACT_NONE
    d.accumulate = 1 if accumulate else 0
    d.out_scale = out_scale
    d.cu_limit = CONV_CU_LIMIT
    kt = getattr(w_packed, "ktail", None)
    if kt is not None and stride == 1 and not transposed and not in_p8:
        d.w_ktail = kt.data_ptr()
    L = _lib.load()
    prof = KERNEL_TIMER
    if prof is not None:
        macs = n * cout * cin * k * kw * (h * w if transposed else ho * wo)
        side = 4.0 * (cin * cout * k * kw + n * cout * ho * wo * ((residual is not None) + bool(accumulate))
                      + (n // pre_div * cout * ho * wo if pre is not None else 0))
        prof.begin("conv2d_" + w_packed.kind, flops=2.0 * macs, nbytes=4.0 * (n * cin * h * w + n * cout * ho * wo), side=side,
                   tag=(n, cin, h, w, cout, k, kw, stride, int(bool(transposed)), int(in_p8), int(bool(out_p8)), int(pre is not None),
                        int(residual is not None), int(bool(accumulate))))
    fn = L.ccvs_conv2d if w_packed.kind == "f32" else L.ccvs_conv2d_bf16x3
    _lib.check(fn(_p(x.data if in_p8 else x), _p(w_packed.data), _p(bias), _p(residual), _p(out), C.byref(d), _stream()),
               "ccvs_conv2d[" + w_packed.kind + "]")
    if prof is not None:
        prof.end()
    return P8Act(out, n, cout, ho, wo) if out_p8 else out


def pack_head_weights(flow_w, occ_w, precision=None):
    """flow_head [2,C,k,k] + occ_head [1,C,k,k] -> PackedConv of an equivalent convolution with 3k outputs whose second
    half is ccvs_tap_shift_add: split-bf16: a 1 x k kernel (row ky*3+co = tap row ky of output co; all k taps of a row
    in one step of the kernel); f32: a k x 1 kernel (row kx*3+co = tap column kx of output co)."""
    precision = precision or CONV_PRECISION
    w = torch.cat([flow_w.detach(), occ_w.detach()], dim=0).float()          # [3, C, k(y), k(x)]
    _, cin, k, _ = w.shape
    if precision == "bf16x3":
        wt = w.permute(2, 0, 1, 3).reshape(3 * k, cin, 1, k).contiguous()      # [(ky,co), C, 1, kx]
    else:
        wt = w.permute(3, 0, 1, 2).reshape(3 * k, cin, k, 1).contiguous()      # [(kx,co), C, ky, 1]
    return pack_conv_weight(wt, precision, scale=1 / math.sqrt(cin * k * k))


def conv_heads(feat, w_packed, bias3, out, accumulate, flops_cb=None):
    """The fused flow/occ heads: MFMA convolution to 3k maps, then the tap sum along the other axis into
    `out` ([N,3,H,W] view, batch stride free)."""
    vertical = w_packed.k == 1 and w_packed.kw > 1
    k = w_packed.kw if vertical else w_packed.k
    n, _, h, w = feat.shape
    t = conv2d(feat, w_packed, None, 3 * k, w_packed.k, pad=k // 2)             # [N,3k,H+k-1,W] or [N,3k,H,W+k-1]
    assert out.shape == (n, 3, h, w) and _planes_dense(out)
    L = _lib.load()
    _lib.check(L.ccvs_tap_shift_add(_p(t), _p(bias3), _p(out), out.stride(0), n, k, h, w, 1 if accumulate else 0,
                                    1 if vertical else 0, _stream()), "ccvs_tap_shift_add")
    return out


# ------------------------------------------------------------------ resampling
def upfirdn2d(x, up=1, down=1, pad=(0, 0), gain=1.0, act=False, residual=None, out_scale=1.0):
    _need_gpu(x, residual)
    x = x.contiguous()
    n, c, h, w = x.shape
    ho = (h * up + pad[0] + pad[1] - 4) // down + 1
    wo = (w * up + pad[0] + pad[1] - 4) // down + 1
    y = torch.empty(n, c, ho, wo, dtype=torch.float32, device=x.device)
    if residual is not None:
        residual = residual.contiguous()
        assert residual.shape == y.shape
    L = _lib.load()
    _lib.check(L.ccvs_upfirdn2d(_p(x), _p(y), _p(residual), n * c, h, w, up, down, pad[0], pad[1], gain,
                                ACT_LRELU if act else ACT_NONE, out_scale, _stream()), "ccvs_upfirdn2d")
    return y


def _planes_dense(t):
    """[N,C,H,W] whose channel planes are dense and consecutive (only the batch stride is free)."""
    return _rows_dense(t) and (t.shape[1] == 1 or t.stride(1) == t.shape[2] * t.shape[3])


def dwconvT4x4s2(x, w, out=None):
    _need_gpu(x, w, out)
    if not _planes_dense(x):
        x = x.contiguous()
    n, c, h, ww = x.shape
    if out is None:
        out = torch.empty(n, c, 2 * h, 2 * ww, dtype=torch.float32, device=x.device)
    assert out.shape == (n, c, 2 * h, 2 * ww) and _planes_dense(out)
    L = _lib.load()
    _lib.check(L.ccvs_dwconvT4x4s2(_p(x), x.stride(0), _p(w.contiguous()), _p(out), out.stride(0), n, c, h, ww, _stream()),
               "ccvs_dwconvT4x4s2")
    return out


# ------------------------------------------------------------------ cost volume / warp
def correlation7x7(first, second, stride, first_div=1, lrelu=False):
    _need_gpu(first, second)
    first, second = first.contiguous(), second.contiguous()
    n, c, h, w = second.shape
    assert first.shape[0] * first_div == n and first.shape[1:] == second.shape[1:]
    out = torch.empty(n, 49, -(-h // stride), -(-w // stride), dtype=torch.float32, device=second.device)
    L = _lib.load()
    _lib.check(L.ccvs_correlation7x7(_p(first), _p(second), _p(out), n, c, h, w, stride, first_div, 1 if lrelu else 0, _stream()),
               "ccvs_correlation7x7")
    return out


def _ctx_list(ctxs):
    """k context tensors [N,C,H,W] (channel planes dense, any batch stride) -> (`ccvs_ctx_list`, tensors kept alive)."""
    assert 1 <= len(ctxs) <= _lib.MAX_CTX
    cl = _lib.CtxList()
    cl.k = len(ctxs)
    keep = []
    n, c, h, w = ctxs[0].shape
    for j, t in enumerate(ctxs):
        _need_gpu(t)
        assert t.shape == (n, c, h, w) and t.dtype == torch.float32
        if not (t.stride(3) == 1 and t.stride(2) == w and t.stride(1) == h * w):
            t = t.contiguous()
        keep.append(t)
        cl.p[j] = t.data_ptr()
        cl.sN[j] = t.stride(0)
    return cl, keep


def backwarp(x, flow, flow_mult=1.0, out=None):
    """x: a tensor [N,C,H,W], or a LIST of k context tensors [N/k,C,H,W] standing for the batch of N pairs in (frame, context)
    order -- read in place, no stacked copy."""
    _need_gpu(flow, out)
    if not _planes_dense(flow):
        flow = flow.contiguous()
    L = _lib.load()
    if isinstance(x, (list, tuple)):
        cl, keep = _ctx_list(x)
        nf, c, h, w = keep[0].shape
        n = nf * cl.k
        assert flow.shape == (n, 2, h, w)
        if out is None:
            out = torch.empty(n, c, h, w, dtype=torch.float32, device=flow.device)
        assert _rows_dense(out)
        _lib.check(L.ccvs_backwarp_ctx(C.byref(cl), h * w, _p(flow), flow.stride(0), flow_mult, _p(out), out.stride(0), out.stride(1),
                                       n, c, h, w, _stream()), "ccvs_backwarp_ctx")
        return out
    _need_gpu(x)
    x = _as_rows_dense(x)
    n, c, h, w = x.shape
    assert flow.shape == (n, 2, h, w)
    if out is None:
        out = torch.empty(n, c, h, w, dtype=torch.float32, device=x.device)
    assert _rows_dense(out)
    _lib.check(L.ccvs_backwarp(_p(x), x.stride(0), x.stride(1), _p(flow), flow.stride(0), flow_mult, _p(out), out.stride(0),
                               out.stride(1), n, c, h, w, _stream()), "ccvs_backwarp")
    return out


def pack_proj_weight(weight):
    """1x1 EqualConv2d weight [Cout,Cin,1,1] -> (w_t [Cin][CoutPad] with the 1/sqrt(Cin) scale multiplied in, CoutPad) for
    `backwarp_proj`; None when Cout has no instantiation."""
    cout, cin = weight.shape[:2]
    pads = [p_ for p_ in (16, 24, 48, 96) if p_ >= cout]
    if not pads:
        return None
    w = (weight.detach().float().view(cout, cin) * (1 / math.sqrt(cin))).t()
    out = torch.zeros(cin, pads[0], dtype=torch.float32, device=weight.device)
    out[:, :cout] = w
    return out.contiguous(), pads[0]


def backwarp_p8(ctxs, flow_occ, flow_mult):
    """[backwarp(ctx, flow * flow_mult) | flow | occ | 0 x 5] for the list of k context tensors [N/k,C,H,W] (see `backwarp`) as a P8Act of
    C + 8 channels (`ccvs_backwarp_p8_ctx`): the packed input of the first Subpixel convolution.  flow_occ: [N,3,H,W]."""
    _need_gpu(flow_occ)
    if not _planes_dense(flow_occ):
        flow_occ = flow_occ.contiguous()
    cl, keep = _ctx_list(ctxs)
    nf, c, h, w = keep[0].shape
    n = nf * cl.k
    assert flow_occ.shape == (n, 3, h, w) and c % 8 == 0 and w % 4 == 0
    out = torch.empty(n * (c + 8) * h * w, dtype=torch.float32, device=flow_occ.device)
    L = _lib.load()
    _lib.check(L.ccvs_backwarp_p8_ctx(C.byref(cl), h * w, _p(flow_occ), flow_occ.stride(0), flow_mult, _p(out), n, c, h, w, _stream()),
               "ccvs_backwarp_p8_ctx")
    return P8Act(out, n, c + 8, h, w)


def backwarp_proj(ctxs, flow, flow_mult, w_t, cout_pad, bias, cout, act=True):
    """act(bias + W . backwarp(ctx, flow * flow_mult)) for the list of k context tensors [N/k,C,H,W] (see `backwarp`):
    [N,cout,H,W], the warped tensor is never materialised (`ccvs_backwarp_proj_ctx`)."""
    _need_gpu(flow, w_t, bias)
    if not _planes_dense(flow):
        flow = flow.contiguous()
    cl, keep = _ctx_list(ctxs)
    nf, c, h, w = keep[0].shape
    n = nf * cl.k
    assert flow.shape == (n, 2, h, w) and w_t.shape == (c, cout_pad)
    out = torch.empty(n, cout, h, w, dtype=torch.float32, device=flow.device)
    L = _lib.load()
    _lib.check(L.ccvs_backwarp_proj_ctx(C.byref(cl), h * w, _p(flow), flow.stride(0), flow_mult, _p(w_t), _p(bias), _p(out), n, c, cout, cout_pad,
                                        h, w, ACT_LRELU if act else ACT_NONE, _stream()), "ccvs_backwarp_proj_ctx")
    return out


def warp_fuse_blend(dec, ctx, flows, occs, flow_mult, k):
    """In place on `dec` (a channel-slice view [N,C,H,W] of the decoder feature).  ctx: [N*k,C,H,W], or a list of k
    context tensors [N,C,H,W] read in place."""
    _need_gpu(dec, flows, occs)
    assert _rows_dense(dec)
    flows = flows if _planes_dense(flows) else flows.contiguous()
    occs = occs if _planes_dense(occs) else occs.contiguous()
    n, c, h, w = dec.shape
    assert flows.shape == (n * k, 2, h, w) and occs.shape == (n * k, 1, h, w)
    L = _lib.load()
    if isinstance(ctx, (list, tuple)):
        cl, keep = _ctx_list(ctx)
        assert cl.k == k and keep[0].shape == (n, c, h, w)
        _lib.check(L.ccvs_warp_fuse_blend_ctx(_p(dec), dec.stride(0), dec.stride(1), C.byref(cl), _p(flows), flows.stride(0), _p(occs),
                                              occs.stride(0), flow_mult, n, c, h, w, _stream()), "ccvs_warp_fuse_blend_ctx")
        return dec
    _need_gpu(ctx)
    ctx = ctx.contiguous()
    assert ctx.shape == (n * k, c, h, w)
    _lib.check(L.ccvs_warp_fuse_blend(_p(dec), dec.stride(0), dec.stride(1), _p(ctx), _p(flows), flows.stride(0), _p(occs),
                                      occs.stride(0), flow_mult, n, k, c, h, w, _stream()), "ccvs_warp_fuse_blend")
    return dec


# ------------------------------------------------------------------ vector quantiser
def vq_argmin(z, codebook_t, e_sq):
    """z [N,C,H,W] -> int64 [N*H*W] (n,h,w raster order)."""
    _need_gpu(z, codebook_t, e_sq)
    z = z.contiguous()
    n, c = z.shape[:2]
    hw = z.shape[2] * z.shape[3]
    idx = torch.empty(n * hw, dtype=torch.int64, device=z.device)
    L = _lib.load()
    _lib.check(L.ccvs_vq_argmin(_p(z), _p(codebook_t), _p(e_sq), _p(idx), n, c, hw, codebook_t.shape[1], _stream()), "ccvs_vq_argmin")
    return idx


def embed_gather(code, codebook, n, hw):
    """code int64 [n*hw] -> z [n, C, hw]."""
    _need_gpu(code, codebook)
    code = code.contiguous()
    n_e, c = codebook.shape
    z = torch.empty(n, c, hw, dtype=torch.float32, device=codebook.device)
    L = _lib.load()
    _lib.check(L.ccvs_embed_gather(_p(code), _p(codebook), _p(z), n, c, hw, n_e, _stream()), "ccvs_embed_gather")
    return z


def l2_normalize_channels_(x):
    """x [n, C, h, w] (dense) /= its L2 norm over the channels, in place (`normalize_out`, skip_autoencoder.py:348-349)."""
    _need_gpu(x)
    assert x.is_contiguous() and x.dtype == torch.float32 and x.dim() == 4
    n, c, h, w = x.shape
    L = _lib.load()
    _lib.check(L.ccvs_l2_normalize_channels(_p(x), n, c, h * w, _stream()), "ccvs_l2_normalize_channels")
    return x


# ------------------------------------------------------------------ transformer
def gpt_embed(idx, tok_emb, pos_table, pos0=0, pos_off=None, pos_dev=None):
    """idx int64 [B,Tq] (row stride free) -> x [B*Tq, C]; positional row = pos_off[b] + pos0 (+ *pos_dev) + t."""
    _need_gpu(idx, pos_off, tok_emb, pos_table, pos_dev)
    b, tq = idx.shape
    assert idx.stride(1) == 1 or tq == 1
    c = tok_emb.shape[1]
    x = torch.empty(b * tq, c, dtype=torch.float32, device=tok_emb.device)
    L = _lib.load()
    _lib.check(L.ccvs_gpt_embed(_p(idx), idx.stride(0), _p(pos_off), pos0, _p(pos_dev), tq, _p(tok_emb), _p(pos_table), _p(x), b, c,
                                tok_emb.shape[0], _stream()), "ccvs_gpt_embed")
    return x


def layernorm(x, gamma, beta, out=None):
    _need_gpu(x, gamma, beta)
    rows, c = x.shape
    assert x.is_contiguous()
    if out is None:
        out = torch.empty_like(x)
    L = _lib.load()
    _lib.check(L.ccvs_layernorm(_p(x), _p(gamma), _p(beta), _p(out), rows, c, _stream()), "ccvs_layernorm")
    return out


def gemm_nt(x, w, bias=None, epilogue=EPI_NONE, residual=None, out=None):
    """y = epilogue(x @ w.T + bias); x [M,K] (row stride free), w [N,K] contiguous."""
    _need_gpu(x, w, bias, residual, out)
    m, k = x.shape
    n = w.shape[0]
    assert x.stride(1) == 1 and w.is_contiguous() and w.shape[1] == k
    if out is None:
        out = torch.empty(m, n, dtype=torch.float32, device=x.device)
    assert out.stride(1) == 1
    if residual is not None:
        assert residual.stride(1) == 1 and residual.stride(0) == out.stride(0)
    L = _lib.load()
    _lib.check(L.ccvs_gemm_nt(_p(x), x.stride(0), _p(w), _p(bias), _p(residual), _p(out), out.stride(0), m, n, k, epilogue,
                              _p(_gemm_workspace(x.device)), _stream()), "ccvs_gemm_nt")  # workspace is per (device, stream)
    return out


_GEMM_WS = {}


def _gemm_workspace(device):
    """Zero-initialised split-K workspace (slabs + arrival counters), one per (device, stream): calls on
    one stream are ordered and the reducer leaves the counters at zero, but two streams (the parallel
    decode lanes of GPT.generate) may run the same GEMM concurrently and must not share slabs."""
    key = (device, torch.cuda.current_stream().cuda_stream)
    ws = _GEMM_WS.get(key)
    if ws is None:
        ws = _GEMM_WS[key] = torch.zeros(int(_lib.load().ccvs_gemm_workspace_bytes()), dtype=torch.uint8, device=device)
    return ws


def pack_ln_linear(weight, bias, gamma, beta):
    """Fold a LayerNorm (gamma, beta) into the Linear (weight [N,K], bias [N]) that follows it:
    returns (W*gamma, bias + W@beta, rowsum(W*gamma)) -- see ccvs_gemm_ln in include/ccvs_hip.h."""
    w = weight.detach().float()
    wg = (w * gamma.detach().float().view(1, -1)).contiguous()
    b = bias.detach().float() if bias is not None else torch.zeros(w.shape[0], device=w.device)
    bb = (b + w.double().matmul(beta.detach().double()).float()).contiguous()
    s = wg.double().sum(dim=1).float().contiguous()
    return wg, bb, s


def gemm_ln(x, wg, bb, s, eps=1e-5, epilogue=EPI_NONE, out=None):
    """epilogue(LayerNorm(x) @ W.T + b) with the LayerNorm folded into (wg, bb, s) = pack_ln_linear(...)."""
    _need_gpu(x, wg, bb, s, out)
    m, k = x.shape
    n = wg.shape[0]
    assert x.stride(1) == 1 and wg.is_contiguous() and wg.shape[1] == k
    if out is None:
        out = torch.empty(m, n, dtype=torch.float32, device=x.device)
    L = _lib.load()
    _lib.check(L.ccvs_gemm_ln(_p(x), x.stride(0), _p(wg), _p(bb), _p(s), eps, _p(out), out.stride(0), m, n, k, epilogue, _stream()),
               "ccvs_gemm_ln")
    return out


def gemm_ln_qkv(x, wg, bb, s, kcache, vcache, b, tq, pos0, pos_dev=None, eps=1e-5):
    """ln1 + fused QKV projection: returns q [b*tq, C]; K / V go straight into the caches."""
    _need_gpu(x, wg, bb, s, kcache, vcache, pos_dev)
    c = x.shape[1]
    _, h, tmax, d = kcache.shape
    assert x.shape[0] == b * tq and x.stride(1) == 1 and wg.shape == (3 * c, c) and h * d == c
    q = torch.empty(b * tq, c, dtype=torch.float32, device=x.device)
    L = _lib.load()
    _lib.check(L.ccvs_gemm_ln_qkv(_p(x), x.stride(0), _p(wg), _p(bb), _p(s), eps, _p(q), _p(kcache), _p(vcache), b, tq, c, h, pos0,
                                  _p(pos_dev), tmax, _stream()), "ccvs_gemm_ln_qkv")
    return q


def kv_append(k, v, kcache, vcache, pos0, pos_dev=None):
    """k, v [B,Tq,H*D] views (row stride shared) -> caches [B,H,Tmax,D] at pos0 (+ *pos_dev).."""
    b, tq, hd = k.shape
    _, h, tmax, d = kcache.shape
    assert k.stride(2) == 1 and v.stride() == k.stride()
    L = _lib.load()
    _lib.check(L.ccvs_kv_append(_p(k), _p(v), k.stride(0), k.stride(1), _p(kcache), _p(vcache), b, h, tq, pos0, _p(pos_dev), tmax, d,
                                _stream()), "ccvs_kv_append")


def attention(q, kcache, vcache, pos0, pos_dev=None):
    """q [B,Tq,H*D] view -> out [B,Tq,H*D]; query t sees cache positions 0..pos0(+*pos_dev)+t."""
    b, tq, hd = q.shape
    _, h, tmax, d = kcache.shape
    assert q.stride(2) == 1
    out = torch.empty(b, tq, hd, dtype=torch.float32, device=q.device)
    L = _lib.load()
    _lib.check(L.ccvs_attention(_p(q), q.stride(0), q.stride(1), _p(kcache), _p(vcache), _p(out), b, h, tq, pos0, _p(pos_dev), tmax, d,
                                _stream()), "ccvs_attention")
    return out


def sample_topk(logits, top_k, temperature, noise=None, out=None, philox=None):
    """logits [B,V] -> int64 [B]; noise None = greedy, else argmax(p / noise).  philox = (key0, key1, row0, step, call):
    draw the Exp(1) noise in the kernel, keyed by the global clip index row0 + b (ccvs_sample_topk_philox)."""
    _need_gpu(logits, noise, out)
    b, v = logits.shape
    assert logits.stride(1) == 1
    if out is None:
        out = torch.empty(b, dtype=torch.int64, device=logits.device)
    if philox is not None:
        assert noise is None
        k0, k1, row0, step, call = (int(t) & 0xffffffff for t in philox)
        L = _lib.load()
        _lib.check(L.ccvs_sample_topk_philox(_p(logits), logits.stride(0), _p(out), out.stride(0), b, v, 0 if top_k is None else int(top_k),
                                             float(temperature), k0, k1, row0, step, call, _stream()), "ccvs_sample_topk_philox")
        return out
    if noise is not None:
        assert noise.shape == (b, v) and noise.is_contiguous()
    L = _lib.load()
    _lib.check(L.ccvs_sample_topk(_p(logits), logits.stride(0), _p(noise), _p(out), out.stride(0), b, v,
                                  0 if top_k is None else int(top_k), float(temperature), _stream()), "ccvs_sample_topk")
    return out


def sample_topn(logits, top_k, temperature, n, noise=None):
    """logits [B,V] -> (picks int64 [B,n], log p float [B,n]): the n best of the top-k softmax (noise None) or of probs / noise
    (torch.multinomial without replacement), best first (`ccvs_sample_topn`: the proposals of beam search)."""
    _need_gpu(logits, noise)
    b, v = logits.shape
    assert logits.stride(1) == 1 and 1 <= n <= v
    if noise is not None:
        assert noise.shape == (b, v) and noise.is_contiguous()
    idx = torch.empty(b, n, dtype=torch.int64, device=logits.device)
    logp = torch.empty(b, n, dtype=torch.float32, device=logits.device)
    L = _lib.load()
    _lib.check(L.ccvs_sample_topn(_p(logits), logits.stride(0), _p(noise), _p(idx), _p(logp), b, v, 0 if top_k is None else int(top_k),
                                  float(temperature), int(n), _stream()), "ccvs_sample_topn")
    return idx, logp


class GptDecodeStep:
    """A filled `ccvs_gpt_decode` descriptor (include/ccvs_hip.h) plus the tensors it points at.
    `launch()` enqueues one whole decode step on the current stream."""

    def __init__(self, layers, B, C, H, Tmax, ln_eps, tok_emb, pos_table, pos_off, head, tok, codes, widx, length,
                 x, q, att, h, logits, noise, top_k, temperature, state, rng=False, groups=1, noise_stream=None, persistent=False):
        hw, hb, hs = head
        keep = [tok_emb, pos_table, hw, hb, hs, tok, codes, widx, length, x, q, att, h, logits, noise, state, noise_stream]
        for t in keep:
            if t is not None:
                _need_gpu(t)
        f32 = [tok_emb, pos_table, hw, hb, hs, x, q, att, h, logits] + ([noise] if noise is not None else [])
        assert all(t.dtype == torch.float32 and t.is_contiguous() for t in f32)
        assert tok.dtype == torch.int64 and tok.numel() == B and tok.is_contiguous() and codes.dtype == torch.int64 and codes.stride(1) == 1
        assert widx.dtype == torch.int32 and length.dtype == torch.int32 and state.dtype == torch.int32
        assert groups >= 1 and B % groups == 0 and widx.numel() == groups and length.numel() == groups and state.numel() == 8 * groups
        assert widx.is_contiguous() and length.is_contiguous() and state.is_contiguous()
        if noise_stream is not None:   # device table of `groups` device pointers (ccvs_gpt_decode.noise_stream)
            assert noise is None and noise_stream.dtype == torch.int64 and noise_stream.numel() == groups and noise_stream.is_contiguous()
        arr = (_lib.GptLayer * len(layers))()
        for i, lay in enumerate(layers):
            for name, t in lay.items():
                t = t.detach()
                _need_gpu(t)
                assert t.dtype == torch.float32 and t.is_contiguous(), name
                keep.append(t)
                setattr(arr[i], name, _p(t))
        F = layers[0]["fc_w"].shape[0]
        assert x.shape == (B, C) and h.shape == (B, F) and logits.shape == (B, hw.shape[0])
        self.ws = _gemm_workspace(x.device)
        d = _lib.GptDecode()
        d.B, d.C, d.H, d.F, d.n_layer, d.Tmax, d.vocab, d.V = B, C, H, F, len(layers), Tmax, tok_emb.shape[0], hw.shape[0]
        d.ln_eps = ln_eps
        d.layers = arr
        d.tok_emb, d.pos_table, d.pos_off = _p(tok_emb.detach()), _p(pos_table), int(pos_off)
        d.head_w, d.head_b, d.head_s = _p(hw), _p(hb), _p(hs)
        d.tok, d.codes, d.codes_sB = _p(tok), _p(codes), codes.stride(0)
        d.widx, d.len = _p(widx), _p(length)
        d.x, d.q, d.att, d.h, d.logits = _p(x), _p(q), _p(att), _p(h), _p(logits)
        d.noise = _p(noise)
        d.rng = 1 if (rng and noise is None and noise_stream is None) else 0
        d.noise_stream = _p(noise_stream)
        d.top_k, d.temperature = 0 if top_k is None else int(top_k), float(temperature)
        d.workspace, d.state = _p(self.ws), _p(state)
        d.groups = groups
        self.desc, self._arr, self._keep = d, arr, keep
        self.persistent = bool(persistent)
        if self.persistent:
            # ONE launch per step (gpt.hip: gpt_step_kernel): its phase table lives in a device buffer of ours, written once, here
            # (a synchronous copy on the current stream: descriptors are built outside graph capture)
            L = _lib.load()
            self.program = torch.empty(int(L.ccvs_gpt_program_bytes(len(layers))), dtype=torch.uint8, device=x.device)
            d.persistent, d.program = 1, _p(self.program)
            import ctypes   # (`C` is the embedding width in this scope)
            _lib.check(L.ccvs_gpt_decode_prepare(ctypes.byref(d), _stream()), "ccvs_gpt_decode_prepare")

    def launch(self):
        L = _lib.load()
        _lib.check(L.ccvs_gpt_decode_step(C.byref(self.desc), _stream()), "ccvs_gpt_decode_step")

    def status(self):
        """Persistent step: synchronise the current stream and raise if a grid barrier of any step launched with this workspace gave
        up (a workgroup that never became resident): the tokens of that step are invalid."""
        if self.persistent:
            L = _lib.load()
            _lib.check(L.ccvs_gpt_decode_status(_p(self.ws), _stream()), "ccvs_gpt_decode_status")


def stream_cu_limit(stream, cu_limit):
    """`ccvs_stream_cu_limit`: kernels launched on `stream` (a torch.cuda.Stream) from now on occupy at most cu_limit CUs
    (0: no budget)."""
    L = _lib.load()
    _lib.check(L.ccvs_stream_cu_limit(C.c_void_p(stream.cuda_stream), int(cu_limit)), "ccvs_stream_cu_limit")


def pack_u8_norm(vid, std, mean):
    """[..., 3, H, W] fp32 -> [..., H, W, 3] uint8 with the imagenet de-normalisation of helpers/generator.py:303-305 in front:
    v*std[c], + mean[c], clamp(0, 1), x255, truncate (each step rounded on its own: byte-exact with the reference's ops)."""
    _need_gpu(vid)
    vid = vid.contiguous()
    lead = vid.shape[:-3]
    h, w = vid.shape[-2:]
    n = int(math.prod(lead)) if len(lead) else 1
    out = torch.empty(*lead, h, w, 3, dtype=torch.uint8, device=vid.device)
    s3, m3 = (C.c_float * 3)(*[float(v) for v in std]), (C.c_float * 3)(*[float(v) for v in mean])
    L = _lib.load()
    _lib.check(L.ccvs_pack_u8_norm(_p(vid), _p(out), n, h, w, s3, m3, _stream()), "ccvs_pack_u8_norm")
    return out


def psnr(x, y, data_range=1.0):
    """Per-image PSNR of [N, C, H, W] fp32 tensors, piq.psnr's formula (tools/pytorch_metrics/metrics.py:24-25): [N] fp32."""
    _need_gpu(x, y)
    assert x.shape == y.shape and x.dim() == 4 and x.dtype == y.dtype == torch.float32, (x.shape, y.shape)
    x, y = x.contiguous(), y.contiguous()
    out = torch.empty(x.shape[0], dtype=torch.float32, device=x.device)
    _lib.check(_lib.load().ccvs_psnr(_p(x), _p(y), _p(out), x.shape[0], x[0].numel(), float(data_range), _stream()), "ccvs_psnr")
    return out


def ssim_planes(x, y, data_range=2.0):
    """skimage 0.17.2 `structural_similarity` (defaults) of every 2-D plane of [..., H, W] fp32 tensors: [...] fp64
    (tools/pytorch_metrics/metrics.py:15-22; data_range 2 = what skimage takes for float planes when none is given)."""
    _need_gpu(x, y)
    assert x.shape == y.shape and x.dim() >= 2 and x.dtype == y.dtype == torch.float32, (x.shape, y.shape)
    x, y = x.contiguous(), y.contiguous()
    h, w = x.shape[-2:]
    planes = x.numel() // (h * w)
    out = torch.empty(planes, dtype=torch.float64, device=x.device)
    L = _lib.load()
    step = 65535   # planes per call
    ws = torch.empty(int(L.ccvs_ssim_workspace_bytes(min(planes, step), h, w)), dtype=torch.uint8, device=x.device)
    xf, yf = x.view(planes, h, w), y.view(planes, h, w)
    for p0 in range(0, planes, step):
        n = min(step, planes - p0)
        _lib.check(L.ccvs_ssim(_p(xf[p0:]), _p(yf[p0:]), _p(out[p0:]), _p(ws), n, h, w, float(data_range), _stream()), "ccvs_ssim")
    return out.view(x.shape[:-2])


def resize_bilinear(x, size):
    """F.interpolate(x, size=size, mode='bilinear') (align_corners False) of [..., H, W] fp32 (tools/pytorch_metrics/metrics.py:124)."""
    _need_gpu(x)
    assert x.dtype == torch.float32 and x.dim() >= 2
    x = x.contiguous()
    h, w = x.shape[-2:]
    oh, ow = int(size[0]), int(size[1])
    out = torch.empty(*x.shape[:-2], oh, ow, dtype=torch.float32, device=x.device)
    _lib.check(_lib.load().ccvs_resize_bilinear(_p(x), _p(out), x.numel() // (h * w), h, w, oh, ow, _stream()), "ccvs_resize_bilinear")
    return out


def pack_u8(vid, lo=-1.0, hi=1.0):
    """[..., 3, H, W] fp32 -> [..., H, W, 3] uint8 (helpers/generator.py:306-309)."""
    _need_gpu(vid)
    vid = vid.contiguous()
    lead = vid.shape[:-3]
    h, w = vid.shape[-2:]
    n = int(math.prod(lead)) if len(lead) else 1
    out = torch.empty(*lead, h, w, 3, dtype=torch.uint8, device=vid.device)
    L = _lib.load()
    _lib.check(L.ccvs_pack_u8(_p(vid), _p(out), n, h, w, lo, hi, _stream()), "ccvs_pack_u8")
    return out
