"""Skip auto-encoder networks of CCVS on the MI355X kernel library.

Host-side mirror of the reference's `models/skip_vid_generator/models/skip_autoencoder.py`
(inference path only): same class names, constructor arguments, sub-module tree and hence
`state_dict()` keys, same initialisers in the same order.  The
`forward`s are not translations: every layer is a call into libccvs_hip.so (implicit-GEMM
MFMA convolution with fused bias / LeakyReLU(0.1) / residual epilogues, FIR resampling,
cost volume, back-warp, confidence fusion), and the torch.cat / repeat / one-hot plumbing of
the reference is replaced by kernels that read and write channel-slice views in place.

Not supported (outside the hot path): layout decoders, skip_rgb / ToRGB,
deformable conv, trade-off and masked-flow variants, `no_corr`, training.
"""
import math

import torch
from torch import nn

from ccvs_amd import ops
from ccvs_amd.tools.utils import flatten_vid, unflatten_vid

INV_SQRT2 = 1.0 / math.sqrt(2.0)


def make_kernel(k):
    """skip_autoencoder.py:19-24."""
    k = torch.tensor(k, dtype=torch.float32)
    if k.ndim == 1:
        k = k[None, :] * k[:, None]
    k /= k.sum()
    return k


class Blur(nn.Module):
    """4-tap FIR low-pass (skip_autoencoder.py:27-37); keeps the `kernel` buffer for state-dict parity."""

    def __init__(self, kernel, pad, upsample_factor=1):
        super().__init__()
        if list(kernel) != [1, 3, 3, 1]:
            raise NotImplementedError("Blur (HIP): only the [1,3,3,1] kernel is supported")
        kernel = make_kernel(kernel)
        self.gain = float(upsample_factor ** 2)
        if upsample_factor > 1:
            kernel = kernel * (upsample_factor ** 2)
        self.register_buffer("kernel", kernel)
        self.pad = pad

    def forward(self, input, act=False, residual=None, out_scale=1.0):
        return ops.upfirdn2d(input, pad=self.pad, gain=self.gain, act=act, residual=residual, out_scale=out_scale)


class EqualConv2d(nn.Module):
    """Parameter holder + packed-weight cache for the HIP conv (skip_autoencoder.py:40-63)."""

    def __init__(self, in_channel, out_channel, kernel_size, stride=1, padding=0, bias=True, transpose=False):
        super().__init__()
        self.weight = nn.Parameter(torch.randn(out_channel, in_channel, kernel_size, kernel_size))
        self.scale = 1 / math.sqrt(in_channel * kernel_size ** 2)
        self.stride, self.padding, self.transpose = stride, padding, transpose
        self.kernel_size, self.out_channel = kernel_size, out_channel
        self.bias = nn.Parameter(torch.zeros(out_channel)) if bias else None
        self._packed = None

    def packed(self):
        w = self.weight
        key = (w.data_ptr(), w._version, w.device, ops.CONV_PRECISION)
        if self._packed is None or self._packed[0] != key:
            self._packed = (key, ops.pack_conv_weight(w))
        return self._packed[1]

    def forward(self, input, act=False, residual=None, out_scale=1.0, out=None, accumulate=False, out_p8=False):
        if self.transpose and self.kernel_size == 1:
            raise NotImplementedError("1x1 transposed conv is folded into ConvLayer's FIR up-sampling")
        return ops.conv2d(input, self.packed(), self.bias, self.out_channel, self.kernel_size, stride=self.stride,
                          pad=self.padding, transposed=self.transpose, act=act, residual=residual, out_scale=out_scale,
                          out=out, accumulate=accumulate, out_p8=out_p8)

    def __repr__(self):
        return (f"{self.__class__.__name__}({self.weight.shape[1]}, {self.weight.shape[0]},"
                f" {self.weight.shape[2]}, stride={self.stride}, padding={self.padding})")


class ConvLayer(nn.Module):
    """[Blur] -> EqualConv2d -> [Blur] -> [LeakyReLU(0.1)] with the reference's child indices
    (skip_autoencoder.py:66-102).  `forward` runs the chain as at most two kernels and can fuse
    a residual add + scale (ResBlock) or an in-place accumulate (flow heads) into the last one."""

    def __init__(self, in_channel, out_channel, kernel_size, downsample=False, upsample=False, blur_kernel=[1, 3, 3, 1],
                 bias=True, activate=True):
        super().__init__()
        self.downsample, self.upsample, self.activate, self.kernel_size = downsample, upsample, activate, kernel_size
        layers = []
        if downsample:
            p = (len(blur_kernel) - 2) + (kernel_size - 1)
            layers.append(Blur(blur_kernel, pad=((p + 1) // 2, p // 2)))
            stride, transpose, padding = 2, False, 0
        elif upsample:
            stride, transpose, padding = 2, True, 0
        else:
            stride, transpose, padding = 1, False, kernel_size // 2
        self.padding = padding
        layers.append(EqualConv2d(in_channel, out_channel, kernel_size, padding=padding, stride=stride, bias=bias, transpose=transpose))
        if upsample:
            p = (len(blur_kernel) - 2) - (kernel_size - 1)
            layers.append(Blur(blur_kernel, pad=((p + 1) // 2 + 1, p // 2 + 1), upsample_factor=2))
        if activate:
            layers.append(nn.LeakyReLU(inplace=False, negative_slope=0.1))
        for i, layer in enumerate(layers):
            self.add_module(str(i), layer)
        # plain attribute aliases (not registered a second time as sub-modules)
        self.__dict__["conv"] = layers[1] if downsample else layers[0]
        self.__dict__["blur"] = layers[0] if downsample else (layers[1] if upsample else None)

    def forward(self, input, residual=None, out_scale=1.0, out=None, accumulate=False, out_p8=False):
        """`input` may be an ops.P8Act and `out_p8` asks for one (plain stride-1 layers on the split-bf16 kernel): packed
        intermediates of a conv -> conv chain, bit-identical to the fp32 chain."""
        conv, act = self.conv, self.activate
        assert not out_p8 or not (self.downsample or self.upsample)
        if self.downsample:
            if self.kernel_size == 1:
                # Blur(pad) then a stride-2 1x1 conv == decimating FIR then a dense 1x1 conv
                x = ops.upfirdn2d(input, down=2, pad=self.blur.pad)
                return ops.conv2d(x, conv.packed(), conv.bias, conv.out_channel, 1, act=act, residual=residual,
                                  out_scale=out_scale, out=out, accumulate=accumulate)
            x = self.blur(input)
            return conv(x, act=act, residual=residual, out_scale=out_scale, out=out, accumulate=accumulate)
        if self.upsample:
            assert out is None and not accumulate
            if self.kernel_size == 1:
                # conv_transpose2d(1x1, s2) + Blur(pad 2,2) == 1x1 conv + zero-insert FIR (up=2, pad 2,1)
                if conv.bias is not None:
                    raise NotImplementedError("1x1 up-sampling ConvLayer with bias")
                x = ops.conv2d(input, conv.packed(), None, conv.out_channel, 1)
                p0, p1 = self.blur.pad
                return ops.upfirdn2d(x, up=2, pad=(p0, p1 - 1), gain=self.blur.gain, act=act, residual=residual, out_scale=out_scale)
            x = conv(input)  # bias is added before the blur, like F.conv_transpose2d(bias=...)
            return self.blur(x, act=act, residual=residual, out_scale=out_scale)
        return conv(input, act=act, residual=residual, out_scale=out_scale, out=out, accumulate=accumulate, out_p8=out_p8)


class ResBlock(nn.Module):
    """skip_autoencoder.py:105-117: (conv2(conv1(x)) + skip(x)) / sqrt(2), the add and the scale
    fused into conv2's last kernel."""

    def __init__(self, in_channel, out_channel, blur_kernel=[1, 3, 3, 1], downsample=False, upsample=False):
        super().__init__()
        self.conv1 = ConvLayer(in_channel, in_channel, 3)
        self.conv2 = ConvLayer(in_channel, out_channel, 3, downsample=downsample, upsample=upsample, blur_kernel=blur_kernel)
        self.skip = ConvLayer(in_channel, out_channel, 1, downsample=downsample, upsample=upsample, blur_kernel=blur_kernel,
                              activate=False, bias=False)

    def forward(self, input):
        skip = self.skip(input)
        out = self.conv1(input)
        return self.conv2(out, residual=skip, out_scale=INV_SQRT2)


def get_backwarp_grid(height, width, device=None):
    """skip_autoencoder.py:120-123.  Kept for API parity; the HIP warp derives the grid in-kernel."""
    horizontal = torch.linspace(-1.0 + (1.0 / width), 1.0 - (1.0 / width), width).view(1, 1, 1, -1).expand(-1, -1, height, -1)
    vertical = torch.linspace(-1.0 + (1.0 / height), 1.0 - (1.0 / height), height).view(1, 1, -1, 1).expand(-1, -1, -1, width)
    grid = torch.cat([horizontal, vertical], dim=1)
    return grid.to(device) if device is not None else grid


def backwarp(input, flow, backwarp_grid=None):
    """skip_autoencoder.py:126-128 (the grid argument is implied by the input size)."""
    return ops.backwarp(input, flow, 1.0)


def _check_variants(opt):
    for flag in ("use_masked_flow", "use_deformed_conv", "use_tradeoff", "no_corr"):
        if getattr(opt, flag, False):
            raise NotImplementedError(f"--q_{flag} is outside the MI355X hot path")


class Matching(nn.Module):
    """Coarse flow / occlusion estimate from the 7x7 cost volume (skip_autoencoder.py:131-206)."""

    def __init__(self, flow_mult, kernel, feat_size, use_corr, corr_stride, use_masked_flow, use_deformed_conv,
                 use_tradeoff, no_proj, first):
        super().__init__()
        if not use_corr or use_masked_flow or use_deformed_conv or use_tradeoff:
            raise NotImplementedError("Matching (HIP): only the correlation variant is supported")
        self.flow_mult, self.corr_stride, self.first = flow_mult, corr_stride, first
        self.proj = ConvLayer(feat_size, max(16, feat_size // 4), 1) if (feat_size > 16 and not no_proj) else None
        if first:
            self.upsample_flow = self.upsample_occ = None
        else:
            self.upsample_flow = nn.ConvTranspose2d(2, 2, kernel_size=4, stride=2, padding=1, bias=False, groups=2)
            self.upsample_occ = nn.ConvTranspose2d(1, 1, kernel_size=4, stride=2, padding=1, bias=False, groups=1)
        self.upsample_corr = None if corr_stride == 1 else nn.ConvTranspose2d(49, 49, kernel_size=4, stride=2, padding=1,
                                                                                 bias=False, groups=49)
        self.convs = nn.Sequential(ConvLayer(49, 128, 3), ConvLayer(128, 64, 3), ConvLayer(64, 32, 3))
        self.flow_head = ConvLayer(32, 2, kernel, activate=False)
        self.occ_head = ConvLayer(32, 1, kernel, activate=False)


class Subpixel(nn.Module):
    """Flow / occlusion refinement on [input, warped, flow, occ] (skip_autoencoder.py:209-228)."""

    def __init__(self, flow_mult, kernel, feat_size, use_tradeoff):
        super().__init__()
        if use_tradeoff:
            raise NotImplementedError
        self.flow_mult = flow_mult
        self.convs = nn.Sequential(ConvLayer(2 * feat_size + 2 + 1, 128, 3), ConvLayer(128, 64, 3), ConvLayer(64, 32, 3))
        self.flow_head = ConvLayer(32, 2, kernel, activate=False)
        self.occ_head = ConvLayer(32, 1, kernel, activate=False)


class _FusedHeads:
    """flow_head (2 ch) and occ_head (1 ch) read the same 32-ch feature with the same kernel
    size: run them as ONE 3-output conv that writes [flow | occ] side by side."""

    def __init__(self, flow_head, occ_head):
        self.fh, self.oh = flow_head.conv, occ_head.conv
        self._cache = None

    def packed(self):
        fw, ow = self.fh.weight, self.oh.weight
        key = (fw.data_ptr(), fw._version, ow.data_ptr(), ow._version, fw.device, ops.CONV_PRECISION)
        if self._cache is None or self._cache[0] != key:
            b = torch.cat([self.fh.bias.detach(), self.oh.bias.detach()], dim=0).contiguous()
            self._cache = (key, ops.pack_head_weights(fw, ow), b)
        return self._cache[1], self._cache[2]

    def __call__(self, feat, out, accumulate):
        w, b = self.packed()
        return ops.conv_heads(feat, w, b, out, accumulate)


class InterBlock(nn.Module):
    """Per-level flow module (skip_autoencoder.py:231-265): Matching -> Subpixel -> warp of the k
    context features -> confidence fusion -> occlusion blend, for all k contexts at once."""

    def __init__(self, opt, height, width, flow_mult, kernel, feat_size, corr_stride, first=False):
        super().__init__()
        _check_variants(opt)
        self.flow_mult, self.feat_size, self.corr_stride = flow_mult, feat_size, corr_stride
        self.height, self.width = int(height), int(width)
        self.matching = Matching(flow_mult, kernel, feat_size, True, corr_stride, False, False, False,
                                 getattr(opt, "no_proj", False), first=first)
        self.subpixel = Subpixel(flow_mult, kernel, feat_size, False)
        self._m_heads = _FusedHeads(self.matching.flow_head, self.matching.occ_head)
        self._s_heads = _FusedHeads(self.subpixel.flow_head, self.subpixel.occ_head)
        self._up_w = None

    def prepare_packed(self):
        """Build every kernel-ready weight form this block caches lazily (the fused heads, the projection of the fused warp, the
        split first Subpixel convolution, the stacked flow / occlusion up-sampling filter), on the current stream."""
        self._m_heads.packed()
        self._s_heads.packed()
        if self.matching.proj is not None and ops.FUSE_WARP_PROJ:
            self._proj_weight()
        if self.matching.upsample_flow is not None:
            self._upsample_fo_weight()
        self._sub0_split()

    def _upsample_fo_weight(self):
        m = self.matching
        a, b = m.upsample_flow.weight, m.upsample_occ.weight
        key = (a.data_ptr(), a._version, b.data_ptr(), b._version, a.device)
        if self._up_w is None or self._up_w[0] != key:
            self._up_w = (key, torch.cat([a.detach(), b.detach()], dim=0).contiguous())  # [3,1,4,4]
        return self._up_w[1]

    @torch.no_grad()
    def forward_fused(self, dec, inters, fo_prev=None):
        """dec: [N,s,H,W] channel-slice VIEW of the decoder feature, updated in place.
        inters: k context features [N,s,H,W].  fo_prev: [N*k,3,h,w] (flow | occ) of the coarser
        level or None.  Returns this level's [N*k,3,H,W] (flow | occ) view."""
        n, s, h, w = dec.shape
        k = len(inters)
        m = self.matching
        ctxs = [t.reshape(n, s, h, w) for t in inters]   # slot views of the context ring: read in place by the warp kernels
        # Subpixel input [dec | warped ctx | flow | occ] (skip_autoencoder.py:224): the `dec` block is the same
        # for the k contexts of a frame, so its share of the first Subpixel conv is computed ONCE per frame
        # (`pre`) and broadcast inside the conv epilogue; only [warped | flow | occ] is materialised per pair.
        p8 = ops.CONV_PRECISION == "bf16x3" and ops.CONV_P8   # conv -> conv intermediates stay in the kernel's packed split-bf16 form
        # ... and the back-warp writes the Subpixel input in that form too (`ops.backwarp_p8`: [warped | flow | occ | 0 x 5], s + 8
        # channels): the 99-channel convolution, the largest of the level, then stages by LDS-DMA like the layers behind it
        p8_warp = p8 and ops.P8_WARP and s % 8 == 0 and w % 4 == 0
        if p8_warp:
            sp_in = None
            fo = torch.empty(n * k, 3, h, w, dtype=torch.float32, device=dec.device)
        else:
            sp_in = torch.empty(n * k, s + 3, h, w, dtype=torch.float32, device=dec.device)
            fo = sp_in[:, s:]
        inter_w = None
        if fo_prev is not None:
            ops.dwconvT4x4s2(fo_prev, self._upsample_fo_weight(), out=fo)       # learned x2 of flow and occ
            proj_w = self._proj_weight() if (m.proj is not None and ops.FUSE_WARP_PROJ) else None
            if proj_w is None:
                inter_w = ops.backwarp(ctxs, fo[:, :2], self.flow_mult)
        else:  # coarsest level: the cost volume reads the contexts themselves
            proj_w = None
            inter_w = torch.stack(ctxs, dim=1).view(n * k, s, h, w)
        if m.proj is not None:
            pa = m.proj(dec)                                                     # input projected once per n, not k times
            if inter_w is None:   # warp and projection in one pass: the warped s-channel tensor is never written
                pb = ops.backwarp_proj(ctxs, fo[:, :2], self.flow_mult, proj_w[0], proj_w[1], m.proj.conv.bias, m.proj.conv.out_channel,
                                       act=m.proj.activate)
            else:
                pb = m.proj(inter_w)
        else:
            pa, pb = dec.contiguous(), inter_w
        corr = ops.correlation7x7(pa, pb, self.corr_stride, first_div=k, lrelu=True)
        if m.upsample_corr is not None:
            corr = ops.dwconvT4x4s2(corr, m.upsample_corr.weight.detach())
        feat = m.convs[2](m.convs[1](m.convs[0](corr, out_p8=p8), out_p8=p8), out_p8=p8)
        self._m_heads(feat, fo, accumulate=fo_prev is not None)
        del corr, feat, pa, pb, inter_w
        sp = self.subpixel
        w_dec, w_rest, w_rest8 = self._sub0_split()
        conv0 = sp.convs[0].conv
        pre = ops.conv2d(dec, w_dec, None, conv0.out_channel, 3, pad=1)          # [N,128,H,W], before dec is blended
        if p8_warp:
            sp_p8 = ops.backwarp_p8(ctxs, fo, self.flow_mult)
            feat = ops.conv2d(sp_p8, w_rest8, conv0.bias, conv0.out_channel, 3, pad=1, act=True, pre=pre, pre_div=k, out_p8=p8)
            del sp_p8
        else:
            ops.backwarp(ctxs, fo[:, :2], self.flow_mult, out=sp_in[:, :s])
            feat = ops.conv2d(sp_in, w_rest, conv0.bias, conv0.out_channel, 3, pad=1, act=True, pre=pre, pre_div=k, out_p8=p8)
        feat = sp.convs[2](sp.convs[1](feat, out_p8=p8), out_p8=p8)
        self._s_heads(feat, fo, accumulate=True)
        del feat, pre
        ops.warp_fuse_blend(dec, ctxs, fo[:, :2], fo[:, 2:3], self.flow_mult, k)
        return fo

    def _proj_weight(self):
        """(w_t, CoutPad) of Matching.proj for `ops.backwarp_proj`, or None when the fused kernel has no instantiation."""
        w = self.matching.proj.conv.weight
        key = (w.data_ptr(), w._version, w.device)
        if getattr(self, "_projw", None) is None or self._projw[0] != key:
            self._projw = (key, ops.pack_proj_weight(w))
        return self._projw[1]

    def _sub0_split(self):
        """First Subpixel conv split along its input channels: (dec block, [warped | flow | occ] block), both
        with the FULL fan-in scale 1/sqrt((2s+3)*9) of the unsplit layer."""
        conv = self.subpixel.convs[0].conv
        w = conv.weight
        key = (w.data_ptr(), w._version, w.device, ops.CONV_PRECISION, ops.P8_WARP)
        if getattr(self, "_sub0", None) is None or self._sub0[0] != key:
            s = self.feat_size
            w8 = None
            if ops.P8_WARP:   # third form, only for the packed back-warp (off by default): the same block with five zero channels behind [warped | flow | occ]
                w8 = ops.pack_conv_weight(torch.cat([w[:, s:].detach(), w.new_zeros(w.shape[0], 5, w.shape[2], w.shape[3])], dim=1), scale=conv.scale)
            self._sub0 = (key, ops.pack_conv_weight(w[:, :s], scale=conv.scale), ops.pack_conv_weight(w[:, s:], scale=conv.scale), w8)
        return self._sub0[1], self._sub0[2], self._sub0[3]

    def forward(self, input, inters, flows=None, occs=None, toffs=None, eps=1e-6):
        """Reference signature (skip_autoencoder.py:246): returns (fused input, flows, occs, toffs)."""
        fo_prev = torch.cat([flows, occs], dim=1) if flows is not None else None
        out = input.contiguous().clone()
        fo = self.forward_fused(out, inters, fo_prev)
        return out, fo[:, :2], fo[:, 2:3], None


class SkipGANEncoder(nn.Module):
    """skip_autoencoder.py:309-351."""

    def __init__(self, opt, blur_kernel=[1, 3, 3, 1], mode="rgb"):
        super().__init__()
        if mode != "rgb":
            raise NotImplementedError("SkipGANEncoder (HIP): rgb mode only (layout conditioning is outside the path)")
        self.normalize_out = bool(getattr(opt, "normalize_out", False))
        necf_mult, necf = opt.necf_mult, opt.necf
        self.num_resolutions = len(necf_mult)
        self.z_size, self.mode = opt.z_size, mode
        block_in = necf * necf_mult[0]
        blocks = [ConvLayer(3, block_in, 1)]
        inter_sizes = [int(opt.inter_p * block_in)]
        block_out = block_in
        for i in range(1, self.num_resolutions):
            block_out = necf * necf_mult[i]
            blocks.append(ResBlock(block_in, block_out, blur_kernel, downsample=True))
            inter_sizes.append(int(opt.inter_p * block_out))
            block_in = block_out
        blocks.append(ConvLayer(block_out, self.z_size, 1))
        self.blocks = nn.ModuleList(blocks)
        self.inter_sizes = inter_sizes

    @torch.no_grad()
    def forward(self, input):
        input, vid_size = flatten_vid(input)
        out = self.blocks[0](input)
        inter_enc = [out[:, :self.inter_sizes[0]]]
        for i in range(1, self.num_resolutions):
            out = self.blocks[i](out)
            inter_enc.append(out[:, :self.inter_sizes[i]])
        out = self.blocks[self.num_resolutions](out)
        if self.normalize_out:   # skip_autoencoder.py:348-349: out / torch.norm(out, p=2, dim=1, keepdim=True)
            out = ops.l2_normalize_channels_(out.contiguous())
        return unflatten_vid(out, vid_size), [unflatten_vid(feat, vid_size) for feat in inter_enc]


class SkipGANDecoder(nn.Module):
    """skip_autoencoder.py:354-476 (mode rgb, no skip_rgb)."""

    def __init__(self, opt, blur_kernel=[1, 3, 3, 1], mode="rgb"):
        super().__init__()
        if mode != "rgb" or getattr(opt, "skip_rgb", False):
            raise NotImplementedError("SkipGANDecoder (HIP): rgb mode without skip_rgb only")
        necf_mult, necf = opt.necf_mult, opt.necf   # (sic) the reference reads necf, not ndcf: skip_autoencoder.py:357-358
        self.num_resolutions = len(necf_mult)
        self.use_inter, self.z_size, self.skip_tanh, self.mode = opt.use_inter, opt.z_size, getattr(opt, "skip_tanh", False), mode
        block_in = necf * necf_mult[-1]
        blocks = [ConvLayer(opt.z_size, block_in, 1)]
        inter_sizes = [int(opt.inter_p * block_in)]
        block_out = block_in
        for i in range(1, self.num_resolutions):
            block_out = necf * necf_mult[-1 - i]
            blocks.append(ResBlock(block_in, block_out, blur_kernel, upsample=True))
            inter_sizes.append(int(opt.inter_p * block_out))
            block_in = block_out
        blocks.append(ConvLayer(block_out, 3, 1, activate=False))
        self.blocks = nn.ModuleList(blocks)
        self.last_flow_mult = None
        if self.use_inter:
            inter_blocks = []
            height = opt.max_dim / (2 ** (self.num_resolutions - 1))
            width = int(height * opt.aspect_ratio)
            for i in range(self.num_resolutions):
                kernel = 2 ** (i // 2 + 1) + 1
                flow_mult = 2 ** i
                corr_stride = 2 if i > 2 else 1
                inter_blocks.append(InterBlock(opt, height, width, flow_mult, kernel, inter_sizes[i], corr_stride, first=i == 0))
                height *= 2
                width *= 2
            self.last_flow_mult = flow_mult
            self.inter_blocks = nn.ModuleList(inter_blocks)
            self.inter_sizes = inter_sizes

    def backwarp_img(self, input, flow):
        return ops.backwarp(input, flow, 1.0)

    @torch.no_grad()
    def forward(self, input, inter_tgts=None, return_all=False, drop_p=0, inter_src=None, alpha_src=None,
                inter_pre_warping=True, has_ctx=True):
        if drop_p > 0 or inter_src is not None:
            raise NotImplementedError("drop_p / inter_src are training-time options")
        input, vid_size = flatten_vid(input)
        use_inter = inter_tgts is not None and self.use_inter and has_ctx
        if use_inter:
            inter_tgts = [[flatten_vid(t)[0] for t in inter_tgt] for inter_tgt in inter_tgts]
        out = self.blocks[0](input)
        inter_flows, inter_occs, inter_dec = [], [], []
        fo = None
        for i in range(self.num_resolutions):
            if i > 0:
                out = self.blocks[i](out)
            if use_inter:
                s = self.inter_sizes[i]
                if inter_pre_warping and return_all:
                    inter_dec.append(out[:, :s].clone())
                fo = self.inter_blocks[i].forward_fused(out[:, :s], [tgt[-1 - i] for tgt in inter_tgts], fo)
                if return_all:
                    if not inter_pre_warping:
                        inter_dec.append(out[:, :s])
                    inter_flows.append(fo[:, :2])
                    inter_occs.append(fo[:, 2:3])
        out1 = self.blocks[self.num_resolutions](out)
        if self.skip_tanh:
            out1 = torch.tanh(out1)
        out1 = unflatten_vid(out1, vid_size)
        out2 = torch.tensor([])
        if return_all:
            return out1, out2, inter_flows, inter_occs, [unflatten_vid(f, vid_size) for f in inter_dec]
        return out1, out2


def prepare_packed_modules(root):
    """Fill the lazily built weight caches of every sub-module of `root` on the current stream: `packed()` of the convolutions /
    linears, `prepare_packed()` of the InterBlocks, the transposed codebook of a VectorQuantizer (`QVidModel.prepare_packed`)."""
    for mod in root.modules():
        if mod is root:
            continue
        if hasattr(mod, "prepare_packed"):
            mod.prepare_packed()
        elif hasattr(mod, "packed"):
            mod.packed()
        elif hasattr(mod, "_tables"):
            mod._tables()


class EqualLinear(nn.Module):
    """skip_autoencoder.py:478-507 (no activation): y = x @ (W * scale)^T + bias * lr_mul on the nn.Linear GEMM kernel."""

    def __init__(self, in_dim, out_dim, bias=True, bias_init=0, lr_mul=1, activation=None):
        super().__init__()
        if activation:
            raise NotImplementedError("EqualLinear (HIP): fused_leaky_relu activation is not on the hot path")
        self.weight = nn.Parameter(torch.randn(out_dim, in_dim).div_(lr_mul))
        self.bias = nn.Parameter(torch.zeros(out_dim).fill_(bias_init)) if bias else None
        self.activation = activation
        self.scale = (1 / math.sqrt(in_dim)) * lr_mul
        self.lr_mul = lr_mul
        self._packed = None

    def packed(self):
        src = (self.weight,) + ((self.bias,) if self.bias is not None else ())
        key = tuple((t.data_ptr(), t._version) for t in src) + (self.weight.device,)
        if self._packed is None or self._packed[0] != key:
            w = self.weight.detach() * self.scale
            w = torch.nn.functional.pad(w, (0, (-w.shape[1]) % 16)).contiguous()   # the GEMM kernel walks K in steps of 16
            b = (self.bias.detach() * self.lr_mul).contiguous() if self.bias is not None else None
            self._packed = (key, w, b)
        return self._packed[1], self._packed[2]

    def forward(self, input):
        w, b = self.packed()
        x = input.reshape(-1, input.shape[-1])
        x = torch.nn.functional.pad(x, (0, w.shape[1] - x.shape[1])).contiguous()
        return ops.gemm_nt(x, w, b).view(*input.shape[:-1], w.shape[0])

    def __repr__(self):
        return f"{self.__class__.__name__}({self.weight.shape[1]}, {self.weight.shape[0]})"


class StateEstimator(nn.Module):
    """skip_autoencoder.py:510-528: quantised latent map [z_size, h, w] -> state in (0, 1)^state_size: blur + stride-2
    3x3 ConvLayers down to 1 x 1, EqualLinear, sigmoid."""

    def __init__(self, opt):
        super().__init__()
        convs = []
        h, w = opt.z_shape
        in_size = opt.z_size
        while h > 1 and w > 1:
            convs.append(ConvLayer(in_size, opt.state_hsize, 3, downsample=True))
            h //= 2
            w //= 2
            in_size = opt.state_hsize
        self.convs = nn.Sequential(*convs)
        self.fc = EqualLinear(opt.state_hsize * h * w, opt.state_size)

    @torch.no_grad()
    def forward(self, input):
        x, vid_size = flatten_vid(input)
        for conv in self.convs:
            x = conv(x.contiguous())
        out = torch.sigmoid(self.fc(x.reshape(x.size(0), -1)))
        return unflatten_vid(out, vid_size)


class StftEncoder(nn.Module):
    """skip_autoencoder.py:530-542: spectrogram frame [1,H,W] -> [stft_size, H/8, W/8]: 1x1 conv, three blur + stride-2
    3x3 convs, one 3x3 conv (all ConvLayers with bias + LeakyReLU(0.1))."""

    def __init__(self, opt):
        super().__init__()
        convs = [ConvLayer(1, opt.stft_hsize, 1, downsample=False)]
        for _ in range(3):
            convs.append(ConvLayer(opt.stft_hsize, opt.stft_hsize, 3, downsample=True))
        convs.append(ConvLayer(opt.stft_hsize, opt.stft_size, 3, downsample=False))
        self.convs = nn.Sequential(*convs)

    def forward(self, input):
        x, vid_size = flatten_vid(input)
        for conv in self.convs:
            x = conv(x.contiguous())
        return unflatten_vid(x, vid_size)
