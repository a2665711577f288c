"""VectorQuantizer with the reference's interface (modules/quantize.py:7-83): inference
only -- nearest-codebook indices, quantised features, `embed_code`."""
import torch
import torch.nn as nn

from ccvs_amd import ops


class VectorQuantizer(nn.Module):
    def __init__(self, n_e, e_dim, beta, mult=1, normalize=False):
        super().__init__()
        assert e_dim % mult == 0
        if mult != 1 or normalize:
            raise NotImplementedError("VectorQuantizer (HIP): mult > 1 / normalize are outside the hot path")
        self.n_e, self.e_dim, self.beta, self.mult, self.normalize = n_e, e_dim, beta, mult, normalize
        self.embedding = nn.Embedding(n_e, e_dim)
        if e_dim <= 1:
            self.embedding.weight.data.uniform_(0, 1.0)               # quantize.py:28
        else:
            self.embedding.weight.data.uniform_(-1.0 / n_e, 1.0 / n_e)  # quantize.py:30
        self._packed = None

    def _tables(self):
        w = self.embedding.weight
        key = (w.data_ptr(), w._version, w.device)
        if self._packed is None or self._packed[0] != key:
            wd = w.detach()
            cbt, esq = wd.t().contiguous(), (wd ** 2).sum(dim=1).contiguous()
            pad = (-wd.shape[0]) % 32  # the argmin kernel streams the codebook in blocks of 32 codes
            if pad:  # padding codes sit at distance +inf: never the nearest
                cbt = torch.cat([cbt, torch.zeros(cbt.shape[0], pad, dtype=cbt.dtype, device=cbt.device)], dim=1).contiguous()
                esq = torch.cat([esq, torch.full((pad,), float("inf"), dtype=esq.dtype, device=esq.device)]).contiguous()
            self._packed = (key, cbt, esq)
        return self._packed[1], self._packed[2]

    def _as_nchw(self, z):
        """quantize.py:41-45: a [b, (t,) c, h, w] map is quantised per position; anything with fewer than 4 dims is a flat
        list of e_dim-vectors (the state stream: [b, t, state_size] with e_dim = 1)."""
        if z.ndim >= 4:
            return z.reshape(-1, *z.shape[-3:])
        return z.reshape(-1, self.e_dim, 1, 1)

    @torch.no_grad()
    def indices(self, z):
        """z [..., C, H, W] -> int64 [N*H*W] in (n, h, w) raster order (quantize.py:40-50)."""
        cbt, esq = self._tables()
        return ops.vq_argmin(self._as_nchw(z), cbt, esq)

    @torch.no_grad()
    def forward(self, z):
        """Returns (z_q, None, (None, None, indices[N,1])) -- the training-only loss /
        perplexity / one-hot outputs of quantize.py:51-68 are not produced."""
        idx = self.indices(z)
        z4 = self._as_nchw(z)
        hw = z4.shape[2] * z4.shape[3]
        zq = ops.embed_gather(idx, self.embedding.weight.detach(), z4.shape[0], hw).view(z.shape)
        return zq, None, (None, None, idx.unsqueeze(1))

    @torch.no_grad()
    def embed_code(self, code):
        """[..., h, w] int64 -> [..., h, w, C] (quantize.py:76-83); any other shape s -> [*s, C]."""
        w = self.embedding.weight.detach()
        if code.ndim < 3:
            return ops.embed_gather(code.reshape(-1), w, code.numel(), 1).view(*code.shape, w.shape[1])
        n = code.numel() // (code.shape[-1] * code.shape[-2])
        hw = code.shape[-1] * code.shape[-2]
        z = ops.embed_gather(code.reshape(-1), w, n, hw)  # [n, C, hw]
        return z.transpose(1, 2).reshape(*code.shape, w.shape[1])

    @torch.no_grad()
    def embed_code_nchw(self, code, n, h, w):
        """Same gather laid out [n, C, h, w] directly (what QVidModel.decode wants after its
        two transposes, quantized_video_model.py:832-833)."""
        return ops.embed_gather(code.reshape(-1), self.embedding.weight.detach(), n, h * w).view(n, -1, h, w)
