"""`upfirdn2d(input, kernel, up, down, pad)` with the reference's signature
(modules/upfirdn2d.py:145-159), forward only, on the HIP FIR kernel.

The hot path only ever passes the separable kernel outer([1,3,3,1])/64 * gain
(skip_autoencoder.py:19-37); that structure is what the HIP kernel implements, so any other
kernel is rejected loudly rather than silently approximated.
"""
import torch

from ccvs_amd import ops

_BASE = torch.tensor([1.0, 3.0, 3.0, 1.0])
_BASE2D = (_BASE[None, :] * _BASE[:, None]) / 64.0


def kernel_gain(kernel):
    """Gain g such that kernel == outer([1,3,3,1])/64 * g; raises otherwise."""
    k = kernel.detach().float().cpu()
    if k.shape != (4, 4):
        raise ValueError("upfirdn2d (HIP): only the 4x4 [1,3,3,1] FIR kernel is supported")
    g = float(k.sum())
    if not torch.allclose(k, _BASE2D * g, rtol=1e-6, atol=1e-7):
        raise ValueError("upfirdn2d (HIP): kernel is not a multiple of outer([1,3,3,1])")
    return g


def upfirdn2d(input, kernel, up=1, down=1, pad=(0, 0)):
    return ops.upfirdn2d(input, up=up, down=down, pad=tuple(pad), gain=kernel_gain(kernel))
