"""The three tensor helpers of the reference's tools/utils.py that sit on the hot path
(tools/utils.py:40-62), plus the small host helpers helpers/generator.py imports."""
import os

import torch


def _device():
    if not torch.cuda.is_available():
        raise RuntimeError("ccvs_amd needs a GPU: the synthesis path runs on HIP kernels only (no CPU fallback)")
    return torch.device("cuda", torch.cuda.current_device())


def to_cuda(tensor_dic, key, flatten_empty=True):
    """tools/utils.py:40-48: move `tensor_dic[key]` (tensor or list of tensors) to the GPU;
    a missing key yields an empty tensor."""
    dev = _device()
    if key in tensor_dic:
        if isinstance(tensor_dic[key], list):
            return [t.to(dev) for t in tensor_dic[key]]
        if 0 in tensor_dic[key].size() and flatten_empty:
            tensor_dic[key] = torch.Tensor([])
        return tensor_dic[key].to(dev)
    return torch.Tensor([]).to(dev)


def flatten_vid(x, vid_ndim=5):
    """tools/utils.py:50-55."""
    vid_size = None
    if x.ndim == vid_ndim:
        vid_size = x.shape[:2]
        x = x.reshape(-1, *x.shape[2:])
    return x, vid_size


def unflatten_vid(x, vid_size):
    """tools/utils.py:57-62."""
    if vid_size is not None and x.size(0) != 0:
        b, t = vid_size
        return x.view(b, t, *x.shape[1:])
    return x


def mkdir(path):
    """tools/utils.py:19-21."""
    os.makedirs(path, exist_ok=True)


def mkdirs(paths):
    """tools/utils.py:12-17."""
    for path in ([paths] if isinstance(paths, str) else list(paths)):
        mkdir(path)


class DummyOpt:
    """tools/utils.py:128-136: a no-op optimiser stand-in."""

    def zero_grad(self):
        pass

    def step(self):
        pass


def color_transfer(im, colormap):
    """tools/utils.py:138-147 colours a semantic layout: layouts are outside the hot path."""
    raise NotImplementedError("layout colour maps are outside the MI355X hot path")

