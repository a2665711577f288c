#!/usr/bin/env python3
"""Which convolution shapes does one BAIR batch spend its time in?  Decodes one batch (random tokens) with a HIP-event
bracket around every convolution launch and prints the shapes by total time: launches, ms, algorithmic TFLOP/s and GB/s
(inputs read once + outputs written once).   python tools/conv_shape_census.py [batch]"""
import os
import sys
from collections import defaultdict

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ccvs_amd import ops  # noqa: E402
from ccvs_amd.tools.options import Options, BAIR_ARGV  # noqa: E402
from ccvs_amd.helpers.generator import Generator  # noqa: E402

batch = int(sys.argv[1]) if len(sys.argv) > 1 else 16
opt = Options().parse(load_qvid_generator=True, load_transformer=True,
                      argv=list(BAIR_ARGV) + ["--batch_size_vid", str(batch), "--x_sample_noise", "device", "--rec_pass", "false"])
torch.manual_seed(0)
gen = Generator(opt).build_models()
data = {"vid": gen.synthetic_batch(batch, seed=1)["vid"].cuda()}


class Census(ops.KernelTimer):
    def begin(self, name, flops=0.0, nbytes=0.0, side=0.0, tag=None):
        super().begin(name, flops, nbytes, side, tag)
        self._open = (self._key, flops, self._open[2], nbytes, side, tag)


census = Census()
orig_conv2d = ops.conv2d


def conv2d(x, w_packed, bias, cout, k, stride=1, pad=0, transposed=False, **kw):
    xs = tuple(x.shape)   # (P8Act.shape is the logical [n, c, h, w])
    census._key = (xs, cout, k, stride, bool(transposed), getattr(w_packed, "kw", k))
    return orig_conv2d(x, w_packed, bias, cout, k, stride, pad, transposed, **kw)


ops.conv2d = conv2d
with torch.no_grad():
    enc = gen.vid_model(data, mode="vid_encoder")
    code = torch.randint(0, 1024, (batch, 1024), generator=torch.Generator().manual_seed(2)).cuda()
    code[:, :64] = enc["code"][:, :64]
    inter = [f[:, :1].contiguous() for f in enc["inter"]]
    gen.vid_model({"code": code, "inter": inter}, mode="vid_decoder")     # warm
    torch.cuda.synchronize()
    ops.KERNEL_TIMER = census
    gen.vid_model({"code": code, "inter": inter}, mode="vid_decoder")
    torch.cuda.synchronize()
    ops.KERNEL_TIMER = None
agg = defaultdict(lambda: [0, 0.0, 0.0, 0.0, 0])
for key, flops, e0, e1, nbytes, _side, _tag in census.records:
    a = agg[(key[0][1:],) + key[1:]]          # launches that differ only in the number of images are folded together
    a[0] += 1
    a[1] += e0.elapsed_time(e1)
    a[2] += flops
    a[3] += nbytes
    a[4] = max(a[4], key[0][0])
total = sum(a[1] for a in agg.values())
print(f"{len(census.records)} convolution launches, {total:.0f} ms (HIP events around each launch)")
for key, (n, ms, fl, nb, nmax) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:60]:
    print(f"{ms:7.1f} ms {100 * ms / total:5.1f}%  x{n:4d}  in [<={nmax:3d}]x{str(key[0]):16s} cout {key[1]:4d} k{key[2]}x{key[5]} s{key[3]} "
          f"{'T' if key[4] else ' '}  {fl / ms / 1e9:6.1f} TF/s  {nb / ms / 1e6:6.0f} GB/s")
