#!/usr/bin/env python3
"""BAIR-size models, greedy sampling: clip i generated inside a batch of 5 equals the same clip generated alone, bit for bit
(ragged batch sizes through every kernel: GEMM row tails, attention grids, conv batch strides, context lists)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ccvs_amd.tools.options import Options, BAIR_ARGV  # noqa: E402
from ccvs_amd.helpers.generator import Generator  # noqa: E402

opt = Options().parse(load_qvid_generator=True, load_transformer=True, argv=list(BAIR_ARGV) + ["--batch_size_vid", "5"])
xopt, qopt = opt["transformer"], opt["qvid_generator"]
xopt.vid_len = qopt.vid_len = 4
xopt.sample = False
torch.manual_seed(0)
gen = Generator(opt).build_models()
with torch.no_grad():
    t = gen.transformer_model.net_t
    t.s_emb.normal_(0, 0.02)
    t.t_emb.normal_(0, 0.02)
    vid = gen.synthetic_batch(5, seed=3)["vid"][:, :4].cuda()
    z_e, _ = gen.vid_model.net_e(vid[:2, :1])
    cb = gen.vid_model.net_q.embedding.weight
    cb.copy_(torch.randn_like(cb) * float(z_e.std()))
    full = gen.generate_vid({"vid": vid.clone()})
    for i in (0, 3):
        one = gen.generate_vid({"vid": vid[i:i + 1].clone()})
        same_code = torch.equal(one["fake"]["code"][0], full["fake"]["code"][i])
        diff = (one["fake"]["vid"][0] - full["fake"]["vid"][i]).abs().max().item()
        print(f"clip {i}: tokens equal {same_code}, max |pixel diff| {diff:.3e}")
        assert same_code and diff == 0.0
print("ok")
