#!/usr/bin/env python3
"""How far do the warps of the random-init BAIR decoder reach?  |flow * flow_mult| per pyramid level over one decode (decode_only's
set-up): decides whether an LDS-tiled gather (bounded displacement) would ever run in the bench."""
import os
import sys
import collections

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from ccvs_amd import ops  # noqa: E402
from ccvs_amd.tools.options import Options, BAIR_ARGV  # noqa: E402
from ccvs_amd.helpers.generator import Generator  # noqa: E402

batch = 4
opt = Options().parse(load_qvid_generator=True, load_transformer=True, argv=list(BAIR_ARGV) + ["--batch_size_vid", str(batch), "--rec_only"])
torch.manual_seed(0)
gen = Generator(opt).build_models()
stats = collections.defaultdict(list)
real_bw, real_fb = ops.backwarp, ops.warp_fuse_blend


def bw(x, flow, flow_mult=1.0, out=None):
    f = (flow[:, :2].float() * flow_mult).abs()
    stats[("backwarp", flow.shape[-1])].append((f.max().item(), f.mean().item(), torch.quantile(f.flatten()[::97], 0.99).item()))
    return real_bw(x, flow, flow_mult, out=out)


def fb(dec, ctx, flows, occs, flow_mult, k):
    f = (flows.float() * flow_mult).abs()
    stats[("fuse_blend", flows.shape[-1])].append((f.max().item(), f.mean().item(), torch.quantile(f.flatten()[::97], 0.99).item()))
    return real_fb(dec, ctx, flows, occs, flow_mult, k)


ops.backwarp, ops.warp_fuse_blend = bw, fb
data = {"vid": gen.synthetic_batch(batch, seed=1)["vid"].cuda()}
with torch.no_grad():
    enc = gen.vid_model(data, mode="vid_encoder")
    code = torch.randint(0, 1024, (batch, 1024), generator=torch.Generator().manual_seed(2)).cuda()
    code[:, :64] = enc["code"][:, :64]
    inter = [f[:, :1].contiguous() for f in enc["inter"]]
    gen.vid_model({"code": code, "inter": inter}, mode="vid_decoder")
torch.cuda.synchronize()
for (name, w), v in sorted(stats.items()):
    mx = max(a for a, _, _ in v); mean = sum(b for _, b, _ in v) / len(v); q = max(c for _, _, c in v)
    print(f"{name:12s} width {w:4d}: {len(v):3d} calls, |flow| max {mx:8.2f} px, mean {mean:7.3f}, 99th percentile (worst call) {q:7.2f}")
