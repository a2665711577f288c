#!/usr/bin/env python3
"""Encode + flow-guided decode of one BAIR batch with RANDOM tokens in place of the transformer's
(same convolution launches as one bench.py step, no hipGraph) -- the target of rocprofv3 --pmc runs,
which cannot follow graph replays.   python tools/decode_only.py [batch]"""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ccvs_amd.tools.options import Options, BAIR_ARGV  # noqa: E402
from ccvs_amd.helpers.generator import Generator  # noqa: E402
from ccvs_amd import ops  # noqa: E402

batch = int(sys.argv[1]) if len(sys.argv) > 1 else 16
opt = Options().parse(load_qvid_generator=True, load_transformer=True, argv=list(BAIR_ARGV) + ["--batch_size_vid", str(batch), "--rec_only"])
torch.manual_seed(0)
gen = Generator(opt).build_models()
data = {"vid": gen.synthetic_batch(batch, seed=1)["vid"].cuda()}
# CCVS_DUMP_CONV_BYTES=path: the algorithmic bytes of every convolution launch, in launch order (main operand + output, and the
# side operands: weights, residual, pre-activation addend, accumulated output) -- tools/pmc_conv_traffic_reduce.py matches them with
# the counter rows of the same launches (same order) for the per-instantiation over-fetch table
dump = os.environ.get("CCVS_DUMP_CONV_BYTES")
if dump:
    ops.KERNEL_TIMER = ops.KernelTimer()
with torch.no_grad():
    enc = gen.vid_model(data, mode="vid_encoder")
    code = torch.randint(0, 1024, (batch, 1024), generator=torch.Generator().manual_seed(2)).cuda()
    code[:, :64] = enc["code"][:, :64]
    inter = [f[:, :1].contiguous() for f in enc["inter"]]
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    out = gen.vid_model({"code": code, "inter": inter}, mode="vid_decoder")
    e1.record()
    torch.cuda.synchronize()
if dump:
    recs = [{"bytes": r[4], "side_bytes": r[5], "flops": r[1], "tag": r[6]} for r in ops.KERNEL_TIMER.records]
    ops.KERNEL_TIMER = None
    json.dump(recs, open(dump, "w"))
print(f"decode of {batch} clips: {e0.elapsed_time(e1):.1f} ms, clip {tuple(out['vid'].shape)}")
