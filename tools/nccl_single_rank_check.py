#!/usr/bin/env python3
"""One-rank RCCL process group around a generation step: the RCCL watchdog thread is alive while the decode step is
captured into a hipGraph, and the uint8 clips go through all_gather_into_tensor -- the N-GPU code path of bench.py
exercised on the one GPU a test box has."""
import os
import sys

import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29533")
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))  # as Engine does
from ccvs_amd import ops  # noqa: E402
from ccvs_amd.tools.options import Options, BAIR_ARGV  # noqa: E402
from ccvs_amd.helpers.generator import Generator  # noqa: E402

opt = Options().parse(load_qvid_generator=True, load_transformer=True, argv=list(BAIR_ARGV) + ["--batch_size_vid", "2", "--x_sample_noise", "device"])
opt["transformer"].vid_len = 3
opt["qvid_generator"].vid_len = 3
torch.manual_seed(0)
gen = Generator(opt).build_models()
data = {"vid": gen.synthetic_batch(2, seed=1)["vid"][:, :3].cuda()}
for _ in range(2):
    out = gen.generate_vid(data)
    packed = ops.pack_u8(out["fake"]["vid"]).contiguous()
    gathered = torch.empty_like(packed)
    dist.all_gather_into_tensor(gathered, packed)
    dist.barrier()
torch.cuda.synchronize()
assert torch.equal(gathered, packed)
print("ok: hipGraph capture + RCCL all-gather with the process group alive", tuple(gathered.shape))
dist.destroy_process_group()
