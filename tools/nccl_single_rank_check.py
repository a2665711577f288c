#!/usr/bin/env python3
"""One-rank RCCL process group around the generation schedules: the RCCL watchdog thread is alive while the decode step is
captured into a hipGraph, and the uint8 clips go through `Engine.all_gather_clips_async` (all_gather_into_tensor on a side
stream) -- the N-GPU code path of bench.py, serial and pipelined, exercised on the one GPU a test box has.
usage: python tools/nccl_single_rank_check.py [port]"""
import os
import sys

import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ["MASTER_PORT"] = sys.argv[1] if len(sys.argv) > 1 else os.environ.get("MASTER_PORT", "29533")
os.environ.update(RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))  # as Engine does
from ccvs_amd import ops  # noqa: E402
from ccvs_amd.tools.engine import Engine  # noqa: E402
from ccvs_amd.tools.options import Options, BAIR_ARGV  # noqa: E402
from ccvs_amd.helpers.generator import Generator  # noqa: E402

opt = Options().parse(load_qvid_generator=True, load_transformer=True,
                      argv=list(BAIR_ARGV) + ["--batch_size_vid", "2", "--x_sample_noise", "device", "--rec_pass", "false"])
opt["transformer"].vid_len = 3
opt["qvid_generator"].vid_len = 3
torch.manual_seed(0)
engine = Engine()
engine.distributed = True          # one rank, but take the collective path (the group above is what Engine would have built)
gen = Generator(opt).build_models()
gen.engine = engine
batches = [{"vid": gen.synthetic_batch(2, seed=1 + i)["vid"][:, :3].cuda()} for i in range(3)]


def finish(i, fake):
    return engine.all_gather_clips_async(ops.pack_u8(fake["vid"]))


serial = []
for i, b in enumerate(batches):
    out = gen.generate_vid({"vid": b["vid"].clone()}, global_iter=i)
    serial.append(finish(i, out["fake"]).wait().clone())
    dist.barrier()
res = gen.run_pipelined(({"vid": b["vid"].clone()} for b in batches), first_iter=0, cu_limit=192, finish=finish)
piped = [r["finished"].wait() for r in res]
torch.cuda.synchronize()
for a, b in zip(serial, piped):
    assert a.dtype == torch.uint8 and torch.equal(a, b)
print("ok: hipGraph capture + RCCL all-gather on a side stream, serial and pipelined schedules agree", tuple(piped[0].shape))
dist.destroy_process_group()
