#!/usr/bin/env python3
"""Soak of Generator.run_pipelined at the benchmark configuration (BAIR 256x256, batch 16, default token groups / chains) with
an RCCL process group alive (one rank: the watchdog thread, the side-stream all-gather): RUNS runs of BATCHES batches each.
Every wait inside run_pipelined has a time limit and raises after dumping all thread stacks; on top of that a
faulthandler watchdog ends the process with the stacks if a run exceeds 4 x its expected time.  Prints one line per run
(batches, seconds, frames/s, slowest inter-batch gap on the decode stream) and a total.
    python tools/soak_pipelined.py [RUNS] [BATCHES] [port]"""
import faulthandler
import os
import sys
import time

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
runs = int(sys.argv[1]) if len(sys.argv) > 1 else 5
n_batches = int(sys.argv[2]) if len(sys.argv) > 2 else 200
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ["MASTER_PORT"] = sys.argv[3] if len(sys.argv) > 3 else os.environ.get("MASTER_PORT", "29541")
os.environ.update(RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
from ccvs_amd import ops  # noqa: E402
from ccvs_amd.tools.engine import Engine  # noqa: E402
from ccvs_amd.tools.options import Options, BAIR_ARGV  # noqa: E402
from ccvs_amd.helpers.generator import Generator  # noqa: E402

batch = 16
opt = Options().parse(load_qvid_generator=True, load_transformer=True,
                      argv=list(BAIR_ARGV) + ["--batch_size_vid", str(batch), "--x_sample_noise", os.environ.get("CCVS_SOAK_NOISE", "host"), "--rec_pass", "false"])   # host: the headline's sampler since round 5
torch.manual_seed(0)
engine = Engine()
engine.distributed = True          # one rank, but the collective path
gen = Generator(opt).build_models()
gen.engine = engine
pool = [{"vid": gen.synthetic_batch(batch, seed=1 + i)["vid"].cuda()} for i in range(4)]   # inputs recycled: resident, not regenerated


def finish(i, fake):
    return engine.all_gather_clips_async(ops.pack_u8(fake["vid"]))


def batches(n):
    for i in range(n):
        yield {"vid": pool[i % len(pool)]["vid"]}


gen.run_pipelined(batches(4), first_iter=0, finish=finish)   # warm-up: graph captures
torch.cuda.synchronize()
total_b, total_s = 0, 0.0
for r in range(runs):
    faulthandler.dump_traceback_later(int(4 * 1.6 * n_batches + 120), exit=True)
    t0 = time.perf_counter()
    res = gen.run_pipelined(batches(n_batches), first_iter=1000 * (r + 1), finish=finish)
    for x in res:
        x["finished"].wait()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    faulthandler.cancel_dump_traceback_later()
    tl = gen.pipeline_timeline()
    ends = sorted(t["d1"] for t in tl)      # (the decodes of consecutive batches overlap: frame by frame, two decode streams)
    gaps = [b - a for a, b in zip(ends[:-1], ends[1:])]
    assert len(res) == n_batches and [x["index"] for x in res] == list(range(1000 * (r + 1), 1000 * (r + 1) + n_batches))
    total_b += n_batches
    total_s += dt
    print(f"run {r + 1}/{runs}: {n_batches} batches in {dt:.1f} s = {15 * batch * n_batches / dt:.1f} frames/s; longest time between two finished "
          f"batches {max(gaps):.0f} ms (median {sorted(gaps)[len(gaps) // 2]:.0f} ms); load average {os.getloadavg()[0]:.1f}", flush=True)
print(f"soak: {total_b} batches, {total_s:.0f} s, {15 * batch * total_b / total_s:.1f} frames/s, no stall, no time limit hit "
      f"(lanes {gen.last_lanes}, chains {gen.last_chains}, decode streams {gen.last_dec_streams}, noise {os.environ.get('CCVS_SOAK_NOISE', 'host')}; "
      f"host RSS {__import__('psutil').Process().memory_info().rss / 2 ** 30:.1f} GB)", flush=True)
dist.destroy_process_group()
