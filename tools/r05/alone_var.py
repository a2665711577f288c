#!/usr/bin/env python3
"""Why does the convolutions' "alone" figure of bench.py vary between runs of one box (183 ... 266 TFLOP/s)?  generate_vid of one
BAIR batch several times under the launch timer; per pass: TFLOP/s of the convolutions of the encode and of the decode, and the
launch-by-launch ratio against the fastest pass.   python tools/r05/alone_var.py [passes]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from ccvs_amd import ops  # noqa: E402
from ccvs_amd.tools.options import Options, BAIR_ARGV  # noqa: E402
from ccvs_amd.helpers.generator import Generator  # noqa: E402
import bench  # noqa: E402

passes = int(sys.argv[1]) if len(sys.argv) > 1 else 4
opt = Options().parse(load_qvid_generator=True, load_transformer=True, argv=list(BAIR_ARGV) + ["--batch_size_vid", "16", "--x_sample_noise", "host", "--rec_pass", "false"])
torch.manual_seed(0)
gen = Generator(opt).build_models()
data = {"vid": gen.synthetic_batch(16, seed=1)["vid"].cuda()}
bench.calibrate_codebook(gen, {"vid": gen.synthetic_batch(2, seed=1, first_clip=0)["vid"].cuda()})
gen.generate_vid({k: v.clone() for k, v in data.items()}, 0)
torch.cuda.synchronize()
runs = []
for p in range(passes):
    if p == passes - 1:
        # a pass with the chip kept busy in front of the decode?  no: the last pass follows 5 s of idle
        time.sleep(5.0)
    t = ops.KernelTimer()
    ops.KERNEL_TIMER = t
    torch.manual_seed(1)
    t0 = time.perf_counter()
    gen.generate_vid({k: v.clone() for k, v in data.items()}, 0)
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    ops.KERNEL_TIMER = None
    rec = [(r[1], r[2].elapsed_time(r[3])) for r in t.records if r[0].startswith("conv2d_")]
    runs.append(rec)
    n_enc = 0
    # the encode's launches come first: up to the first launch after the token loop (a gap in the event times cannot be seen here;
    # split by count: the decode's share is the last 15 frames' launches) -- report the thirds instead
    k = len(rec)
    parts = [rec[: k // 3], rec[k // 3: 2 * k // 3], rec[2 * k // 3:]]
    tf = lambda rr: sum(f for f, _ in rr) / (sum(m for _, m in rr) * 1e-3) / 1e12
    print(f"pass {p}: wall {wall:.2f} s, {k} conv launches, {tf(rec):.1f} TFLOP/s (thirds: " + ", ".join(f"{tf(x):.1f}" for x in parts) + ")", flush=True)
best = min(range(passes), key=lambda i: sum(m for _, m in runs[i]))
for p in range(passes):
    ratios = sorted(runs[p][i][1] / max(runs[best][i][1], 1e-6) for i in range(len(runs[p])))
    big = [(i, runs[p][i][1], runs[best][i][1]) for i in range(len(runs[p])) if runs[p][i][1] > 2 * runs[best][i][1] and runs[p][i][1] > 0.2]
    print(f"pass {p} / pass {best}: launch-time ratio median {ratios[len(ratios) // 2]:.3f}, p90 {ratios[int(0.9 * len(ratios))]:.3f}, max {ratios[-1]:.2f}; "
          f"{len(big)} launches over 2x and 0.2 ms: " + " ".join(f"#{i}:{a:.2f}/{b:.2f}ms" for i, a, b in big[:12]))
