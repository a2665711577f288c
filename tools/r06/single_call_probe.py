#!/usr/bin/env python3
"""The single-call drop-in (`Generator.generate_vid`, one batch, nothing else in flight): today the token loop (960 steps) runs first
and the decoder afterwards; the pipelined machinery can let the decoder follow the token loop frame by frame inside ONE batch as
well (`run_pipelined` over a one-batch list, lanes = chains = 1).  Times both on the BAIR configuration and checks that the clips are
the same bits.    python tools/r06/single_call_probe.py [batches]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from ccvs_amd.tools.options import Options, BAIR_ARGV  # noqa: E402
from ccvs_amd.helpers.generator import Generator  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 3
opt = Options().parse(load_qvid_generator=True, load_transformer=True,
                      argv=list(BAIR_ARGV) + ["--batch_size_vid", "16", "--x_sample_noise", "host", "--rec_pass", "false"])
torch.manual_seed(0)
gen = Generator(opt).build_models()
batches = [{"vid": gen.synthetic_batch(16, seed=1 + i)["vid"].cuda()} for i in range(n + 1)]
with torch.no_grad():
    torch.manual_seed(5)
    gen.generate_vid({k: v.clone() for k, v in batches[n].items()}, 100)            # warm (captures, packed weights)
    gen.run_pipelined([{k: v.clone() for k, v in batches[n].items()}], first_iter=100, lanes=1, chains=1)
    torch.cuda.synchronize()
    serial, streamed, ts, tp = [], [], [], []
    for i in range(n):
        torch.manual_seed(77 + i)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        out = gen.generate_vid({k: v.clone() for k, v in batches[i].items()}, i)
        torch.cuda.synchronize()
        ts.append(time.perf_counter() - t0)
        serial.append((out["fake"]["code"].clone(), out["fake"]["vid"].clone()))
        del out
    for i in range(n):
        torch.manual_seed(77 + i)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        res = gen.run_pipelined([{k: v.clone() for k, v in batches[i].items()}], first_iter=i, lanes=1, chains=1)
        torch.cuda.synchronize()
        tp.append(time.perf_counter() - t0)
        streamed.append((res[0]["fake"]["code"].clone(), res[0]["fake"]["vid"].clone()))
        del res
same = all(torch.equal(a[0], b[0]) and torch.equal(a[1], b[1]) for a, b in zip(serial, streamed))
print(f"generate_vid (token loop, then decode):      {[round(1e3 * t) for t in ts]} ms per call = {16 * 15 / min(ts):.1f} frames/s (best)")
print(f"one-batch run_pipelined (decoder follows):   {[round(1e3 * t) for t in tp]} ms per call = {16 * 15 / min(tp):.1f} frames/s (best)")
print(f"clips bit-identical: {same}")
