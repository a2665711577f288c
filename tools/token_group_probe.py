#!/usr/bin/env python3
"""One token loop over the rows of SEVERAL batches beside the frame decoder of one batch: what would a schedule that
groups the token stages of G consecutive batches into one loop of G x 16 rows (weights streamed once per step for all of
them) reach?  For each row count: decoder ms per batch and token-loop ms per step while both run, and the batch period
max(decoder, 960 steps / G) they imply.   python tools/token_group_probe.py [rows ...]"""
import os
import sys
import threading
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ccvs_amd.tools.options import Options, BAIR_ARGV  # noqa: E402
from ccvs_amd.helpers.generator import Generator  # noqa: E402

rows_list = [int(v) for v in sys.argv[1:]] or [16, 32, 48, 64]
batch = 16
opt = Options().parse(load_qvid_generator=True, load_transformer=True,
                      argv=list(BAIR_ARGV) + ["--batch_size_vid", str(batch), "--x_sample_noise", "device", "--rec_pass", "false"])
torch.manual_seed(0)
gen = Generator(opt).build_models()
data = {"vid": gen.synthetic_batch(batch, seed=1)["vid"].cuda()}
s_bg = torch.cuda.Stream(priority=-1)
STEPS = 240
with torch.no_grad():
    enc = gen.vid_model(data, mode="vid_encoder")
    code = torch.randint(0, 1024, (batch, 1024), generator=torch.Generator().manual_seed(2)).cuda()
    code[:, :64] = enc["code"][:, :64]
    inter = [f[:, :1].contiguous() for f in enc["inter"]]
    del enc

    def decode(n):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n):
            gen.vid_model({"code": code, "inter": inter}, mode="vid_decoder")
        e1.record()
        e1.synchronize()
        return e0.elapsed_time(e1) / n

    decode(1)
    print(f"decoder alone: {decode(2):.0f} ms per batch", flush=True)
    if os.environ.get("DECODER_GRAPH", "0") == "1":     # the decoder as ONE hipGraph: the host is quiet while it runs
        s_cap = torch.cuda.Stream()
        dgraph = torch.cuda.CUDAGraph()
        with torch.cuda.stream(s_cap):
            with torch.cuda.graph(dgraph, stream=s_cap):
                gen.vid_model({"code": code, "inter": inter}, mode="vid_decoder")
        eager_decode = decode

        def decode(n):   # noqa: F811
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(n):
                dgraph.replay()
            e1.record()
            e1.synchronize()
            return e0.elapsed_time(e1) / n

        decode(1)
        print(f"decoder as one hipGraph, alone: {decode(2):.0f} ms per batch", flush=True)
    net = gen.transformer_model.net_t
    for rows in rows_list:
        cond = torch.randint(0, 1024, (rows, 512), generator=torch.Generator().manual_seed(3)).cuda()   # mid-clip context length
        net.drop_engine_state()
        with torch.cuda.stream(s_bg):
            net.generate(cond, 16, sample=True, top_k=100, noise="device")      # capture
            s_bg.synchronize()
            t0 = time.perf_counter()
            net.generate(cond, STEPS, sample=True, top_k=100, noise="device")
            s_bg.synchronize()
            alone = (time.perf_counter() - t0) * 1e3 / STEPS
        stop, count, t_tok = False, [0], [0.0]

        def background():
            with torch.cuda.stream(s_bg):
                t0 = time.perf_counter()
                while not stop:
                    net.generate(cond, STEPS, sample=True, top_k=100, noise="device")
                    s_bg.synchronize()
                    count[0] += STEPS
                t_tok[0] = time.perf_counter() - t0

        th = threading.Thread(target=background)
        th.start()
        time.sleep(0.3)
        ms_dec = decode(5)
        stop = True
        th.join()
        torch.cuda.synchronize()
        ms_tok = 1e3 * t_tok[0] / max(count[0], 1)      # includes the prefill of the 512-token context once per 240 steps
        g = rows / batch
        period = max(ms_dec, 960 * ms_tok / g)
        print(f"rows {rows}: token step {alone:.2f} ms alone, {ms_tok:.2f} ms beside the decoder; decoder {ms_dec:.0f} ms per batch beside it; "
              f"batch period max({ms_dec:.0f}, {960 * ms_tok / g:.0f}) = {period:.0f} ms -> {240 / period * 1e3:.0f} frames/s", flush=True)
