#!/usr/bin/env python3
"""Why does the frame decoder slow down beside the token loop?  Decode one BAIR batch (random tokens) three times:
alone; beside a hipGraph of 123 EMPTY kernel launches replayed back to back on a high-priority stream (kernel boundaries --
cache write-back / invalidate, dispatch -- without any memory traffic or CU occupancy to speak of); beside the real token
loop.   python tools/decoder_noise_probe.py [batch]"""
import os
import sys
import threading

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ccvs_amd.tools.options import Options, BAIR_ARGV  # noqa: E402
from ccvs_amd.helpers.generator import Generator  # noqa: E402

batch = int(sys.argv[1]) if len(sys.argv) > 1 else 16
opt = Options().parse(load_qvid_generator=True, load_transformer=True,
                      argv=list(BAIR_ARGV) + ["--batch_size_vid", str(batch), "--x_sample_noise", "device", "--rec_pass", "false"])
torch.manual_seed(0)
gen = Generator(opt).build_models()
data = {"vid": gen.synthetic_batch(batch, seed=1)["vid"].cuda()}
s_bg = torch.cuda.Stream(priority=-1)
with torch.no_grad():
    enc = gen.vid_model(data, mode="vid_encoder")
    code = torch.randint(0, 1024, (batch, 1024), generator=torch.Generator().manual_seed(2)).cuda()
    code[:, :64] = enc["code"][:, :64]
    inter = [f[:, :1].contiguous() for f in enc["inter"]]
    del enc

    def decode():
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        gen.vid_model({"code": code, "inter": inter}, mode="vid_decoder")
        e1.record()
        e1.synchronize()
        return e0.elapsed_time(e1)

    decode()
    print(f"decoder alone: {decode():.0f} ms", flush=True)

    # a graph of 123 empty launches (one 64-thread workgroup that does nothing)
    tiny = torch.zeros(64, device="cuda")
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.stream(s_bg):
        for _ in range(3):
            tiny.add_(0)
        s_bg.synchronize()
        with torch.cuda.graph(graph, stream=s_bg):
            for _ in range(123):
                tiny.add_(0)
    stop = False

    def background(fn):
        with torch.cuda.stream(s_bg):
            n = 0
            while not stop:
                fn()
                n += 1
                if n % 64 == 0:
                    s_bg.synchronize()      # keep the queue a few hundred replays deep, not unbounded
        return n

    for name, fn in (("123 empty launches per replay", graph.replay),):
        stop = False
        th = threading.Thread(target=background, args=(fn,))
        th.start()
        ms = decode()
        stop = True
        th.join()
        torch.cuda.synchronize()
        print(f"decoder beside {name}: {ms:.0f} ms", flush=True)

    # the real token loop in the background
    tr = gen.transformer_model
    cond = code[:, :64].contiguous()

    def tokens():
        tr.net_t.generate(cond, 200, sample=True, top_k=100, noise="device")

    with torch.cuda.stream(s_bg):
        tokens()
    stop = False
    th = threading.Thread(target=background, args=(tokens,))
    th.start()
    ms = decode()
    stop = True
    th.join()
    torch.cuda.synchronize()
    print(f"decoder beside the token loop: {ms:.0f} ms", flush=True)
