"""Decode-step timing of the BAIR-size GPT (24 x 1024, 16 heads, batch 16): us per token, pipelined vs one stream.
usage: python tools/gpt_decode_bench.py [tokens] [batch]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ccvs_amd.models.skip_vid_generator.models import mingpt  # noqa: E402

tokens = int(sys.argv[1]) if len(sys.argv) > 1 else 300
batch = int(sys.argv[2]) if len(sys.argv) > 2 else 16
torch.manual_seed(0)
net = mingpt.GPT(vocab_size=1024, block_size=1024, num_blocks=16, n_layer=24, n_head=16, n_embd=1024, emb_mode="temporal", shape=(8, 8)).cuda()
code = torch.randint(0, 1024, (batch, 64), device="cuda")
modes = [m for m in os.environ.get("MODES", "1,0").split(",")]
for pipe in modes:
    mingpt.DECODE_PIPELINE = pipe == "1"
    net.drop_engine_state()
    out = net.generate(code, 8, sample=True, top_k=100)  # capture
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    out = net.generate(code, tokens, sample=True, top_k=100)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f"pipeline={pipe}: {dt * 1e6 / tokens:.1f} us/token ({tokens} tokens, batch {batch})", flush=True)
