"""The decode step of the BAIR-size GPT (24 x 1024, 16 heads) over the stacked rows of several batches: ms per step for each
row count, alone on the chip.  usage: python tools/token_step_probe.py [tokens] [rows ...]   (CCVS_GEMM_RB=1: one weight pass
per 16 rows, the round-2 behaviour)."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ccvs_amd.models.skip_vid_generator.models import mingpt  # noqa: E402

tokens = int(sys.argv[1]) if len(sys.argv) > 1 else 400
rows_list = [int(v) for v in sys.argv[2:]] or [16, 32, 48, 64]
graph = os.environ.get("CCVS_PROBE_EAGER", "0") != "1"   # 1: every step launched eagerly (rocprofv3 --pmc cannot count inside a hipGraph replay)
torch.manual_seed(0)
net = mingpt.GPT(vocab_size=1024, block_size=1024, num_blocks=16, n_layer=24, n_head=16, n_embd=1024, emb_mode="temporal", shape=(8, 8)).cuda()
for rows in rows_list:
    groups = max(rows // 16, 1)
    code = torch.randint(0, 1024, (rows, 64), device="cuda")
    net.noise_key, net.row_offset, net.noise_call = [(1 + g, 2 + g) for g in range(groups)], [0] * groups, 0
    net.generate(code, 20, sample=True, top_k=100, use_graph=graph)  # capture
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    net.generate(code, tokens, sample=True, top_k=100, use_graph=graph)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f"rows {rows} ({groups} groups): {dt * 1e3 / tokens:.3f} ms/step = {dt * 1e3 / tokens / groups:.3f} ms per 16-row batch-token "
          f"({tokens} tokens from cache length 64)", flush=True)
    net.drop_engine_state()
