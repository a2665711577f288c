"""How does the token loop (BAIR-size GPT, batch 16, captured decode step) behave when only R compute units are free?
A do-nothing kernel holds 256 - R CUs (tools/micro/cu_hog.hip) while the loop runs on a high-priority stream.
usage: python tools/token_hog_probe.py [tokens]      (needs tools/micro/libcuhog.so)"""
import ctypes
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from ccvs_amd.models.skip_vid_generator.models import mingpt  # noqa: E402

hog = ctypes.CDLL(os.path.join(ROOT, "tools", "micro", "libcuhog.so"))
hog.cu_hog_launch.argtypes = [ctypes.c_int, ctypes.c_double, ctypes.c_void_p]
hog.mem_hog_launch.argtypes = [ctypes.c_int, ctypes.c_double, ctypes.c_void_p, ctypes.c_long, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p]
tokens = int(sys.argv[1]) if len(sys.argv) > 1 else 400
torch.manual_seed(0)
net = mingpt.GPT(vocab_size=1024, block_size=1024, num_blocks=16, n_layer=24, n_head=16, n_embd=1024, emb_mode="temporal", shape=(8, 8)).cuda()
code = torch.randint(0, 1024, (16, 64), device="cuda")
s_tok, s_hog = torch.cuda.Stream(priority=-1), torch.cuda.Stream()
with torch.cuda.stream(s_tok):
    net.generate(code, 8, sample=True, top_k=100)   # capture


big = torch.empty(2 << 30, dtype=torch.uint8, device="cuda")
sink = torch.zeros(4, device="cuda")


def run(n_hog, mem=False):
    torch.cuda.synchronize()
    if n_hog:
        secs = 1.6e-3 * tokens * 4 + 0.2
        if mem:
            hog.mem_hog_launch(n_hog, secs, ctypes.c_void_p(big.data_ptr()), big.numel(), ctypes.c_void_p(sink.data_ptr()), 150,
                               ctypes.c_void_p(s_hog.cuda_stream))
        else:
            hog.cu_hog_launch(n_hog, secs, ctypes.c_void_p(s_hog.cuda_stream))
        time.sleep(0.05)
    with torch.cuda.stream(s_tok):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        net.generate(code, tokens, sample=True, top_k=100)
        e1.record()
    e1.synchronize()
    ms = e0.elapsed_time(e1)
    torch.cuda.synchronize()
    return 1e3 * ms / tokens


for n_hog in (0, 128, 192, 224, 240):
    print(f"idle hog on {n_hog:3d} CUs (free {256 - n_hog:3d}): {run(n_hog):7.1f} us/token over {tokens} tokens", flush=True)
for n_hog in (64, 128, 192):
    print(f"STREAMING hog (1 workgroup of 512 threads per CU, float4 reads) on {n_hog:3d} CUs: {run(n_hog, True):7.1f} us/token", flush=True)
