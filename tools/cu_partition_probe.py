#!/usr/bin/env python3
"""Feasibility probe: the latency-bound transformer decode loop on a CU-masked stream (N CUs) beside the conv-bound frame
decoder on the complementary CUs.  Prints us/token and decode ms, alone and concurrently.
    python tools/cu_partition_probe.py [n_cu_transformer]"""
import ctypes
import os
import sys
import threading
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ccvs_amd.tools.options import Options, BAIR_ARGV  # noqa: E402
from ccvs_amd.helpers.generator import Generator  # noqa: E402

hip = ctypes.CDLL("libamdhip64.so")


def masked_stream(bits):
    words = (ctypes.c_uint32 * 8)(*[(bits >> (32 * i)) & 0xFFFFFFFF for i in range(8)])
    st = ctypes.c_void_p()
    rc = hip.hipExtStreamCreateWithCUMask(ctypes.byref(st), 8, words)
    assert rc == 0, rc
    return torch.cuda.ExternalStream(st.value)


n_t = int(sys.argv[1]) if len(sys.argv) > 1 else 64
# CU i of the mask: take every (256 // n_t)-th CU for the transformer so that it gets a slice of every XCD / shader engine
stride = 256 // n_t
t_bits = sum(1 << i for i in range(256) if i % stride == 0)
d_bits = ((1 << 256) - 1) ^ t_bits
st_t, st_d = masked_stream(t_bits), masked_stream(d_bits)

opt = Options().parse(load_qvid_generator=True, load_transformer=True, argv=list(BAIR_ARGV) + ["--x_sample_noise", "device"])
torch.manual_seed(0)
gen = Generator(opt).build_models()
batch = 16
data = {"vid": gen.synthetic_batch(batch, seed=1)["vid"].cuda()}
net = gen.transformer_model.net_t
code0 = torch.randint(0, 1024, (batch, 64), device="cuda")
with torch.no_grad():
    enc = gen.vid_model(data, mode="vid_encoder")
    code = torch.randint(0, 1024, (batch, 1024), device="cuda")
    inter = [f[:, :1].contiguous() for f in enc["inter"]]


def run_t(stream, tokens=400):
    with torch.no_grad(), torch.cuda.stream(stream):
        net.generate(code0, 8, sample=True, top_k=100)       # capture on this stream
        stream.synchronize()
        t0 = time.perf_counter()
        net.generate(code0, tokens, sample=True, top_k=100)
        stream.synchronize()
        return (time.perf_counter() - t0) * 1e6 / tokens


def run_d(stream, reps=1):
    with torch.no_grad(), torch.cuda.stream(stream):
        gen.vid_model({"code": code, "inter": inter}, mode="vid_decoder")
        stream.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            gen.vid_model({"code": code, "inter": inter}, mode="vid_decoder")
        stream.synchronize()
        return (time.perf_counter() - t0) * 1e3 / reps


full = torch.cuda.current_stream()
if os.environ.get("MASK_CHECK"):
    print(f"mask check: decoder on the {n_t}-CU stream {run_d(st_t):7.1f} ms (all CUs: {run_d(full):7.1f} ms)")
    sys.exit(0)
print(f"alone, all CUs     : transformer {run_t(full):7.1f} us/token   decoder {run_d(full):7.1f} ms")
print(f"alone, masked      : transformer ({n_t} CUs) {run_t(st_t):7.1f} us/token   decoder ({256 - n_t} CUs) {run_d(st_d):7.1f} ms")
res = {}
th = threading.Thread(target=lambda: res.__setitem__("d", run_d(st_d, reps=1)))
th.start()
res["t"] = run_t(st_t, tokens=700)
th.join()
print(f"concurrent, masked : transformer {res['t']:7.1f} us/token   decoder {res['d']:7.1f} ms")
