"""A chain of small dependent kernels (the shape of the decode step's GEMMs) beside the real frame decoder, for several
workgroup footprints: microseconds per launch alone and while the decoder of a BAIR batch runs on another stream.
    python tools/chain_probe.py         (builds tools/micro/libchainprobe.so on the box if it is missing)"""
import ctypes
import os
import subprocess
import sys
import threading

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
so = os.path.join(ROOT, "tools", "micro", "libchainprobe.so")
if not os.path.exists(so):
    subprocess.run(["/opt/rocm/bin/hipcc", "-O2", "--offload-arch=gfx950", "-shared", "-fPIC", os.path.join(ROOT, "tools", "micro", "chain_probe.hip"), "-o", so], check=True)
lib = ctypes.CDLL(so)
lib.chain_launch.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_long, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int]
lib.dma_launch.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_long, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]

from ccvs_amd.tools.options import Options, BAIR_ARGV  # noqa: E402
from ccvs_amd.helpers.generator import Generator  # noqa: E402

opt = Options().parse(load_qvid_generator=True, load_transformer=True,
                      argv=list(BAIR_ARGV) + ["--batch_size_vid", "16", "--x_sample_noise", "device", "--rec_pass", "false"])
torch.manual_seed(0)
gen = Generator(opt)
from ccvs_amd.models.skip_vid_generator.models.quantized_video_model import QVidModel  # noqa: E402
gen.vid_model = QVidModel(gen.qvid_opt, is_train=False, is_main=True, logger=None).eval()
data = {"vid": gen.synthetic_batch(16, seed=1)["vid"].cuda()}
with torch.no_grad():
    enc = gen.vid_model(data, mode="vid_encoder")
    code = torch.randint(0, 1024, (16, 1024), generator=torch.Generator().manual_seed(2)).cuda()
    code[:, :64] = enc["code"][:, :64]
    inter = [f[:, :1].contiguous() for f in enc["inter"]]
    del enc
s_dec, s_chain = torch.cuda.Stream(), torch.cuda.Stream(priority=-1)
weights = torch.randn(24 * 4 << 20, device="cuda")          # 24 "layers" x 16 MB, so that successive launches stream fresh bytes
big = torch.randn(768 << 20 >> 2, device="cuda")            # 768 MB for the attention-sized launches (192 MB each, 4 regions)
buf = [torch.zeros(16384, device="cuda"), torch.zeros(16384, device="cuda")]
LAUNCHES = 120


def make_graph(threads, vg, wgs, mbytes, lds_kb=0, dma=None):
    g = torch.cuda.CUDAGraph()
    n_launch = LAUNCHES if mbytes <= 16 else 24
    with torch.cuda.stream(s_chain):
        def body():
            for i in range(n_launch):
                base = weights.data_ptr() + (i % 24) * (16 << 20) if mbytes <= 16 else big.data_ptr() + (i % 4) * (192 << 20)
                st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
                if dma is not None:
                    rc = lib.dma_launch(dma[0], dma[1], wgs, ctypes.c_void_p(base), mbytes << 20, ctypes.c_void_p(buf[i & 1].data_ptr()),
                                        ctypes.c_void_p(buf[(i + 1) & 1].data_ptr()), st)
                else:
                    rc = lib.chain_launch(threads, vg, wgs, ctypes.c_void_p(base), mbytes << 20, ctypes.c_void_p(buf[i & 1].data_ptr()),
                                          ctypes.c_void_p(buf[(i + 1) & 1].data_ptr()), st, lds_kb)
                assert rc == 0, rc
        body()
        s_chain.synchronize()
        with torch.cuda.graph(g, stream=s_chain, capture_error_mode="thread_local"):
            body()
    g.n_launch = n_launch
    return g


def time_chain(g, reps):
    with torch.cuda.stream(s_chain):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        g.replay()
        e0.record()
        for _ in range(reps):
            g.replay()
        e1.record()
    e1.synchronize()
    return 1e3 * e0.elapsed_time(e1) / (reps * g.n_launch)


def decode(n):
    with torch.cuda.stream(s_dec), torch.no_grad():
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n):
            gen.vid_model({"code": code, "inter": inter}, mode="vid_decoder")
        e1.record()
    return e0, e1


e0, e1 = decode(1)
e1.synchronize()
e0, e1 = decode(1)
e1.synchronize()
print(f"decoder alone: {e0.elapsed_time(e1):.0f} ms per batch", flush=True)
# (threads, VGPRs, workgroups, MB per launch, LDS KB, dma (stages, wave-instructions per stage) or None)
cases = [(512, 200, 256, 16, 0, None), (512, 100, 256, 16, 0, None), (256, 48, 256, 16, 0, None), (256, 56, 256, 16, 0, None), (256, 64, 256, 16, 0, None),
         (256, 80, 256, 16, 0, None), (256, 48, 256, 16, 32, None), (256, 48, 256, 16, 48, None), (256, 48, 256, 16, 56, None), (256, 48, 256, 16, 64, None),
         (256, 0, 256, 16, 0, (3, 4)), (256, 0, 256, 16, 0, (3, 3)), (256, 0, 256, 16, 0, (2, 4)), (256, 0, 512, 16, 0, (3, 2)), (256, 0, 256, 16, 0, (4, 3)),
         (256, 0, 256, 4, 0, (3, 4)), (256, 48, 256, 4, 0, None),
         (512, 100, 768, 192, 0, None), (256, 48, 768, 192, 0, None), (256, 0, 768, 192, 0, (3, 4)), (256, 0, 768, 192, 0, (3, 3)), (256, 0, 768, 192, 0, (4, 3)),
         (256, 0, 768, 192, 0, (2, 4)), (256, 0, 256, 192, 0, (3, 4))]
for threads, vg, wgs, mb, lds_kb, dma in cases:
    g = make_graph(threads, vg, wgs, mb, lds_kb, dma)
    alone = time_chain(g, 10 if mb <= 16 else 4)
    torch.cuda.synchronize()
    d0, d1 = decode(3)          # enqueued from this thread (the GPU needs ~2.6 s for it, the host ~1 s): no second host thread, no GIL effects
    beside = time_chain(g, (40 if mb <= 16 else 8))
    still = not d1.query()      # the decoder was still running when the chain finished
    torch.cuda.synchronize()
    what = f"LDS-DMA {dma[0]} stages x {dma[1]} KB per wave ({dma[0] * dma[1] * 4} KB LDS)" if dma else f"{threads:3d} threads x {vg:3d} VGPRs, {lds_kb:2d} KB LDS"
    print(f"{what:52s} {wgs:4d} workgroups, {mb:3d} MB per launch: alone {alone:6.1f} us/launch ({mb * 1.048576 / alone:5.2f} TB/s), beside the decoder "
          f"{beside:6.1f} us/launch ({mb * 1.048576 / beside:5.2f} TB/s){'' if still else '  [decoder had finished: invalid]'}", flush=True)
