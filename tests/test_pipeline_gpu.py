"""-m gpu: Generator.run_pipelined (two batches in flight: decoder of batch i on one stream, token loop of batch i+1 on a
high-priority stream, convolutions capped to a share of the CUs) must give exactly what the serial schedule gives."""
import pytest
import torch

from tests.test_e2e_gpu import tiny, TINY_ARGV  # noqa: F401  (fixture)

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("sample,cu_limit,lanes", [(False, 0, 1), (True, 6, 2), (True, 200, 3)])
def test_pipelined_equals_serial(tiny, sample, cu_limit, lanes):
    from ccvs_amd.helpers.generator import Generator
    from ccvs_amd import ops
    xopt = tiny["xopt"]
    xopt.sample, xopt.top_k, xopt.rec_pass = sample, 10, False
    old_noise = tiny["tr"].sample_noise
    tiny["tr"].sample_noise = "device"
    try:
        gen = Generator(tiny["opt"])
        gen.vid_model, gen.transformer_model = tiny["qv"], tiny["tr"]
        batches = [gen.synthetic_batch(2, seed=80 + i)["vid"] for i in range(5)]
        serial = [gen.generate_vid({"vid": b.clone()}, global_iter=10 + i) for i, b in enumerate(batches)]
        packed = []
        res = gen.run_pipelined(({"vid": b.clone()} for b in batches), first_iter=10, cu_limit=cu_limit, lanes=lanes,
                                finish=lambda i, fake: packed.append((i, ops.pack_u8(fake["vid"]))))
        torch.cuda.synchronize()
        assert [r["index"] for r in res] == [10, 11, 12, 13, 14] and [i for i, _ in packed] == [10, 11, 12, 13, 14]
        for want, got, (_, u8) in zip(serial, res, packed):
            assert torch.equal(got["enc_code"], want["enc_code"])
            assert torch.equal(got["fake"]["code"], want["fake"]["code"])
            assert torch.equal(got["fake"]["vid"], want["fake"]["vid"]), "pipelined clip differs from the serial schedule"
            assert torch.equal(u8, ops.pack_u8(want["fake"]["vid"]))
        if sample:
            assert not torch.equal(res[0]["fake"]["code"], res[1]["fake"]["code"])
        ms = gen.pipeline_stage_ms()
        assert all(v > 0 for v in ms.values())
    finally:
        xopt.sample, xopt.rec_pass = False, True
        tiny["tr"].sample_noise = old_noise


def test_conv_cu_limit_is_bit_identical():
    """ccvs_conv_desc.cu_limit only changes how the tiles are launched (chunks of cu_limit x occupancy workgroups): same
    values for dense 3x3, stride-2, transposed, 1x1 and head-shaped layers, ragged sizes included."""
    from ccvs_amd import ops
    g = torch.Generator().manual_seed(0)
    cases = [  # (N, Cin, H, W, Cout, k, stride, pad, transposed)
        (5, 48, 64, 64, 128, 3, 1, 1, False), (3, 195, 40, 72, 64, 3, 1, 1, False), (4, 32, 33, 47, 27, 1, 1, 0, False),
        (3, 64, 66, 66, 96, 3, 2, 0, False), (3, 96, 17, 17, 48, 3, 2, 0, True), (6, 24, 64, 64, 32, 1, 1, 0, False),
    ]
    for n, cin, h, w, cout, k, stride, pad, tr in cases:
        x = torch.randn(n, cin, h, w, generator=g).cuda()
        wt = torch.randn(cout, cin, k, k, generator=g).cuda()
        bias = torch.randn(cout, generator=g).cuda()
        pk = ops.pack_conv_weight(wt)
        outs = []
        for lim in (0, 3, 61):
            ops.CONV_CU_LIMIT = lim
            try:
                outs.append(ops.conv2d(x, pk, bias, cout, k, stride=stride, pad=pad, transposed=tr, act=True))
            finally:
                ops.CONV_CU_LIMIT = 0
        assert torch.equal(outs[0], outs[1]) and torch.equal(outs[0], outs[2]), (n, cin, h, w, cout, k, stride, pad, tr)



def test_stream_cu_budget_is_bit_identical(tiny):
    """ccvs_stream_cu_limit turns the HBM-bound kernels into persistent grids and chunks the convolutions: a whole decoder
    frame (FIR, cost volume, warps, tap sums, depthwise up-sampling, convolutions) and the uint8 pack give the same bits
    with a budget of 3 CUs, 100 CUs and none."""
    from ccvs_amd import ops
    g, qv = tiny["gold"], tiny["qv"]
    vid = torch.from_numpy(g["vid"])
    side = torch.cuda.Stream()
    outs = []
    for lim in (0, 3, 100):
        ops.stream_cu_limit(side, lim)
        try:
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                enc = qv({"vid": vid.clone()}, mode="vid_encoder")
                inter = [f[:, :1].contiguous() for f in enc["inter"]]
                fake = qv({"code": torch.from_numpy(g["gen_code_greedy"]), "inter": inter}, mode="vid_decoder")["vid"]
                outs.append((enc["code"], fake, ops.pack_u8(fake)))
            torch.cuda.current_stream().wait_stream(side)
        finally:
            ops.stream_cu_limit(side, 0)
    torch.cuda.synchronize()
    for other in outs[1:]:
        for a, b in zip(outs[0], other):
            assert torch.equal(a, b)


def test_rccl_group_alive_serial_and_pipelined():
    """tools/nccl_single_rank_check.py: with an RCCL process group initialised (watchdog thread running) the decode-step
    hipGraph is captured, the clips are all-gathered on a side stream, and the pipelined schedule equals the serial one."""
    import os
    import socket
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    res = subprocess.run([sys.executable, os.path.join(root, "tools", "nccl_single_rank_check.py"), str(port)], capture_output=True, text=True,
                         timeout=600, env=env)
    assert res.returncode == 0 and "ok:" in res.stdout, res.stdout[-2000:] + res.stderr[-4000:]
