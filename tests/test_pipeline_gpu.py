"""-m gpu: Generator.run_pipelined (two batches in flight: decoder of batch i on one stream, token loop of batch i+1 on a
high-priority stream, convolutions capped to a share of the CUs) must give exactly what the serial schedule gives."""
import pytest
import torch

from tests.test_e2e_gpu import tiny, TINY_ARGV  # noqa: F401  (fixture)

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("sample,cu_limit", [(False, 0), (True, 6), (True, 200)])
def test_pipelined_equals_serial(tiny, sample, cu_limit):
    from ccvs_amd.helpers.generator import Generator
    from ccvs_amd import ops
    xopt = tiny["xopt"]
    xopt.sample, xopt.top_k, xopt.rec_pass = sample, 10, False
    old_noise = tiny["tr"].sample_noise
    tiny["tr"].sample_noise = "device"
    try:
        gen = Generator(tiny["opt"])
        gen.vid_model, gen.transformer_model = tiny["qv"], tiny["tr"]
        batches = [gen.synthetic_batch(2, seed=80 + i)["vid"] for i in range(4)]
        serial = [gen.generate_vid({"vid": b.clone()}, global_iter=10 + i) for i, b in enumerate(batches)]
        packed = []
        res = gen.run_pipelined(({"vid": b.clone()} for b in batches), first_iter=10, cu_limit=cu_limit,
                                finish=lambda i, fake: packed.append((i, ops.pack_u8(fake["vid"]))))
        torch.cuda.synchronize()
        assert [r["index"] for r in res] == [10, 11, 12, 13] and [i for i, _ in packed] == [10, 11, 12, 13]
        for want, got, (_, u8) in zip(serial, res, packed):
            assert torch.equal(got["enc_code"], want["enc_code"])
            assert torch.equal(got["fake"]["code"], want["fake"]["code"])
            assert torch.equal(got["fake"]["vid"], want["fake"]["vid"]), "pipelined clip differs from the serial schedule"
            assert torch.equal(u8, ops.pack_u8(want["fake"]["vid"]))
        if sample:
            assert not torch.equal(res[0]["fake"]["code"], res[1]["fake"]["code"])
        ms = gen.pipeline_stage_ms()
        assert all(v > 0 for v in ms.values())
        assert ops.CONV_CU_LIMIT == 0
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


def test_compact_decode_gemms_bit_identical():
    """`ccvs_gpt_decode.gemm_tiles`: the compact GEMM forms (T tiles per workgroup / whole K depth in one workgroup, 64
    workgroups per launch) add in the same order as the whole-chip forms -- same logits, same tokens, bit for bit, at the
    BAIR width (1024 / 4096, 16 heads), batch 16 and a ragged batch of 5."""
    from ccvs_amd import ops
    from ccvs_amd.models.skip_vid_generator.models.mingpt import GPT
    torch.manual_seed(0)
    net = GPT(vocab_size=1024, block_size=1024, num_blocks=16, n_layer=3, n_head=16, n_embd=1024, emb_mode="temporal", shape=[8, 8]).cuda().eval()
    with torch.no_grad():
        net.s_emb.normal_(0, 0.02)
        net.t_emb.normal_(0, 0.02)
        for blk in net.blocks:
            blk.ln1.weight.normal_(1, 0.1)
            blk.ln2.bias.normal_(0, 0.1)
    old = ops.DECODE_GEMM_TILES
    try:
        for batch in (16, 5):
            code = torch.randint(0, 1024, (batch, 70), generator=torch.Generator().manual_seed(batch)).cuda()
            runs = {}
            for tiles in (0, 1, 2, 4):
                ops.DECODE_GEMM_TILES = tiles
                trace = []
                out = net.generate(code, 6, sample=False, trace=trace)       # eager steps: the logits of every step are kept
                graph = net.generate(code, 40, sample=True, top_k=100, noise="device")   # captured steps, in-kernel noise
                runs[tiles] = (out, torch.stack(trace), graph)
            for tiles in (1, 2, 4):
                for a, b in zip(runs[0], runs[tiles]):
                    assert torch.equal(a, b), (batch, tiles)
    finally:
        ops.DECODE_GEMM_TILES = old
