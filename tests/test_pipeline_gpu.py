"""-m gpu: Generator.run_pipelined (several batches in flight: the decoder of one batch on one stream, ONE token loop over the
stacked rows of the next `lanes` batches on a high-priority stream) must give exactly what the serial schedule gives."""
import os

import pytest
import torch

from tests.test_e2e_gpu import tiny, TINY_ARGV  # noqa: F401  (fixture)

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("sample,cu_limit,lanes,chains,ramp,by_frame,dec_streams", [
    (False, 0, 1, 1, (), True, 1), (True, 6, 2, 1, (), True, 2), (True, 200, 3, 2, (), True, 2), (True, 0, 2, 2, (1,), True, 1), (True, 0, 1, 3, (), True, 3),
    (True, 0, 3, 2, (), False, 2), (False, 0, 2, 1, (1,), False, 1), (True, 0, 4, 2, (), True, 2)])
def test_pipelined_equals_serial(tiny, monkeypatch, sample, cu_limit, lanes, chains, ramp, by_frame, dec_streams):
    """`by_frame`: the decoder takes the tokens of a batch frame by frame while its token loop is still running
    (`CCVS_PIPELINE_STREAM`, the default) / when the whole token stage is done.  `dec_streams`: the decode of batch i runs on stream
    i % dec_streams (the default: 2 for frames of 128^2 and more)."""
    from ccvs_amd.helpers.generator import Generator
    from ccvs_amd import ops
    monkeypatch.setenv("CCVS_PIPELINE_STREAM", "1" if by_frame else "0")
    monkeypatch.setenv("CCVS_PIPELINE_DEC_STREAMS", str(dec_streams))
    xopt = tiny["xopt"]
    xopt.sample, xopt.top_k, xopt.rec_pass = sample, 10, False
    old_noise = tiny["tr"].sample_noise
    tiny["tr"].sample_noise = "device"
    try:
        gen = Generator(tiny["opt"])
        gen.vid_model, gen.transformer_model = tiny["qv"], tiny["tr"]
        batches = [gen.synthetic_batch(2, seed=80 + i)["vid"] for i in range(5)]
        serial = [gen.generate_vid({"vid": b.clone()}, global_iter=10 + i, schedule="serial") for i, b in enumerate(batches)]
        packed = []
        res = gen.run_pipelined(({"vid": b.clone()} for b in batches), first_iter=10, cu_limit=cu_limit, lanes=lanes, chains=chains, ramp=ramp,
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
        sizes = [g for g, _ in gen.pipeline_token_groups()]
        assert sum(sizes) == 5 and max(sizes) <= lanes and (not ramp or sizes[0] == ramp[0])
        # 4 frames, 1 given: the decoder runs a batch in 1 + 3 pieces and got its tokens in 4 hand-overs (one when not by frame)
        assert all(len(ev["segs"]) == 4 and ev["hand_overs"] == (4 if by_frame else 1) for ev in gen._pipeline_events)
    finally:
        xopt.sample, xopt.rec_pass = False, True
        tiny["tr"].sample_noise = old_noise


@pytest.mark.parametrize("sample,noise,rec", [(False, "device", False), (True, "device", True), (True, "host", False), (True, "host", True)])
def test_generate_vid_stream_equals_serial(tiny, sample, noise, rec):
    """`generate_vid(schedule="stream")` -- the single call with the decoder following the token loop frame by frame, the default since
    round 6 -- against `schedule="serial"` (token loop, then decode: the reference's order, helpers/generator.py:161-164): codes, fp32
    pixels, the rec pass and the stage timers, greedy / Philox / the reference's host-drawn noise under one seed; and the default IS the
    streamed schedule."""
    from ccvs_amd.helpers.generator import Generator
    xopt = tiny["xopt"]
    xopt.sample, xopt.top_k, xopt.rec_pass = sample, 10, rec
    old_noise = tiny["tr"].sample_noise
    tiny["tr"].sample_noise = noise
    try:
        gen = Generator(tiny["opt"])
        gen.vid_model, gen.transformer_model = tiny["qv"], tiny["tr"]
        for i in range(3):
            data = gen.synthetic_batch(2, seed=300 + i)["vid"]
            torch.manual_seed(40 + i)
            want = gen.generate_vid({"vid": data.clone()}, global_iter=7 + i, schedule="serial")
            ms_serial = gen.stage_ms()
            torch.manual_seed(40 + i)
            got = gen.generate_vid({"vid": data.clone()}, global_iter=7 + i, schedule="stream")
            ms_stream = gen.stage_ms()
            torch.manual_seed(40 + i)
            dflt = gen.generate_vid({"vid": data.clone()}, global_iter=7 + i)
            torch.cuda.synchronize()
            for out in (got, dflt):
                assert torch.equal(out["enc_code"], want["enc_code"])
                assert torch.equal(out["fake"]["code"], want["fake"]["code"])
                assert torch.equal(out["fake"]["vid"], want["fake"]["vid"]), "streamed single call differs from the serial order"
                assert (out["rec"] is None) == (want["rec"] is None)
                if rec:
                    assert torch.equal(out["rec"]["vid"], want["rec"]["vid"])
            assert all(v > 0 for v in ms_serial.values()) and all(v > 0 for v in ms_stream.values())
            assert len(gen._events_stream["segs"]) == 4                # the default went through the frame-by-frame decode: 1 + 3 pieces
        # a single call in between does not replace the records of the last run_pipelined (bench.py reads them after its single-call leg)
        batches = [gen.synthetic_batch(2, seed=330 + i)["vid"] for i in range(4)]
        gen.run_pipelined(({"vid": b.clone()} for b in batches), first_iter=0, lanes=2, chains=2)
        groups, events = gen.pipeline_token_groups(), gen._pipeline_events
        gen.generate_vid({"vid": batches[0].clone()}, global_iter=0)
        assert (gen.last_lanes, gen.last_chains) == (2, 2) and gen._pipeline_events is events
        assert [g for g, _ in gen.pipeline_token_groups()] == [g for g, _ in groups] == [2, 2]
    finally:
        xopt.sample, xopt.rec_pass = False, True
        tiny["tr"].sample_noise = old_noise


def test_pipelined_ragged_batch_sizes(tiny, monkeypatch):
    """Batches of another size in the middle of a run: they start their own token groups, their decode steps are captured from
    the calling thread once no worker is launching, and every clip still equals the serial schedule's."""
    from ccvs_amd.helpers.generator import Generator
    xopt = tiny["xopt"]
    xopt.sample, xopt.top_k, xopt.rec_pass = True, 10, False
    old_noise = tiny["tr"].sample_noise
    tiny["tr"].sample_noise = "device"
    try:
        gen = Generator(tiny["opt"])
        gen.vid_model, gen.transformer_model = tiny["qv"], tiny["tr"]
        sizes = [2, 2, 3, 3, 3, 1, 2]
        batches = [gen.synthetic_batch(n, seed=60 + i)["vid"] for i, n in enumerate(sizes)]
        serial = [gen.generate_vid({"vid": b.clone()}, global_iter=50 + i, schedule="serial") for i, b in enumerate(batches)]
        for tr, _ in gen._chains:            # the serial calls above left captured steps behind: start cold
            tr.net_t.drop_engine_state()
        gen.transformer_model.net_t.drop_engine_state()
        monkeypatch.setenv("CCVS_PIPELINE_DEC_STREAMS", "2")
        res = gen.run_pipelined(({"vid": b.clone()} for b in batches), first_iter=50, lanes=2, chains=2)
        torch.cuda.synchronize()
        assert [r["index"] for r in res] == list(range(50, 57))
        assert [g for g, _ in gen.pipeline_token_groups()] == [2, 2, 1, 1, 1]     # 2+2 | 3+3 | 3 | 1 | 2
        for want, got in zip(serial, res):
            assert torch.equal(got["fake"]["code"], want["fake"]["code"])
            assert torch.equal(got["fake"]["vid"], want["fake"]["vid"])
    finally:
        xopt.sample, xopt.rec_pass = False, True
        tiny["tr"].sample_noise = old_noise


def test_failed_token_stage_surfaces_and_the_next_run_is_clean(tiny, monkeypatch):
    """A token stage that dies while the decoder is already consuming its frames: run_pipelined raises that error (no hang, no
    wait for the other queued stages) and the same Generator then runs the same batches bit-equal to the serial schedule."""
    import time
    from ccvs_amd.helpers import generator as G
    xopt = tiny["xopt"]
    xopt.sample, xopt.top_k, xopt.rec_pass = True, 10, False
    old_noise = tiny["tr"].sample_noise
    tiny["tr"].sample_noise = "device"
    try:
        gen = G.Generator(tiny["opt"])
        gen.vid_model, gen.transformer_model = tiny["qv"], tiny["tr"]
        batches = [gen.synthetic_batch(2, seed=90 + i)["vid"] for i in range(5)]
        serial = [gen.generate_vid({"vid": b.clone()}, global_iter=30 + i, schedule="serial") for i, b in enumerate(batches)]
        real = G._FrameFeed.on_tokens
        calls = {"n": 0}

        def failing(self, n, codes):
            calls["n"] += 1
            if calls["n"] == 3:          # the first group has handed over two frames
                raise RuntimeError("token stage broke")
            return real(self, n, codes)

        monkeypatch.setattr(G._FrameFeed, "on_tokens", failing)
        monkeypatch.setenv("CCVS_PIPELINE_DEC_STREAMS", "2")
        t0 = time.perf_counter()
        with pytest.raises(RuntimeError, match="token stage broke"):
            gen.run_pipelined(({"vid": b.clone()} for b in batches), first_iter=30, lanes=2, chains=2)
        assert time.perf_counter() - t0 < 60
        monkeypatch.setattr(G._FrameFeed, "on_tokens", real)
        res = gen.run_pipelined(({"vid": b.clone()} for b in batches), first_iter=30, lanes=2, chains=2)
        torch.cuda.synchronize()
        for want, got in zip(serial, res):
            assert torch.equal(got["fake"]["code"], want["fake"]["code"])
            assert torch.equal(got["fake"]["vid"], want["fake"]["vid"])
    finally:
        xopt.sample, xopt.rec_pass = False, True
        tiny["tr"].sample_noise = old_noise


def test_decode_gemm_rows_do_not_depend_on_the_launch():
    """The weight-stream GEMM (`gemm16_kernel<RB>`): a row's result is the same bits whether it is computed in a launch of
    16 rows (RB = 1) or stacked with the rows of other batches (RB = 2, 3, 4, two workgroup rows): plain, split-K (K = 4096),
    LayerNorm-folded + GELU and the QKV form with its cache scatter, ragged row counts included."""
    from ccvs_amd import ops
    g = torch.Generator().manual_seed(5)
    C, F = 256, 1024
    x = torch.randn(80, C, generator=g).cuda()
    h = torch.randn(80, 4096, generator=g).cuda()
    w_proj, b_proj = (torch.randn(C, C, generator=g) * 0.05).cuda(), torch.randn(C, generator=g).cuda()
    w_fc2, b_fc2 = (torch.randn(C, 4096, generator=g) * 0.02).cuda(), torch.randn(C, generator=g).cuda()
    w_fc, b_fc = (torch.randn(F, C, generator=g) * 0.05).cuda(), torch.randn(F, generator=g).cuda()
    gamma, beta = (1 + 0.1 * torch.randn(C, generator=g)).cuda(), (0.1 * torch.randn(C, generator=g)).cuda()
    fc_packed = ops.pack_ln_linear(w_fc, b_fc, gamma, beta)
    w_qkv, b_qkv = (torch.randn(3 * C, C, generator=g) * 0.05).cuda(), torch.randn(3 * C, generator=g).cuda()
    qkv_packed = ops.pack_ln_linear(w_qkv, b_qkv, gamma, beta)
    H, Tmax = 4, 8
    for m in (16, 23, 32, 48, 64, 80):
        xs, hs = x[:m], h[:m]
        whole = [ops.gemm_nt(xs, w_proj, b_proj, ops.EPI_RESIDUAL, residual=xs),
                 ops.gemm_nt(hs, w_fc2, b_fc2, ops.EPI_RESIDUAL, residual=xs),
                 ops.gemm_ln(xs, *fc_packed, epilogue=ops.EPI_GELU)]
        kc, vc = torch.zeros(m, H, Tmax, C // H, device="cuda"), torch.zeros(m, H, Tmax, C // H, device="cuda")
        whole.append(ops.gemm_ln_qkv(xs, *qkv_packed, kc, vc, m, 1, 3))
        whole += [kc, vc]
        for lo in range(0, m, 16):
            hi = min(lo + 16, m)
            part = [ops.gemm_nt(xs[lo:hi], w_proj, b_proj, ops.EPI_RESIDUAL, residual=xs[lo:hi]),
                    ops.gemm_nt(hs[lo:hi], w_fc2, b_fc2, ops.EPI_RESIDUAL, residual=xs[lo:hi]),
                    ops.gemm_ln(xs[lo:hi], *fc_packed, epilogue=ops.EPI_GELU)]
            kc1, vc1 = torch.zeros(hi - lo, H, Tmax, C // H, device="cuda"), torch.zeros(hi - lo, H, Tmax, C // H, device="cuda")
            part.append(ops.gemm_ln_qkv(xs[lo:hi], *qkv_packed, kc1, vc1, hi - lo, 1, 3))
            part += [kc1, vc1]
            for a, b in zip(whole, part):
                assert torch.equal(a[lo:hi], b), (m, lo)
        want = torch.nn.functional.gelu(torch.nn.functional.layer_norm(xs, (C,), gamma, beta) @ w_fc.t() + b_fc)
        assert (whole[2] - want).abs().max().item() < 2e-4


def test_gemm_block_tile_switch_is_bit_identical(tmp_path):
    """the 2 x 2 block tile (`gemm16_kernel<.., 2, 2, 1>`, rows > 32) against the one-block form on the SAME M, by flipping
    CCVS_GEMM_TILE2 (read once per process, hence two worker processes): M = 48 / 64 / 37, N not a multiple of 32 (the ragged-block
    `continue` and the per-block slab indexing), K = 4096 (K also split over workgroups), with and without the LayerNorm prologue."""
    import subprocess
    import sys
    import numpy as np
    here = os.path.dirname(os.path.abspath(__file__))
    outs = []
    for flag in ("0", "1"):
        path = str(tmp_path / f"tile{flag}.npz")
        env = dict(os.environ, CCVS_GEMM_TILE2=flag)
        subprocess.run([sys.executable, os.path.join(here, "gemm_tile_worker.py"), path], env=env, check=True, timeout=600)
        outs.append(np.load(path))
    assert sorted(outs[0].files) == sorted(outs[1].files) and len(outs[0].files) == 27
    for key in outs[0].files:
        assert np.array_equal(outs[0][key], outs[1][key]), key


def test_sequence_gemm_rows_do_not_depend_on_the_launch():
    """Whole-sequence calls (`CCVS_GEMM_SEQ`; the QKV form with Tq > 1) run ONE kernel form whatever the row count: the rows of
    a 256-row prefill (a batch alone) and the same rows inside a 272- / 768-row launch (the batch stacked into a token group)
    are the same bits -- the old pick by M put 256 rows on the weight-stream kernel and 257+ on the row-blocked one, whose K
    partitions differ.  Logits-level check: plain, deep-K, LayerNorm-folded + GELU, and the QKV form with its cache scatter."""
    from ccvs_amd import ops
    g = torch.Generator().manual_seed(9)
    C, F, M = 256, 1024, 768
    x = torch.randn(M, C, generator=g).cuda()
    h = torch.randn(M, 4096, generator=g).cuda()
    w_proj, b_proj = (torch.randn(C, C, generator=g) * 0.05).cuda(), torch.randn(C, generator=g).cuda()
    w_fc2, b_fc2 = (torch.randn(C, 4096, generator=g) * 0.02).cuda(), torch.randn(C, generator=g).cuda()
    w_fc, b_fc = (torch.randn(F, C, generator=g) * 0.05).cuda(), torch.randn(F, generator=g).cuda()
    gamma, beta = (1 + 0.1 * torch.randn(C, generator=g)).cuda(), (0.1 * torch.randn(C, generator=g)).cuda()
    fc_packed = ops.pack_ln_linear(w_fc, b_fc, gamma, beta)
    w_qkv, b_qkv = (torch.randn(3 * C, C, generator=g) * 0.05).cuda(), torch.randn(3 * C, generator=g).cuda()
    qkv_packed = ops.pack_ln_linear(w_qkv, b_qkv, gamma, beta)
    H, Tq = 4, 16
    SEQ = ops.GEMM_SEQ

    def run(m):
        xs, hs = x[:m], h[:m]
        out = [ops.gemm_nt(xs, w_proj, b_proj, ops.EPI_RESIDUAL | SEQ, residual=xs),
               ops.gemm_nt(hs, w_fc2, b_fc2, ops.EPI_RESIDUAL | SEQ, residual=xs),
               ops.gemm_ln(xs, *fc_packed, epilogue=ops.EPI_GELU | SEQ)]
        kc, vc = torch.zeros(m // Tq, H, Tq, C // H, device="cuda"), torch.zeros(m // Tq, H, Tq, C // H, device="cuda")
        out.append(ops.gemm_ln_qkv(xs, *qkv_packed, kc, vc, m // Tq, Tq, 0))
        return out + [kc, vc]

    ref = run(256)
    for m in (64, 272, 768):
        got = run(m)
        n = min(m, 256)
        for a, b in zip(ref[:4], got[:4]):
            assert torch.equal(a[:n], b[:n]), m
        for a, b in zip(ref[4:], got[4:]):
            assert torch.equal(a[:n // Tq], b[:n // Tq]), m
    want = torch.nn.functional.gelu(torch.nn.functional.layer_norm(x[:256], (C,), gamma, beta) @ w_fc.t() + b_fc)
    assert (ref[2] - want).abs().max().item() < 2e-4


@pytest.mark.parametrize("batch,groups,sample", [(16, 3, True), (5, 3, True), (24, 2, True), (16, 4, False)])
def test_grouped_token_loop_equals_per_batch(batch, groups, sample):
    """`ccvs_gpt_decode.groups`: the token loops of several batches as ONE loop over their stacked rows (weights streamed once
    per token for all of them) give each batch exactly the tokens of its own loop -- per-group Philox words, cache lengths and
    counters; 5 rows per group puts two groups into one 16-row tile."""
    from ccvs_amd.models.skip_vid_generator.models import mingpt
    torch.manual_seed(11)
    net = mingpt.GPT(vocab_size=200, block_size=400, num_blocks=25, n_layer=3, n_head=4, n_embd=256, emb_mode="temporal", shape=(4, 4)).cuda()
    for p in net.parameters():
        p.data.add_(0.05 * torch.randn_like(p))
    codes = [torch.randint(0, 200, (batch, 16), device="cuda") for _ in range(groups)]
    keys = [(0x1234567 + 977 * g, 0xabcdef01 ^ (g << 7)) for g in range(groups)]
    n_new = 150
    alone = []
    for g in range(groups):
        net.noise_key, net.row_offset, net.noise_call = keys[g], 32, 0
        alone.append(net.generate(codes[g], n_new, sample=sample, top_k=20).clone())
    net.noise_key, net.row_offset, net.noise_call = list(keys), [32] * groups, 0
    stacked = net.generate(torch.cat(codes), n_new, sample=sample, top_k=20)
    eager = None
    net.noise_call = 0
    eager = net.generate(torch.cat(codes), n_new, sample=sample, top_k=20, use_graph=False)
    net.noise_key, net.row_offset = None, 0
    for g in range(groups):
        assert torch.equal(stacked[g * batch:(g + 1) * batch], alone[g]), f"group {g} of the stacked loop differs from its own loop"
        assert torch.equal(eager[g * batch:(g + 1) * batch], alone[g])
    if sample:
        assert not torch.equal(alone[0][:, 16:], alone[1][:, 16:])


def test_decode_step_with_more_rows_than_the_weight_stream_form():
    """A graph-replayed generation with more than 256 rows and one group (a large --batch on a small configuration) runs: the
    decode step's GEMMs fall to the row-blocked form inside the launcher.  Greedy tokens equal the eager (per-op) loop's, and
    the first 8 rows equal a generation of those 8 rows alone up to the first near-tie (different GEMM forms: low bits)."""
    from ccvs_amd.models.skip_vid_generator.models import mingpt
    torch.manual_seed(3)
    net = mingpt.GPT(vocab_size=64, block_size=48, num_blocks=3, n_layer=2, n_head=2, n_embd=64, emb_mode="temporal", shape=(4, 4)).cuda()
    for p in net.parameters():
        p.data.add_(0.05 * torch.randn_like(p))
    codes = torch.randint(0, 64, (320, 16), device="cuda")
    graph = net.generate(codes, 12, sample=False)
    eager = net.generate(codes, 12, sample=False, use_graph=False)
    assert graph.shape == (320, 28) and torch.equal(graph, eager)


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
