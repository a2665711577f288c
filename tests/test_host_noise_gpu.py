"""-m gpu: reference-seed sampling (`--x_sample_noise host`: the Exp(1) stream torch.multinomial draws from the process
generator, transformer_model.py:395-409) on the FAST schedule -- inside the captured decode step
(`ccvs_gpt_decode.noise_stream`), inside token groups, and with several batches in flight -- must give, token for token, what
the eager one-batch-at-a-time loop gives, and what the oracle gives under the same seed."""
import pytest
import torch

from oracle import ccvs_oracle as O
from tests.test_e2e_gpu import tiny, TINY_ARGV, maxdiff, PIX_TOL  # noqa: F401  (fixture)

pytestmark = pytest.mark.gpu


def _drawer(seed):
    g = torch.Generator().manual_seed(seed)
    return lambda nb, nv: torch.empty(nb, nv, dtype=torch.float32).exponential_(1, generator=g)


@pytest.mark.parametrize("batch,groups", [(16, 3), (5, 3), (24, 2)])
def test_host_noise_in_the_captured_step_and_in_row_groups(batch, groups):
    """One batch: graph replay with the call's noise stream resident == the eager loop that uploads one block per step.  Several
    batches stacked into one loop, each reading its own pre-drawn stream == each batch's own loop (5 rows per group: two groups
    inside one 16-row tile).  Eager steps fed from pre-drawn streams agree too."""
    from ccvs_amd.models.skip_vid_generator.models import mingpt
    torch.manual_seed(11)
    net = mingpt.GPT(vocab_size=200, block_size=400, num_blocks=25, n_layer=3, n_head=4, n_embd=256, emb_mode="temporal", shape=(4, 4)).cuda()
    for p in net.parameters():
        p.data.add_(0.05 * torch.randn_like(p))
    codes = [torch.randint(0, 200, (batch, 16), device="cuda") for _ in range(groups)]
    n_new = 150
    alone, eager = [], []
    for g in range(groups):
        alone.append(net.generate(codes[g], n_new, sample=True, top_k=20, noise="host", host_noise=_drawer(100 + g)).clone())
        eager.append(net.generate(codes[g], n_new, sample=True, top_k=20, noise="host", host_noise=_drawer(100 + g), use_graph=False).clone())
        assert torch.equal(alone[g], eager[g]), f"graph replay with a resident noise stream differs from the eager loop (batch {g})"
    assert not torch.equal(alone[0][:, 16:], alone[1][:, 16:])

    def streams():
        out = []
        for g in range(groups):
            draw = _drawer(100 + g)
            out.append(torch.stack([draw(batch, 200) for _ in range(n_new)]).cuda())
        return out

    def forbidden(nb, nv):
        raise AssertionError("a stacked loop must not draw for itself")

    net.noise_key, net.row_offset = [(0, 0)] * groups, [0] * groups        # `groups` row groups
    try:
        net.noise_streams = streams()
        stacked = net.generate(torch.cat(codes), n_new, sample=True, top_k=20, noise="host", host_noise=forbidden)
        assert net.noise_streams is None
        net.noise_streams = streams()
        stacked_eager = net.generate(torch.cat(codes), n_new, sample=True, top_k=20, noise="host", host_noise=forbidden, use_graph=False)
    finally:
        net.noise_key, net.row_offset, net.noise_streams = None, 0, None
    for g in range(groups):
        assert torch.equal(stacked[g * batch:(g + 1) * batch], alone[g]), f"group {g} of the stacked loop differs from its own loop"
        assert torch.equal(stacked_eager[g * batch:(g + 1) * batch], alone[g])


def test_warm_only_call_draws_nothing():
    """Capturing the decode step up front (`GPT.warm_only`, what `run_pipelined` does before its workers start) consumes nothing
    of the host generator: the tokens of the next real call are those of a cold call."""
    from ccvs_amd.models.skip_vid_generator.models import mingpt
    torch.manual_seed(5)
    net = mingpt.GPT(vocab_size=64, block_size=96, num_blocks=6, n_layer=2, n_head=2, n_embd=64, emb_mode="temporal", shape=(4, 4)).cuda()
    code = torch.randint(0, 64, (4, 16), device="cuda")
    want = net.generate(code, 40, sample=True, top_k=8, noise="host", host_noise=_drawer(3), use_graph=False).clone()
    net.drop_engine_state()
    calls = []

    def counting(nb, nv):
        calls.append(1)
        return _drawer(0)(nb, nv)

    net.warm_only = True
    net.generate(code, 40, sample=True, top_k=8, noise="host", host_noise=counting)
    net.warm_only = False
    assert not calls
    got = net.generate(code, 40, sample=True, top_k=8, noise="host", host_noise=_drawer(3))
    assert torch.equal(got, want)


@pytest.mark.parametrize("lanes,chains,dec_streams", [(2, 2, 2), (3, 1, 1), (1, 2, 2)])
def test_pipelined_host_noise_equals_serial_and_oracle(tiny, monkeypatch, lanes, chains, dec_streams):
    """Several batches in flight, token groups x chains, host-drawn noise pre-drawn by the noise thread in batch order ==
    generate_vid batch after batch under the same `torch.manual_seed` == the ORACLE's sampled run (its torch.multinomial on the
    same generator): tokens identical, pixels within 1e-3."""
    from ccvs_amd.helpers.generator import Generator
    monkeypatch.setenv("CCVS_PIPELINE_DEC_STREAMS", str(dec_streams))
    xopt, qopt = tiny["xopt"], tiny["qopt"]
    xopt.sample, xopt.top_k, xopt.rec_pass = True, 10, False
    old = (tiny["tr"].sample_noise, tiny["tr"].generator)
    tiny["tr"].sample_noise, tiny["tr"].generator = "host", None
    try:
        gen = Generator(tiny["opt"])
        gen.vid_model, gen.transformer_model = tiny["qv"], tiny["tr"]
        batches = [gen.synthetic_batch(2, seed=70 + i)["vid"] for i in range(5)]
        torch.manual_seed(321)
        serial = [gen.generate_vid({"vid": b.clone()}, global_iter=i, schedule="serial") for i, b in enumerate(batches)]
        torch.manual_seed(321)
        res = gen.run_pipelined(({"vid": b.clone()} for b in batches), lanes=lanes, chains=chains)
        torch.cuda.synchronize()
        assert gen._token_group_size(2, lanes) == lanes
        assert [g for g, _ in gen.pipeline_token_groups()] == ([2, 2, 1] if lanes == 2 else [3, 2] if lanes == 3 else [1] * 5)
        for want, got in zip(serial, res):
            assert torch.equal(got["fake"]["code"], want["fake"]["code"]), "pipelined host-noise tokens differ from the serial schedule"
            assert torch.equal(got["fake"]["vid"], want["fake"]["vid"])
        assert not torch.equal(res[0]["fake"]["code"], res[1]["fake"]["code"])
        torch.manual_seed(321)    # the oracle consumes the same generator: one multinomial per token, batch after batch
        for i, b in enumerate(batches[:3]):
            want = O.generate_vid(tiny["nets"], qopt, xopt, b)
            assert torch.equal(res[i]["fake"]["code"].cpu(), want["code"]), f"batch {i}: sampled tokens differ from the oracle's seeded stream"
            assert maxdiff(res[i]["fake"]["vid"], want["vid"]) < PIX_TOL
    finally:
        xopt.sample, xopt.rec_pass = False, True
        tiny["tr"].sample_noise, tiny["tr"].generator = old


def test_cold_packed_weights_with_two_decode_streams(tiny, monkeypatch):
    """the decoder's kernel-ready weights are plain Python caches filled by pack kernels on first use.  With the
    decode of batch i on decode stream i % 2, a cold cache used to be packed on stream 0 and read on stream 1 with nothing
    ordering the two.  `run_pipelined` now packs everything up front on the encode stream and orders every stream behind it:
    a run whose FIRST decoder use is pipelined, on cold caches, equals the serial schedule."""
    from ccvs_amd.helpers.generator import Generator
    monkeypatch.setenv("CCVS_PIPELINE_DEC_STREAMS", "2")
    xopt = tiny["xopt"]
    xopt.sample, xopt.top_k, xopt.rec_pass = False, 10, False
    try:
        gen = Generator(tiny["opt"])
        gen.vid_model, gen.transformer_model = tiny["qv"], tiny["tr"]
        batches = [gen.synthetic_batch(2, seed=40 + i)["vid"] for i in range(4)]
        serial = [gen.generate_vid({"vid": b.clone()}, global_iter=i, schedule="serial") for i, b in enumerate(batches)]
        torch.cuda.synchronize()
        cleared = 0
        for mod in gen.vid_model.modules():          # start cold: every lazily built form is dropped
            for attr in ("_packed", "_sub0", "_projw", "_up_w"):
                if getattr(mod, attr, None) is not None:
                    setattr(mod, attr, None)
                    cleared += 1
            for heads in ("_m_heads", "_s_heads"):
                if hasattr(mod, heads):
                    getattr(mod, heads)._cache = None
                    cleared += 1
        assert cleared > 20
        res = gen.run_pipelined(({"vid": b.clone()} for b in batches), lanes=2, chains=2)
        torch.cuda.synchronize()
        for want, got in zip(serial, res):
            assert torch.equal(got["fake"]["vid"], want["fake"]["vid"])
    finally:
        xopt.rec_pass = True


def test_pipelined_without_the_flow_guided_decoder(monkeypatch):
    """a configuration without `--q_use_inter` decodes a clip in one plain decoder call
    (quantized_video_model.py:849-853); the pipelined schedule hands it over as one piece instead of raising."""
    from ccvs_amd.tools.options import Options
    from ccvs_amd.helpers.generator import Generator
    argv = [a for a in TINY_ARGV if a != "--q_use_inter"] + ["--rec_pass", "false", "--x_sample_noise", "device"]
    torch.manual_seed(2)
    gen = Generator(Options().parse(load_qvid_generator=True, load_transformer=True, argv=argv)).build_models()
    with torch.no_grad():
        cb = gen.vid_model.net_q.embedding.weight
        cb.copy_(torch.randn(cb.shape, generator=torch.Generator().manual_seed(4)).cuda())
    batches = [gen.synthetic_batch(2, seed=20 + i)["vid"] for i in range(3)]
    serial = [gen.generate_vid({"vid": b.clone()}, global_iter=i, schedule="serial") for i, b in enumerate(batches)]
    res = gen.run_pipelined(({"vid": b.clone()} for b in batches), lanes=2, chains=1)
    torch.cuda.synchronize()
    for want, got in zip(serial, res):
        assert got["fake"]["vid"].shape == (2, 4, 3, 32, 32)
        assert torch.equal(got["fake"]["code"], want["fake"]["code"]) and torch.equal(got["fake"]["vid"], want["fake"]["vid"])
    assert all(len(ev["segs"]) == 1 for ev in gen._pipeline_events)


def test_run_entry_point_pipelined_equals_serial(tmp_path, monkeypatch):
    """`Generator(opt).run()` -- what every scripts/*/save_videos*.sh calls (helpers/generator.py:248-282) -- through the real
    option line: the several-batches-in-flight schedule writes, file for file, the bytes of the one-batch-at-a-time loop (real /
    fake / rec clips of every batch, the rec pass included), under the reference's process seed (host noise)."""
    import os
    import numpy as np
    from ccvs_amd.tools.options import Options
    from ccvs_amd.helpers.generator import Generator
    argv = TINY_ARGV + ["--n_iter", "5", "--x_top_k", "10", "--x_sample"]
    outs = {}
    for mode in ("serial", "pipelined"):
        monkeypatch.setenv("CCVS_RUN_SCHEDULE", mode)
        torch.manual_seed(0)
        opt = Options().parse(load_qvid_generator=True, load_transformer=True, argv=argv + ["--save_path", str(tmp_path / mode)])
        assert opt["transformer"].sample and getattr(opt["transformer"], "rec_pass", True)
        gen = Generator(opt)
        torch.manual_seed(9)
        last = gen.run()
        files = {}
        root = opt["transformer"].result_path
        for kind in ("real", "fake", "rec"):
            names = sorted(os.listdir(os.path.join(root, kind)))
            assert len(names) == 10, (kind, names)                  # 5 batches x 2 clips
            for n in names:
                path = os.path.join(root, kind, n)
                files[(kind, n)] = np.load(path) if n.endswith(".npy") else open(path, "rb").read()
        outs[mode] = (last.cpu(), files, getattr(gen, "last_lanes", 0))
    assert outs["pipelined"][2] >= 1
    assert torch.equal(outs["serial"][0], outs["pipelined"][0])
    assert outs["serial"][1].keys() == outs["pipelined"][1].keys()
    for key, a in outs["serial"][1].items():
        b = outs["pipelined"][1][key]
        assert (np.array_equal(a, b) if isinstance(a, np.ndarray) else a == b), key
    fakes = [v for (kind, _), v in outs["pipelined"][1].items() if kind == "fake"]
    assert isinstance(fakes[0], np.ndarray) and not np.array_equal(fakes[0], fakes[2])


def test_sampler_switch_between_runs_recaptures_up_front(tiny, monkeypatch):
    """bench.py's legs run the same Generator with host noise, device noise and host noise again.  The captured decode steps belong to
    ONE sampler: after a run with the other one the up-front warm-up must see the chain as cold again (a capture left to a worker
    thread would race with the other threads' launches), and the third run still equals the first."""
    from ccvs_amd.helpers.generator import Generator
    from ccvs_amd.helpers.pipeline import PipelinedRun
    monkeypatch.setenv("CCVS_PIPELINE_DEC_STREAMS", "2")
    xopt = tiny["xopt"]
    xopt.sample, xopt.top_k, xopt.rec_pass = True, 10, False
    old = (tiny["tr"].sample_noise, tiny["tr"].generator)
    try:
        gen = Generator(tiny["opt"])
        gen.vid_model, gen.transformer_model = tiny["qv"], tiny["tr"]
        batches = [gen.synthetic_batch(2, seed=30 + i)["vid"] for i in range(4)]
        runs = []
        for noise in ("host", "device", "host"):
            for tr_ in [gen.transformer_model] + [t for t, _ in gen._chains]:
                tr_.sample_noise, tr_.generator = noise, None
            probe = PipelinedRun(gen, iter([{"vid": batches[0].clone()}]), lanes=2, chains=2)
            if runs:
                assert probe._is_cold(0, 2, 2), f"chain 0 still looks warm after the sampler changed to {noise}"
            torch.manual_seed(5)
            runs.append(gen.run_pipelined(({"vid": b.clone()} for b in batches), lanes=2, chains=2))
            torch.cuda.synchronize()
            assert not PipelinedRun(gen, iter([{"vid": batches[0].clone()}]), lanes=2, chains=2)._is_cold(0, 2, 2)
        for a, b in zip(runs[0], runs[2]):
            assert torch.equal(a["fake"]["code"], b["fake"]["code"]) and torch.equal(a["fake"]["vid"], b["fake"]["vid"])
        assert not torch.equal(runs[0][0]["fake"]["code"], runs[1][0]["fake"]["code"])
    finally:
        xopt.sample, xopt.rec_pass = False, True
        for tr_ in [tiny["tr"]] + [t for t, _ in gen._chains]:
            tr_.sample_noise, tr_.generator = old
