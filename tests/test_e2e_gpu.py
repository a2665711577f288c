"""-m gpu: the whole hot path through the reference-shaped API (QVidModel / Transformer /
Generator) against the golden run of the reference (tests/golden/tiny_e2e.npz) and against
the CPU oracle on fresh seeded inputs.  Bars: VQ indices bit-exact; pixels within 1e-3 abs."""
import os

import numpy as np
import pytest
import torch

from oracle import ccvs_oracle as O

pytestmark = pytest.mark.gpu

TINY_ARGV = [
    "--name", "tiny", "--dataset", "bairhd", "--max_dim", "32", "--vid_len", "4",
    "--q_z_num", "32", "--q_z_size", "16", "--q_z_shape", "8", "8",
    "--q_use_enc", "--q_use_dec", "--q_necf", "8", "--q_necf_mult", "1", "2", "2",
    "--q_enc_model", "skipgan", "--q_dec_model", "skipgan", "--q_use_inter", "--q_inter_p", "0.75",
    "--q_skip_context", "1", "2", "3", "--q_skip_memory", "3",
    "--x_z_num", "32", "--x_z_len", "256", "--x_n_layer", "2", "--x_n_head", "2", "--x_n_embd", "32",
    "--x_z_chunk", "64", "--x_cond_len", "64", "--x_emb_mode", "temporal", "--x_num_blocks", "4",
    "--batch_size_vid", "2",
]

PIX_TOL = 1e-3


def _sd(gold, prefix):
    return {k[len(prefix) + 1:]: torch.from_numpy(gold[k]) for k in gold.files if k.startswith(prefix + "/")}


def _load(module, sd):
    missing, unexpected = module.load_state_dict(sd, strict=False)
    assert not unexpected, unexpected
    assert all(k.endswith(".kernel") for k in missing), missing  # FIR buffers are recomputed


def maxdiff(a, b):
    return (a.detach().float().cpu() - b.detach().float().cpu()).abs().max().item()


@pytest.fixture(scope="module")
def tiny(golden_dir):
    assert torch.cuda.is_available()
    from ccvs_amd.tools.options import Options
    from ccvs_amd.models.skip_vid_generator.models.quantized_video_model import QVidModel
    from ccvs_amd.models.skip_vid_generator.models.transformer_model import Transformer
    gold = np.load(os.path.join(golden_dir, "tiny_e2e.npz"))
    opt = Options().parse(load_qvid_generator=True, load_transformer=True, argv=TINY_ARGV)
    qopt, xopt = opt["qvid_generator"], opt["transformer"]
    qv = QVidModel(qopt, is_train=False, is_main=True).eval()
    tr = Transformer(xopt, is_train=False, is_main=True).eval()
    _load(qv.net_e, _sd(gold, "e"))
    _load(qv.net_q, _sd(gold, "q"))
    _load(qv.net_g, _sd(gold, "g"))
    _load(tr.net_t, _sd(gold, "t"))
    nets = {"e": _sd(gold, "e"), "q": _sd(gold, "q"), "g": _sd(gold, "g"), "t": _sd(gold, "t")}
    return dict(gold=gold, qv=qv, tr=tr, qopt=qopt, xopt=xopt, nets=nets, opt=opt)


def test_state_dict_layout(tiny):
    """The MI355X modules expose exactly the reference's parameter names and shapes."""
    for net, pre in ((tiny["qv"].net_e, "e"), (tiny["qv"].net_q, "q"), (tiny["qv"].net_g, "g"), (tiny["tr"].net_t, "t")):
        ref = _sd(tiny["gold"], pre)
        own = {k: v for k, v in net.state_dict().items() if not k.endswith(".kernel")}
        assert set(own) == set(ref), (set(own) ^ set(ref))
        assert all(tuple(own[k].shape) == tuple(ref[k].shape) for k in ref)


def test_encode_golden(tiny):
    g, qv = tiny["gold"], tiny["qv"]
    enc = qv({"vid": torch.from_numpy(g["vid"])}, mode="vid_encoder")
    assert enc["code"].dtype == torch.int64
    assert torch.equal(enc["code"].cpu(), torch.from_numpy(g["enc_code"])), "VQ token indices must be bit-exact"
    for i, f in enumerate(enc["inter"]):
        assert maxdiff(f, torch.from_numpy(g[f"enc_inter{i}"])) < 1e-4
    assert float(g["vq_min_gap"]) > 1e-4  # the fixture's argmin is well separated


def test_decoder_cond_frame_golden(tiny):
    g, qv = tiny["gold"], tiny["qv"]
    enc = qv({"vid": torch.from_numpy(g["vid"])}, mode="vid_encoder")
    rgb, _, flows, occs, _ = qv.net_g(enc["z"][:, :1].contiguous(), [[f[:, :1] for f in enc["inter"]]], return_all=True)
    assert maxdiff(rgb, torch.from_numpy(g["dec_cond_vid"])) < PIX_TOL
    for i in range(3):
        assert maxdiff(flows[i], torch.from_numpy(g[f"dec_cond_flow{i}"])) < PIX_TOL
        assert maxdiff(occs[i], torch.from_numpy(g[f"dec_cond_occ{i}"])) < PIX_TOL


def test_gpt_logits_golden(tiny):
    g, tr = tiny["gold"], tiny["tr"]
    code = torch.from_numpy(g["gen_code_greedy"]).cuda()
    logits = tr.net_t(code[:, :-1])
    assert maxdiff(logits, torch.from_numpy(g["gpt_logits"])) < 1e-4


def _audit_tokens(got, want, logits_at):
    """Token streams must agree; a divergence is only acceptable at a logit near-tie."""
    got, want = got.cpu(), want.cpu()
    if torch.equal(got, want):
        return
    b, t = (got != want).nonzero()[0].tolist()
    lg = logits_at(b, t)
    gap = (lg[got[b, t]] - lg[want[b, t]]).abs().item()
    assert gap < 1e-4, f"token mismatch at {(b, t)} with logit gap {gap:.3e}"


def _assert_pixels(gen, vid_in, out, want, extra=None):
    """Decoded pixels vs the oracle, ALWAYS: directly when the token streams agree; if a logit near-tie flipped a token
    (audited by the caller) the HIP decoder is run teacher-forced on the ORACLE's tokens instead, so the pixel bar is
    checked either way."""
    if torch.equal(out["fake"]["code"].cpu(), want["code"]):
        got = out["fake"]["vid"]
    else:
        data = {"vid": vid_in.clone()}
        data.update(extra or {})
        ws = gen.condition(data)
        got = gen.decode_codes(ws, want["code"].cuda(), None if want.get("state_code") is None else want["state_code"].cuda())["vid"]
    assert got.shape == want["vid"].shape
    assert maxdiff(got, want["vid"]) < PIX_TOL, maxdiff(got, want["vid"])


def test_generate_greedy_golden(tiny):
    g, tr, xopt = tiny["gold"], tiny["tr"], tiny["xopt"]
    xopt.sample, xopt.top_k = False, 10
    want = torch.from_numpy(g["gen_code_greedy"])
    out = tr({"code": want[:, :64].clone()}, mode="inference", total_len=256)
    ref_logits = torch.from_numpy(g["gpt_logits"])
    _audit_tokens(out["code"], want, lambda b, t: ref_logits[b, t - 1])


def test_generate_sampled_golden(tiny):
    """torch.multinomial stream of the reference reproduced from the same CPU generator state."""
    g, tr, xopt = tiny["gold"], tiny["tr"], tiny["xopt"]
    xopt.sample, xopt.top_k = True, 10
    want = torch.from_numpy(g["gen_code_sampled_seed7"])
    torch.manual_seed(7)
    tr.sample_noise, tr.generator = "host", None
    out = tr({"code": want[:, :64].clone()}, mode="inference", total_len=64 + 8)
    assert torch.equal(out["code"].cpu(), want)
    xopt.sample = False


def test_decode_loop_golden(tiny):
    g, qv = tiny["gold"], tiny["qv"]
    enc = qv({"vid": torch.from_numpy(g["vid"])}, mode="vid_encoder")
    inter = [f[:, :1].contiguous() for f in enc["inter"]]
    fake = qv({"code": torch.from_numpy(g["gen_code_greedy"]), "inter": inter}, mode="vid_decoder")
    assert fake["vid"].shape == (2, 4, 3, 32, 32)
    assert maxdiff(fake["vid"], torch.from_numpy(g["fake_vid"])) < PIX_TOL


def test_step_decoder_golden(tiny):
    g, qv = tiny["gold"], tiny["qv"]
    enc = qv({"vid": torch.from_numpy(g["vid"])}, mode="vid_encoder")
    inter = [f[:, :1].contiguous() for f in enc["inter"]]
    step = qv({"code": torch.from_numpy(g["gen_code_greedy"])[:, 64:128].contiguous(), "inter": inter}, mode="vid_step_decoder")
    assert maxdiff(step["vid"], torch.from_numpy(g["step_vid"])) < PIX_TOL
    assert torch.equal(step["code"].cpu(), torch.from_numpy(g["step_code"]))
    assert step["inter"][0].shape[1] == 2


def test_invalid_mode_raises(tiny):
    with pytest.raises(ValueError):
        tiny["qv"]({}, mode="nope")
    with pytest.raises(ValueError):
        tiny["tr"]({"code": torch.zeros(1, 64, dtype=torch.long)}, mode="nope")


def test_generator_vs_oracle_fresh_input(tiny):
    """Generator.generate_vid on a NEW seeded clip == oracle.generate_vid (greedy), end to end."""
    from ccvs_amd.helpers.generator import Generator
    gen = Generator(tiny["opt"])
    gen.vid_model, gen.transformer_model = tiny["qv"], tiny["tr"]
    tiny["xopt"].sample, tiny["xopt"].top_k = False, 10
    data = gen.synthetic_batch(2, seed=5)
    out = gen.generate_vid({"vid": data["vid"].clone()})
    trace = []
    want = O.generate_vid(tiny["nets"], tiny["qopt"], tiny["xopt"], data["vid"], trace=trace)
    assert torch.equal(out["enc_code"].cpu(), want["enc_code"])
    _audit_tokens(out["fake"]["code"], want["code"], lambda b, t: trace[t - 64][b])
    _assert_pixels(gen, data["vid"], out, want)
    # the teacher-forced "rec" pass runs by default like the reference's (helpers/generator.py:172-189)
    rec_want = O.qvid_decode(tiny["nets"], tiny["qopt"], want["enc_code"], [f[:, :1].contiguous() for f in O.qvid_encode(tiny["nets"], tiny["qopt"], data["vid"])["inter"]])
    assert out["rec"] is not None and maxdiff(out["rec"]["vid"], rec_want) < PIX_TOL


def test_p2p_vs_oracle(tiny):
    """Point-to-point (cond prefix with shifted t_emb, extra decoder context), SURVEY 8f row f1."""
    from ccvs_amd.helpers.generator import Generator
    xopt = tiny["xopt"]
    xopt.p2p, xopt.sample = True, False
    try:
        gen = Generator(tiny["opt"])
        gen.vid_model, gen.transformer_model = tiny["qv"], tiny["tr"]
        data = gen.synthetic_batch(2, seed=9)
        out = gen.generate_vid({"vid": data["vid"].clone()})
        trace = []
        want = O.generate_vid(tiny["nets"], tiny["qopt"], xopt, data["vid"], trace=trace)
        _audit_tokens(out["fake"]["code"], want["code"], lambda b, t: trace[t - 64][b])
        assert out["fake"]["vid"].shape == want["vid"].shape == (2, 4, 3, 32, 32)
        _assert_pixels(gen, data["vid"], out, want)
    finally:
        xopt.p2p = False


def test_determinism(tiny):
    """Same inputs twice -> bitwise identical clip (fixed-order reductions, no float atomics)."""
    g, qv = tiny["gold"], tiny["qv"]
    enc = qv({"vid": torch.from_numpy(g["vid"])}, mode="vid_encoder")
    inter = [f[:, :1].contiguous() for f in enc["inter"]]
    a = qv({"code": torch.from_numpy(g["gen_code_greedy"]), "inter": [f.clone() for f in inter]}, mode="vid_decoder")["vid"]
    b = qv({"code": torch.from_numpy(g["gen_code_greedy"]), "inter": [f.clone() for f in inter]}, mode="vid_decoder")["vid"]
    assert torch.equal(a, b)


def test_bair_scale_encode_decode_frame_vs_oracle():
    """Full BAIR geometry (256x256, 6 levels, 128..512 channels, 9x9 heads, stride-2 cost volumes, k = 2
    contexts): one encode + one flow-guided decoder frame through the split-bf16 convolutions against the
    fp32 CPU oracle.  VQ indices exact, pixels within the 1e-3 bar of the north star."""
    from ccvs_amd.tools.options import Options, BAIR_ARGV
    from ccvs_amd.models.skip_vid_generator.models.quantized_video_model import QVidModel
    opt = Options().parse(load_qvid_generator=True, load_transformer=True, argv=BAIR_ARGV)
    qopt = opt["qvid_generator"]
    torch.manual_seed(0)
    qv = QVidModel(qopt, is_train=False, is_main=True).eval()
    g = torch.Generator().manual_seed(1)
    vid = torch.rand(1, 2, 3, 256, 256, generator=g) * 2 - 1
    with torch.no_grad():
        z_e, _ = qv.net_e(vid.cuda())
        cb = qv.net_q.embedding.weight
        cb.copy_((torch.randn(cb.shape, generator=torch.Generator().manual_seed(4)) * float(z_e.std())).cuda())
        enc = qv({"vid": vid.clone()}, mode="vid_encoder")
        ctx = [[f[:, :1] for f in enc["inter"]], [f[:, 1:] for f in enc["inter"]]]
        rgb, _, flows, occs, _ = qv.net_g(enc["z"][:, :1].contiguous(), ctx, return_all=True, inter_pre_warping=False)
    cpu = lambda m: {k: v.detach().cpu() for k, v in m.state_dict().items()}
    nets = {"e": cpu(qv.net_e), "q": cpu(qv.net_q), "g": cpu(qv.net_g)}
    with torch.no_grad():
        want = O.qvid_encode(nets, qopt, vid)
        assert torch.equal(enc["code"].cpu(), want["code"]), "VQ indices must be bit-exact at BAIR scale"
        for a, b in zip(enc["inter"], want["inter"]):
            assert maxdiff(a, b) < 1e-4
        octx = [[f[:, :1] for f in want["inter"]], [f[:, 1:] for f in want["inter"]]]
        orgb, oflows, ooccs = O.decoder_forward(nets["g"], qopt, want["z"][:, :1].contiguous(), octx, return_all=True)
    assert maxdiff(rgb, orgb) < PIX_TOL, maxdiff(rgb, orgb)
    assert maxdiff(flows[-1], oflows[-1]) < PIX_TOL and maxdiff(occs[-1], ooccs[-1]) < PIX_TOL


def test_multi_frame_conditioning_vs_oracle(tiny):
    """Kinetics-style conditioning (BASELINE config 3 in miniature): 2 conditioning frames -> 2 predicted."""
    from ccvs_amd.helpers.generator import Generator
    xopt = tiny["xopt"]
    old = xopt.cond_len
    xopt.cond_len, xopt.sample = 128, False
    try:
        gen = Generator(tiny["opt"])
        gen.vid_model, gen.transformer_model = tiny["qv"], tiny["tr"]
        data = gen.synthetic_batch(2, seed=21)
        out = gen.generate_vid({"vid": data["vid"].clone()})
        trace = []
        want = O.generate_vid(tiny["nets"], tiny["qopt"], xopt, data["vid"], trace=trace)
        assert torch.equal(out["enc_code"].cpu(), want["enc_code"])
        _audit_tokens(out["fake"]["code"], want["code"], lambda b, t: trace[t - 128][b])
        _assert_pixels(gen, data["vid"], out, want)
    finally:
        xopt.cond_len = old


def test_sliding_token_window_vs_oracle(tiny):
    """total_len > z_len: the token window slides by one frame per fill (transformer_model.py:283-323); every slide
    restarts positions at 0, so the KV-cached engine re-prefills the window."""
    tr, xopt = tiny["tr"], tiny["xopt"]
    xopt.sample, xopt.top_k = False, 10
    code = torch.from_numpy(tiny["gold"]["gen_code_greedy"])[:, :64].clone()
    total = 256 + 2 * 64   # z_len = 256: two slides
    out = tr({"code": code.clone()}, mode="inference", total_len=total)["code"].cpu()
    trace = []
    want = O.generate_fake(tiny["nets"]["t"], xopt, code, total_len=total, trace=trace)
    assert out.shape == want.shape == (2, total)
    if not torch.equal(out, want):
        b, t = (out != want).nonzero()[0].tolist()
        lg = trace[t - 64][b]
        assert (lg[out[b, t]] - lg[want[b, t]]).abs().item() < 1e-4


TINY_STATE_ARGV = TINY_ARGV + [
    "--x_stft", "--x_state_num", "24", "--x_state_size", "2", "--x_z_len", "256", "--x_z_chunk", "66",
    "--x_top_k_state", "5", "--x_temperature_state", "0.8",
    "--a_stft_num", "24", "--a_stft_size", "16", "--a_stft_hsize", "8", "--a_stft_shape", "2", "1",
]


@pytest.fixture(scope="module")
def state_stream(golden_dir):
    from ccvs_amd.tools.options import Options
    from ccvs_amd.models.skip_vid_generator.models.transformer_model import Transformer
    from ccvs_amd.models.skip_vid_generator.models.stft_model import StftModel
    gold = np.load(os.path.join(golden_dir, "tiny_state.npz"))
    opt = Options().parse(load_qvid_generator=True, load_transformer=True, load_stft_ae=True, argv=TINY_STATE_ARGV)
    xopt, aopt = opt["transformer"], opt["stft_ae"]
    tr = Transformer(xopt, is_train=False, is_main=True).eval()
    sm = StftModel(aopt, is_train=False, is_main=True).eval()
    _load(tr.net_t, _sd(gold, "t"))
    _load(sm.net_e, _sd(gold, "ae"))
    _load(sm.net_q, _sd(gold, "aq"))
    nets = {"t": _sd(gold, "t"), "ae": _sd(gold, "ae"), "aq": _sd(gold, "aq")}
    return dict(gold=gold, tr=tr, sm=sm, xopt=xopt, aopt=aopt, nets=nets)


def test_stft_encode_golden(state_stream):
    """StftModel.encode: spectrogram frames -> ancillary tokens, bit-exact."""
    g, sm = state_stream["gold"], state_stream["sm"]
    z = sm.net_e(torch.from_numpy(g["stft"]).cuda())
    assert maxdiff(z, torch.from_numpy(g["stft_z"])) < 1e-4
    got = sm({"stft": torch.from_numpy(g["stft"])}, mode="vid_encoder")["state_code"]
    assert torch.equal(got.cpu(), torch.from_numpy(g["state_code"]))


def test_gpt_state_interleave_logits_golden(state_stream):
    """GPT.forward with the per-frame [state | frame] interleave (mingpt.py:246-282), teacher-forced."""
    g, tr = state_stream["gold"], state_stream["tr"]
    code, state = torch.from_numpy(g["tf_code"]).cuda(), torch.from_numpy(g["state_code"])[:, :8].cuda()
    got = tr.net_t(code, state_idx=state)
    want = torch.from_numpy(g["tf_logits"])
    assert got.shape == want.shape
    assert maxdiff(got, want) < 2e-4


def _audit_pair(got_c, got_s, want_c, want_s):
    assert got_c.shape == want_c.shape and got_s.shape == want_s.shape
    assert torch.equal(got_c.cpu(), want_c) and torch.equal(got_s.cpu(), want_s)


def test_generate_with_given_and_predicted_state_golden(state_stream):
    """Transformer('inference') with an ancillary stream: STFT tokens given for every frame (graph replay for the frame
    tokens, `extend` at frame boundaries), predicted along with the frames, and through the sliding window."""
    g, tr, xopt = state_stream["gold"], state_stream["tr"], state_stream["xopt"]
    xopt.sample, xopt.top_k, xopt.sample_state = False, 10, False
    code, state = torch.from_numpy(g["tf_code"])[:, :64], torch.from_numpy(g["state_code"])
    for use_graph in (True, False):
        xopt.use_graph = use_graph
        out = tr({"code": code.clone(), "state_code": state[:, :8].clone()}, mode="inference", total_len=264)
        _audit_pair(out["code"], out["state_code"], torch.from_numpy(g["given_code"]), torch.from_numpy(g["given_state"]))
        out = tr({"code": code.clone(), "state_code": state[:, :2].clone()}, mode="inference", total_len=264)
        _audit_pair(out["code"], out["state_code"], torch.from_numpy(g["pred_code"]), torch.from_numpy(g["pred_state"]))
        out = tr({"code": code.clone(), "state_code": state.clone()}, mode="inference", total_len=330)
        _audit_pair(out["code"], out["state_code"], torch.from_numpy(g["slide_code"]), torch.from_numpy(g["slide_state"]))
    xopt.use_graph = True


def test_generate_sampled_state_stream_golden(state_stream):
    """Both streams sampled: the reference's torch.multinomial stream reproduced from the same CPU generator state."""
    g, tr, xopt = state_stream["gold"], state_stream["tr"], state_stream["xopt"]
    xopt.sample, xopt.top_k, xopt.sample_state = True, 10, True
    tr.sample_noise, tr.generator = "host", None
    torch.manual_seed(7)
    code, state = torch.from_numpy(g["tf_code"])[:, :64], torch.from_numpy(g["state_code"])
    out = tr({"code": code.clone(), "state_code": state[:, :2].clone()}, mode="inference", total_len=64 + 2 + 2 + 6)
    _audit_pair(out["code"], out["state_code"], torch.from_numpy(g["samp_code"]), torch.from_numpy(g["samp_state"]))
    xopt.sample, xopt.sample_state = False, False


def test_state_front_golden(state_stream, golden_dir):
    """`--x_state_front` (every ancillary token in front of the frame tokens, mingpt.py:261-263): teacher-forced logits, and
    generation -- a full forward per pick, the sequence is re-ordered by every new ancillary token -- with given, predicted
    and sampled ancillary tokens, against the reference's outputs."""
    from ccvs_amd.tools.options import Options
    from ccvs_amd.models.skip_vid_generator.models.transformer_model import Transformer
    g = np.load(os.path.join(golden_dir, "tiny_state_front.npz"))
    base = state_stream["gold"]
    xopt = Options().parse(load_qvid_generator=True, load_transformer=True, load_stft_ae=True,
                           argv=TINY_STATE_ARGV + ["--x_state_front"])["transformer"]
    tr = Transformer(xopt, is_train=False, is_main=True).eval()
    _load(tr.net_t, _sd(base, "t"))
    code, state = torch.from_numpy(base["tf_code"]), torch.from_numpy(base["state_code"])
    got = tr.net_t(code.cuda(), state_idx=state[:, :8].cuda())
    want = torch.from_numpy(g["tf_logits"])
    assert got.shape == want.shape and maxdiff(got, want) < 2e-4
    xopt.sample, xopt.top_k, xopt.sample_state = False, 10, False
    out = tr({"code": code[:, :64].clone(), "state_code": state[:, :8].clone()}, mode="inference", total_len=2 * 64 + 4 * 2 + 20)
    _audit_pair(out["code"], out["state_code"], torch.from_numpy(g["given_code"]), torch.from_numpy(g["given_state"]))
    out = tr({"code": code[:, :64].clone(), "state_code": state[:, :2].clone()}, mode="inference", total_len=64 + 2 + 2 + 40)
    _audit_pair(out["code"], out["state_code"], torch.from_numpy(g["pred_code"]), torch.from_numpy(g["pred_state"]))
    xopt.sample, xopt.sample_state = True, True
    tr.sample_noise, tr.generator = "host", None
    torch.manual_seed(7)
    out = tr({"code": code[:, :64].clone(), "state_code": state[:, :2].clone()}, mode="inference", total_len=64 + 2 + 2 + 6)
    _audit_pair(out["code"], out["state_code"], torch.from_numpy(g["samp_code"]), torch.from_numpy(g["samp_state"]))
    tr.sample_noise = "device"      # in-kernel Philox: runs, stays in range, depends on the key
    out2 = tr({"code": code[:, :64].clone(), "state_code": state[:, :2].clone()}, mode="inference", total_len=64 + 2 + 2 + 6)
    assert out2["code"].shape == out["code"].shape and int(out2["state_code"].max()) < xopt.state_num


def test_keep_first_ring_golden(tiny, golden_dir):
    """`--q_keep_first --q_n_first 1`: slot 0 of the context ring pinned once the ring is full (quantized_video_model.py:
    896-898), 6 frames through 3 slots -- against the reference's own decode."""
    g = np.load(os.path.join(golden_dir, "tiny_keepfirst.npz"))
    qv, qopt = tiny["qv"], tiny["qopt"]
    old = (qopt.vid_len, getattr(qopt, "keep_first", False), getattr(qopt, "n_first", 1))
    qopt.vid_len, qopt.keep_first, qopt.n_first = 6, True, 1
    try:
        enc = qv({"vid": torch.from_numpy(g["vid"])}, mode="vid_encoder")
        inter = [f[:, :1].contiguous() for f in enc["inter"]]
        fake = qv({"code": torch.from_numpy(g["code"]), "inter": inter}, mode="vid_decoder")["vid"]
        assert fake.shape == (2, 6, 3, 32, 32)
        assert maxdiff(fake, torch.from_numpy(g["fake_vid"])) < PIX_TOL
    finally:
        qopt.vid_len, qopt.keep_first, qopt.n_first = old


def test_audio_conditioned_generator_vs_oracle(tiny, state_stream):
    """BASELINE config 5 in miniature: Generator.generate_vid with `--x_stft --keep_state` -- frames encoded, spectrogram
    frames tokenised, frame tokens predicted around the given STFT tokens, clip decoded -- against the oracle."""
    from ccvs_amd.helpers.generator import Generator
    from ccvs_amd.tools.options import Options
    opt = Options().parse(load_qvid_generator=True, load_transformer=True, load_stft_ae=True, argv=TINY_STATE_ARGV + ["--keep_state"])
    xopt = opt["transformer"]
    xopt.sample, xopt.top_k, xopt.sample_state = False, 10, False
    gen = Generator(opt)
    gen.vid_model, gen.transformer_model, gen.stft_model = tiny["qv"], state_stream["tr"], state_stream["sm"]
    gen.transformer_model.opt = xopt
    torch.manual_seed(31)
    data = gen.synthetic_batch(2, seed=33)
    stft = torch.rand(2, 4, 1, 16, 8) * 2 - 1
    out = gen.generate_vid({"vid": data["vid"].clone(), "stft": stft.clone()})
    nets = dict(tiny["nets"])
    nets.update(state_stream["nets"])   # "t" is the state-stream transformer
    trace = []
    want = O.generate_vid(nets, tiny["qopt"], xopt, data["vid"], trace=trace, stft=stft, aopt=state_stream["aopt"])
    assert torch.equal(out["enc_code"].cpu(), want["enc_code"])
    assert torch.equal(out["fake"]["state_code"].cpu(), want["state_code"])
    _audit_tokens(out["fake"]["code"], want["code"], lambda b, t: trace[t - 64][b])
    _assert_pixels(gen, data["vid"], out, want, extra={"stft": stft.clone()})


def test_bair_scale_batch_invariance():
    """BAIR-size networks, greedy sampling: a clip generated inside a ragged batch of 5 equals the same clip generated alone,
    bit for bit -- GEMM row tails, attention grids, conv batch strides and the context lists at full channel counts."""
    from ccvs_amd.tools.options import Options, BAIR_ARGV
    from ccvs_amd.helpers.generator import Generator
    opt = Options().parse(load_qvid_generator=True, load_transformer=True, argv=list(BAIR_ARGV) + ["--batch_size_vid", "5"])
    xopt, qopt = opt["transformer"], opt["qvid_generator"]
    xopt.vid_len = qopt.vid_len = 3
    xopt.sample = False
    torch.manual_seed(0)
    gen = Generator(opt).build_models()
    with torch.no_grad():
        t = gen.transformer_model.net_t
        t.s_emb.normal_(0, 0.02)
        t.t_emb.normal_(0, 0.02)
        vid = gen.synthetic_batch(5, seed=3)["vid"][:, :3].cuda()
        z_e, _ = gen.vid_model.net_e(vid[:2, :1])
        cb = gen.vid_model.net_q.embedding.weight
        cb.copy_(torch.randn_like(cb) * float(z_e.std()))
        full = gen.generate_vid({"vid": vid.clone()})
        one = gen.generate_vid({"vid": vid[3:4].clone()})
    assert torch.equal(one["fake"]["code"][0], full["fake"]["code"][3])
    assert torch.equal(one["fake"]["vid"][0], full["fake"]["vid"][3])
