"""-m gpu: parity at the REAL geometries of BASELINE.json configs[3] (BAIR point-to-point) and configs[4] (AudioSet-Drums,
audio-conditioned) against the CPU oracle -- the flags are those of scripts/bairhd/save_videos_p2p.sh and
scripts/drums/save_videos_audio_on.sh (`BAIR_P2P_ARGV`, `DRUMS_ARGV` in ccvs_amd/tools/options.py).

  * Drums: StftModel.encode on [1, T, 1, 64, 16] spectrogram frames with 512-channel hidden / code sizes -> 8 x 2 tokens per
    frame, bit-exact; the 24-layer GPT with the 16-token ancillary interleave at the full 1280-token window, teacher-forced
    and through the KV cache after one slide (re-prefill of ~1200 tokens + decode steps); the 128x128 / 5-level decoder for
    18 frames through the 15-slot context ring with the first 8 slots pinned (`--q_keep_first --q_n_first 8`);
  * p2p: one 256x256 decoder frame with k ring contexts PLUS the end frame's `cond_inter` context
    (quantized_video_model.py:868-873), and the Generator's crop bookkeeping for it.

Every comparison is teacher-forced (given tokens), so nothing depends on a sampled token; the slow ones (tens of seconds
of CPU oracle) can be skipped with CCVS_SKIP_SLOW=1.
"""
import os
import time

import pytest
import torch

from oracle import ccvs_oracle as O

pytestmark = pytest.mark.gpu

PIX_TOL = 1e-3
slow = pytest.mark.skipif(os.environ.get("CCVS_SKIP_SLOW", "0") == "1", reason="CCVS_SKIP_SLOW=1")


def maxdiff(a, b):
    return (a.detach().float().cpu() - b.detach().float().cpu()).abs().max().item()


def cpu_sd(m):
    return {k: v.detach().cpu() for k, v in m.state_dict().items()}


def _calibrate(net_e, net_q, x, seed=4):
    """codebook = randn * std(z_e): the default init (+-1/n_e) makes every argmin a near-tie (hard part 2)."""
    with torch.no_grad():
        out = net_e(x.cuda())
        z_e = out[0] if isinstance(out, (tuple, list)) else out
        cb = net_q.embedding.weight
        cb.copy_((torch.randn(cb.shape, generator=torch.Generator().manual_seed(seed)) * float(z_e.std())).cuda())


@pytest.fixture(scope="module")
def drums():
    from ccvs_amd.tools.options import Options, DRUMS_ARGV
    opt = Options().parse(load_qvid_generator=True, load_transformer=True, load_stft_ae=True, argv=list(DRUMS_ARGV))
    return opt


def test_drums_options_are_the_reference_launch_line(drums):
    x, q, a = drums["transformer"], drums["qvid_generator"], drums["stft_ae"]
    assert (x.vid_len, x.cond_len, x.z_len, x.z_chunk, x.state_size, x.state_num, x.num_blocks) == (45, 960, 1280, 80, 16, 1024, 16)
    assert x.stft and x.keep_state and q.keep_first and q.n_first == 8 and q.skip_memory == 15 and q.necf_mult == [1, 1, 2, 2, 4]
    assert (a.stft_num, a.stft_size, a.stft_hsize, a.stft_shape) == (1024, 512, 512, [8, 2]) and q.max_dim == 128 and q.aspect_ratio == 1


def test_drums_stft_encode_real_geometry(drums):
    """StftEncoder 1 -> 512 -> ... -> 512 on 64 x 16 spectrogram frames (three blur + stride-2 convolutions down to 8 x 2) and
    the 1024 x 512 quantiser: z_e close, the 16 tokens per frame bit-exact."""
    from ccvs_amd.models.skip_vid_generator.models.stft_model import StftModel
    aopt = drums["stft_ae"]
    torch.manual_seed(0)
    sm = StftModel(aopt, is_train=False, is_main=True).eval()
    stft = torch.rand(1, 6, 1, 64, 16, generator=torch.Generator().manual_seed(2)) * 2 - 1
    _calibrate(sm.net_e, sm.net_q, stft[:, :2])
    nets = {"ae": cpu_sd(sm.net_e), "aq": cpu_sd(sm.net_q)}
    got = sm({"stft": stft.clone()}, mode="vid_encoder")["state_code"]
    with torch.no_grad():
        want = O.stft_encode(nets, aopt, stft)
    assert got.shape == want.shape == (1, 6 * 16)
    assert torch.equal(got.cpu(), want), "STFT token indices must be bit-exact"


@pytest.fixture(scope="module")
def drums_gpt():
    from ccvs_amd.models.skip_vid_generator.models.mingpt import GPT
    torch.manual_seed(0)
    net = GPT(vocab_size=1024, block_size=1280, num_blocks=16, n_layer=24, n_head=16, n_embd=1024, emb_mode="temporal", shape=[8, 8],
              state_vocab_size=1024, state_size=16).cuda().eval()
    g = torch.Generator().manual_seed(3)
    with torch.no_grad():
        for p_ in (net.s_emb, net.t_emb, net.state_s_emb):
            p_.copy_(torch.randn(p_.shape, generator=g) * 0.02)
        for blk in net.blocks:       # non-trivial LayerNorm parameters and biases: the folded-LN algebra must hold for them
            for ln in (blk.ln1, blk.ln2):
                ln.weight.copy_(1 + 0.1 * torch.randn(1024, generator=g))
                ln.bias.copy_(0.05 * torch.randn(1024, generator=g))
    cfg = O.namespace(z_shape=[8, 8], emb_mode="temporal", n_layer=24, n_head=16, z_len=1280, num_blocks=16, state_size=16, state_front=False)
    return net, cpu_sd(net), cfg


@slow
def test_drums_gpt_full_window_interleave_vs_oracle(drums_gpt):
    """The whole 1280-token window of the Drums transformer: 16 frames x (16 STFT tokens + 64 frame tokens), teacher-forced,
    B = 1 -- logits of every position against the oracle's merge (mingpt.py:246-282)."""
    net, sd, cfg = drums_gpt
    g = torch.Generator().manual_seed(4)
    code = torch.randint(0, 1024, (1, 16 * 64), generator=g)
    state = torch.randint(0, 1024, (1, 16 * 16), generator=g)
    got = net(code.cuda(), state_idx=state.cuda())
    with torch.no_grad():
        want = O.gpt_forward(sd, cfg, code, state_idx=state)
    assert got.shape == want.shape == (1, 1280, 1024)
    d = maxdiff(got, want)
    print(f"\nDrums GPT, 1280-token window: max|logit diff| = {d:.2e} (logit std {want.std().item():.3f})")
    assert d < 2e-4, d


@slow
def test_drums_gpt_slide_reprefill_and_decode_steps_vs_oracle(drums_gpt):
    """One slide of the token window (transformer_model.py:301-326): the window restarts one frame later, so the kept 15
    frames (1200 merged tokens) are RE-PREFILLED at positions 0.., then frame tokens are appended one by one.  KV-cached
    engine: prefill of 1200 + 16 given STFT tokens + 2 frame tokens, then three decode steps == rows of the oracle's
    teacher-forced forward over the same window."""
    net, sd, cfg = drums_gpt
    g = torch.Generator().manual_seed(5)
    code = torch.randint(0, 1024, (1, 17 * 64), generator=g)        # 17 frames known so far
    state = torch.randint(0, 1024, (1, 17 * 16), generator=g)       # --keep_state: the audio tokens of every frame are given
    win_code, win_state = code[:, 64:64 + 15 * 64 + 5], state[:, 16:]    # the slid window: frames 1..15 whole, frame 16 begun (5 tokens)
    with torch.no_grad():
        want = O.gpt_forward(sd, cfg, win_code, state_idx=win_state)   # [1, 15*80 + 16 + 5, V]
    assert want.shape[1] == 1200 + 16 + 5
    rows = net._stream_rows(win_code.cuda(), win_state.cuda())
    assert rows.shape == (1, 1221)
    net.begin(1, 1280, stream=True)
    lg = net.prefill(rows[:, :1218].contiguous())                       # 1200 re-prefilled + 16 STFT + 2 frame tokens
    assert maxdiff(lg, want[:, 1217]) < 2e-4
    for t in range(1218, 1221):
        lg = net.extend(rows[:, t:t + 1].contiguous())
        assert maxdiff(lg, want[:, t]) < 2e-4, t


@slow
def test_drums_decoder_18_frames_keep_first_ring_vs_oracle(drums):
    """128 x 128 / 5 levels: 15 conditioning frames fill the 15-slot context ring, then 3 frames are decoded recurrently with
    the first 8 slots PINNED (`--q_keep_first --q_n_first 8`, quantized_video_model.py:896-898) -- every synthesized frame is
    re-encoded into the ring behind the pinned ones.  Teacher-forced tokens, B = 1, per-frame pixel error vs the oracle."""
    from ccvs_amd.models.skip_vid_generator.models.quantized_video_model import QVidModel
    qopt = drums["qvid_generator"]
    old = qopt.vid_len
    qopt.vid_len = 18
    try:
        torch.manual_seed(0)
        qv = QVidModel(qopt, is_train=False, is_main=True).eval()
        assert len(qopt.necf_mult) == 5
        vid = torch.rand(1, 15, 3, 128, 128, generator=torch.Generator().manual_seed(1)) * 2 - 1
        _calibrate(qv.net_e, qv.net_q, vid[:, :2])
        nets = {"e": cpu_sd(qv.net_e), "q": cpu_sd(qv.net_q), "g": cpu_sd(qv.net_g)}
        enc = qv({"vid": vid.clone()}, mode="vid_encoder")
        t0 = time.time()
        with torch.no_grad():
            want_enc = O.qvid_encode(nets, qopt, vid)
            assert torch.equal(enc["code"].cpu(), want_enc["code"])
            code = torch.randint(0, 1024, (1, 18 * 64), generator=torch.Generator().manual_seed(9))
            code[:, :15 * 64] = want_enc["code"]
            fake = qv({"code": code.clone(), "inter": [f.contiguous() for f in enc["inter"]]}, mode="vid_decoder")["vid"]
            want = O.qvid_decode(nets, qopt, code, [f.contiguous() for f in want_enc["inter"]])
        assert fake.shape == want.shape == (1, 18, 3, 128, 128)
        per_frame = [maxdiff(fake[:, t], want[:, t]) for t in range(18)]
        print(f"\nDrums 15->3 teacher-forced decode, keep_first n_first=8 ({time.time() - t0:.0f}s of oracle), max|pixel diff| per frame:",
              " ".join(f"{d:.1e}" for d in per_frame))
        assert max(per_frame) < PIX_TOL, per_frame
    finally:
        qopt.vid_len = old


@slow
def test_bair_p2p_decoder_frame_with_cond_inter_vs_oracle():
    """configs[3] at BAIR geometry: a 256 x 256 frame decoded from k = 2 ring contexts PLUS the end frame's skip features as
    one more context (`cond_inter`, quantized_video_model.py:868-873): `QVidModel.decode` over start frame + 2 synthesized
    frames + end-frame context, teacher-forced, against the oracle."""
    from ccvs_amd.tools.options import Options, BAIR_P2P_ARGV
    from ccvs_amd.models.skip_vid_generator.models.quantized_video_model import QVidModel
    opt = Options().parse(load_qvid_generator=True, load_transformer=True, argv=list(BAIR_P2P_ARGV))
    qopt, xopt = opt["qvid_generator"], opt["transformer"]
    assert xopt.p2p and xopt.vid_len == 16
    old = qopt.vid_len
    qopt.vid_len = 4                      # start frame, two interpolated frames, (end frame: context only, not decoded)
    try:
        torch.manual_seed(0)
        qv = QVidModel(qopt, is_train=False, is_main=True).eval()
        g = torch.Generator().manual_seed(1)
        vid = torch.rand(1, 2, 3, 256, 256, generator=g) * 2 - 1      # start and end frame
        _calibrate(qv.net_e, qv.net_q, vid)
        nets = {"e": cpu_sd(qv.net_e), "q": cpu_sd(qv.net_q), "g": cpu_sd(qv.net_g)}
        enc = qv({"vid": vid.clone()}, mode="vid_encoder")
        with torch.no_grad():
            want_enc = O.qvid_encode(nets, qopt, vid)
        assert torch.equal(enc["code"].cpu(), want_enc["code"])
        code = torch.randint(0, 1024, (1, 3 * 64), generator=torch.Generator().manual_seed(9))
        code[:, :64] = want_enc["code"][:, :64]
        inter = [f[:, :1].contiguous() for f in enc["inter"]]
        cond_inter = [f[:, -1:].contiguous() for f in enc["inter"]]
        fake = qv({"code": code.clone(), "inter": inter, "cond_inter": cond_inter}, mode="vid_decoder")["vid"]
        with torch.no_grad():
            want = O.qvid_decode(nets, qopt, code, [f[:, :1].contiguous() for f in want_enc["inter"]],
                                 cond_inter=[f[:, -1:].contiguous() for f in want_enc["inter"]])
        assert fake.shape == want.shape == (1, 3, 3, 256, 256)
        per_frame = [maxdiff(fake[:, t], want[:, t]) for t in range(3)]
        print("\nBAIR p2p decode (k ring contexts + cond_inter), max|pixel diff| per frame:", " ".join(f"{d:.1e}" for d in per_frame))
        assert max(per_frame) < PIX_TOL, per_frame
    finally:
        qopt.vid_len = old
