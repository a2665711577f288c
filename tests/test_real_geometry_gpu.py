"""-m gpu: parity at the REAL geometries of BASELINE.json (not miniatures) against the CPU oracle.

  * Kinetics-600 (configs[2]): 16384-code vector quantiser and 1024 -> 16384 head + top-k pick, a 64x64 / 4-level
    decoder frame with 8 contexts, and a teacher-forced 5 -> 11 clip through the 8-slot context ring;
  * BAIR (configs[1]): a teacher-forced 1 -> 15 decode of one 256x256 clip through the split-bf16 convolutions, with the
    per-frame error growth printed (the recurrence feeds every synthesized frame back through the encoder);
  * the full-size transformer (24 layers x 1024 x 16 heads): teacher-forced logits with and without the p2p prefix.

The slow ones (minutes of CPU oracle time) can be skipped with CCVS_SKIP_SLOW=1; they run by default.
"""
import os
import time

import pytest
import torch

from oracle import ccvs_oracle as O

pytestmark = pytest.mark.gpu

PIX_TOL = 1e-3   # north star: decoded pixels within 1e-3 abs of the fp32 CPU path
slow = pytest.mark.skipif(os.environ.get("CCVS_SKIP_SLOW", "0") == "1", reason="CCVS_SKIP_SLOW=1")


def maxdiff(a, b):
    return (a.detach().float().cpu() - b.detach().float().cpu()).abs().max().item()


def cpu_sd(m):
    return {k: v.detach().cpu() for k, v in m.state_dict().items()}


def _calibrate(qv, frames, seed=4):
    """The documented synthetic codebook randn * std(z_e) (hard part 2): default-init codebooks
    (+-1/n_e) make every argmin a near-tie."""
    with torch.no_grad():
        z_e, _ = qv.net_e(frames.cuda())
        cb = qv.net_q.embedding.weight
        cb.copy_((torch.randn(cb.shape, generator=torch.Generator().manual_seed(seed)) * float(z_e.std())).cuda())


@pytest.fixture(scope="module")
def kinetics():
    from ccvs_amd.tools.options import Options, KINETICS_ARGV
    from ccvs_amd.models.skip_vid_generator.models.quantized_video_model import QVidModel
    opt = Options().parse(load_qvid_generator=True, load_transformer=True, argv=list(KINETICS_ARGV))
    qopt = opt["qvid_generator"]
    torch.manual_seed(0)
    qv = QVidModel(qopt, is_train=False, is_main=True).eval()
    g = torch.Generator().manual_seed(1)
    vid = torch.rand(2, 16, 3, 64, 64, generator=g) * 2 - 1
    _calibrate(qv, vid[:, :2])
    nets = {"e": cpu_sd(qv.net_e), "q": cpu_sd(qv.net_q), "g": cpu_sd(qv.net_g)}
    return dict(qv=qv, qopt=qopt, xopt=opt["transformer"], vid=vid, nets=nets)


def test_kinetics_vq_16384_codes_bit_exact(kinetics):
    """VectorQuantizer at n_e = 16384 x 512 (Kinetics codebook, 32 MB): indices of 32 frames x 64 positions bit-exact."""
    qv, qopt, vid, nets = kinetics["qv"], kinetics["qopt"], kinetics["vid"], kinetics["nets"]
    assert qv.net_q.embedding.weight.shape == (16384, 512)
    enc = qv({"vid": vid.clone()}, mode="vid_encoder")
    with torch.no_grad():
        want = O.qvid_encode(nets, qopt, vid)
    assert enc["code"].shape == (2, 16 * 64)
    assert torch.equal(enc["code"].cpu(), want["code"]), "VQ indices must be bit-exact at n_e = 16384"
    for a, b in zip(enc["inter"], want["inter"]):
        assert maxdiff(a, b) < 1e-4
    # argmin on the oracle's own z_e too (the kernel alone, free of conv round-off): exact, ties to the lowest index
    z = torch.randn(8, 512, 8, 8, generator=torch.Generator().manual_seed(2)) * float(nets["q"]["embedding.weight"].std())
    z[0, :, 0, 0] = nets["q"]["embedding.weight"][777]          # an exact hit
    got = qv.net_q.indices(z.cuda()).cpu()
    assert torch.equal(got, O.vq_indices(z, nets["q"]["embedding.weight"]))
    assert got[0] == 777


def test_kinetics_head_and_pick_v16384():
    """ln_f + head 1024 -> 16384 (the folded-LayerNorm GEMM) and get_icode at V = 16384 (66 KB LDS pick): logits vs torch
    fp32, greedy pick and the seeded torch.multinomial stream vs the oracle."""
    from ccvs_amd import ops
    g = torch.Generator().manual_seed(5)
    B, C, V = 64, 1024, 16384
    x = torch.randn(B, C, generator=g)
    w = torch.randn(V, C, generator=g) * 0.02
    gamma, beta = 1 + 0.1 * torch.randn(C, generator=g), 0.1 * torch.randn(C, generator=g)
    want_logits = torch.nn.functional.linear(torch.nn.functional.layer_norm(x, (C,), gamma, beta), w)
    packed = ops.pack_ln_linear(w.cuda(), None, gamma.cuda(), beta.cuda())
    logits = ops.gemm_ln(x.cuda(), *packed)
    assert maxdiff(logits, want_logits) < 2e-4
    # the pick on the ORACLE's logits (so a last-bit logit difference cannot flip a choice): top-k 100, T = 1
    lg = want_logits.cuda().contiguous()
    for temperature, top_k in ((1.0, 100), (0.7, 7), (1.0, None)):
        greedy = ops.sample_topk(lg, top_k, temperature).cpu()
        assert torch.equal(greedy, O.get_icode(want_logits[:, None], temperature, top_k, False)[0][:, 0])
        torch.manual_seed(11)
        want = O.get_icode(want_logits[:, None], temperature, top_k, True)[0][:, 0]
        torch.manual_seed(11)
        noise = torch.empty(B, V).exponential_(1)        # the stream torch.multinomial consumes
        got = ops.sample_topk(lg, top_k, temperature, noise=noise.cuda()).cpu()
        assert torch.equal(got, want), (temperature, top_k)


def test_kinetics_decoder_frame_k8_vs_oracle(kinetics):
    """One 64x64 / 4-level decoder frame with all 8 contexts of the Kinetics ring (skip_memory 8), B = 2."""
    qv, qopt, vid, nets = kinetics["qv"], kinetics["qopt"], kinetics["vid"], kinetics["nets"]
    enc = qv({"vid": vid[:, :9].clone()}, mode="vid_encoder")
    ctx = [[f[:, j:j + 1] for f in enc["inter"]] for j in range(8)]
    rgb, _, flows, occs, _ = qv.net_g(enc["z"][:, 8:9].contiguous(), ctx, return_all=True, inter_pre_warping=False)
    with torch.no_grad():
        want = O.qvid_encode(nets, qopt, vid[:, :9])
        octx = [[f[:, j:j + 1] for f in want["inter"]] for j in range(8)]
        orgb, oflows, ooccs = O.decoder_forward(nets["g"], qopt, want["z"][:, 8:9].contiguous(), octx, return_all=True)
    assert rgb.shape == (2, 1, 3, 64, 64)
    assert maxdiff(rgb, orgb) < PIX_TOL, maxdiff(rgb, orgb)
    for a, b in zip(flows, oflows):
        assert maxdiff(a, b) < PIX_TOL
    for a, b in zip(occs, ooccs):
        assert maxdiff(a, b) < PIX_TOL


@slow
def test_kinetics_teacher_forced_clip_vs_oracle(kinetics):
    """BASELINE configs[2] geometry end to end on the decoder side: 5 conditioning frames -> 11 frames decoded through the
    8-slot context ring from given tokens (teacher-forced: the comparison never depends on a sampled token), per frame."""
    qv, qopt, vid, nets = kinetics["qv"], kinetics["qopt"], kinetics["vid"][:1], kinetics["nets"]
    enc = qv({"vid": vid.clone()}, mode="vid_encoder")
    with torch.no_grad():
        want_enc = O.qvid_encode(nets, qopt, vid)
    assert torch.equal(enc["code"].cpu(), want_enc["code"])
    code = torch.randint(0, 16384, (1, 16 * 64), generator=torch.Generator().manual_seed(9))
    code[:, :5 * 64] = want_enc["code"][:, :5 * 64]
    inter = [f[:, :5].contiguous() for f in enc["inter"]]
    fake = qv({"code": code.clone(), "inter": inter}, mode="vid_decoder")["vid"]
    with torch.no_grad():
        want = O.qvid_decode(nets, qopt, code, [f[:, :5].contiguous() for f in want_enc["inter"]])
    assert fake.shape == want.shape == (1, 16, 3, 64, 64)
    per_frame = [maxdiff(fake[:, t], want[:, t]) for t in range(16)]
    print("\nKinetics 5->11 teacher-forced decode, max|pixel diff| per frame:", " ".join(f"{d:.1e}" for d in per_frame))
    assert max(per_frame) < PIX_TOL, per_frame


@slow
def test_bair_teacher_forced_15_frames_vs_oracle():
    """BASELINE configs[1] geometry: ONE 256x256 clip, 1 conditioning frame -> 15 frames decoded recurrently (k grows to 15
    contexts, every synthesized frame is re-encoded into the ring) through the split-bf16 convolutions, against the fp32
    CPU oracle, frame by frame.  This measures the ACCUMULATION of the 2^-16-per-product split error over the recurrence."""
    from ccvs_amd.tools.options import Options, BAIR_ARGV
    from ccvs_amd.models.skip_vid_generator.models.quantized_video_model import QVidModel
    opt = Options().parse(load_qvid_generator=True, load_transformer=True, argv=list(BAIR_ARGV))
    qopt = opt["qvid_generator"]
    torch.manual_seed(0)
    qv = QVidModel(qopt, is_train=False, is_main=True).eval()
    g = torch.Generator().manual_seed(1)
    vid = torch.rand(1, 1, 3, 256, 256, generator=g) * 2 - 1
    _calibrate(qv, vid)
    enc = qv({"vid": vid.clone()}, mode="vid_encoder")
    nets = {"e": cpu_sd(qv.net_e), "q": cpu_sd(qv.net_q), "g": cpu_sd(qv.net_g)}
    code = torch.randint(0, 1024, (1, 16 * 64), generator=torch.Generator().manual_seed(9))
    t0 = time.time()
    with torch.no_grad():
        want_enc = O.qvid_encode(nets, qopt, vid)
        code[:, :64] = want_enc["code"]
        assert torch.equal(enc["code"].cpu(), want_enc["code"])
        fake = qv({"code": code.clone(), "inter": [f.contiguous() for f in enc["inter"]]}, mode="vid_decoder")["vid"]
        want = O.qvid_decode(nets, qopt, code, [f.contiguous() for f in want_enc["inter"]])
    assert fake.shape == want.shape == (1, 16, 3, 256, 256)
    per_frame = [maxdiff(fake[:, t], want[:, t]) for t in range(16)]
    print(f"\nBAIR 1->15 teacher-forced decode ({time.time() - t0:.0f}s of oracle), max|pixel diff| per frame:",
          " ".join(f"{d:.1e}" for d in per_frame))
    assert max(per_frame) < PIX_TOL, per_frame


def test_bair_default_init_codebook_tie_audit():
    """the run with the reference's DEFAULT codebook initialiser U(-1/n_e, 1/n_e) (quantize.py:30), reported
    separately with a tie audit.  One BAIR encode (B = 1, two 256 x 256 frames = 128 latent positions against 1024 codes of
    512 dims): the distances |z|^2 + |e|^2 - 2 z.e are dominated by |z|^2 (|e| <= 0.03), so neighbouring codes sit within a few
    ulp of each other and every argmin is a near-tie.  Indices vs the oracle; every mismatch must lie inside the oracle's own
    rounding of the distance (the rule of test_vq_golden[default]): |d[got] - d[want]| <= 4 eps max|d|.  The count is printed."""
    from ccvs_amd.tools.options import Options, BAIR_ARGV
    from ccvs_amd.models.skip_vid_generator.models.quantized_video_model import QVidModel
    opt = Options().parse(load_qvid_generator=True, load_transformer=True, argv=list(BAIR_ARGV))
    qopt = opt["qvid_generator"]
    torch.manual_seed(0)
    qv = QVidModel(qopt, is_train=False, is_main=True).eval()   # codebook as the reference initialises it: no calibration
    cb = qv.net_q.embedding.weight.detach()
    assert cb.shape == (1024, 512) and float(cb.abs().max()) <= 1.0 / 1024
    vid = torch.rand(1, 2, 3, 256, 256, generator=torch.Generator().manual_seed(1)) * 2 - 1
    enc = qv({"vid": vid.clone()}, mode="vid_encoder")
    nets = {"e": cpu_sd(qv.net_e), "q": cpu_sd(qv.net_q), "g": cpu_sd(qv.net_g)}
    with torch.no_grad():
        want = O.qvid_encode(nets, qopt, vid)
        z_e = O.encoder_forward(nets["e"], qopt, vid)[0].reshape(-1, 512, 8, 8)       # the oracle's pre-quantisation latents
        z_hip = qv.net_e(vid.cuda())[0].reshape(-1, 512, 8, 8).cpu()                    # the HIP encoder's
    got, ref = enc["code"].cpu().flatten(), want["code"].flatten()
    assert got.shape == ref.shape == (128,)
    c = nets["q"]["embedding.weight"]
    zf = z_e.permute(0, 2, 3, 1).reshape(-1, 512)
    dz = (z_hip - z_e).permute(0, 2, 3, 1).reshape(-1, 512).norm(dim=1)               # |z_hip - z_oracle| per position
    d = (zf ** 2).sum(1, keepdim=True) + (c ** 2).sum(1) - 2 * zf @ c.t()
    tol = 4 * torch.finfo(torch.float32).eps * d.abs().max(dim=1).values

    def audit(a, b, what, moved):
        bad = (a != b).nonzero().flatten().tolist()
        for r in bad:
            gap = (d[r, a[r]] - d[r, b[r]]).abs().item()
            # a latent that moved by dz shifts the distance DIFFERENCE of two codes by at most 2 |dz| |e_a - e_b| (Cauchy-Schwarz)
            lim = tol[r].item() + (2 * dz[r] * (c[a[r]] - c[b[r]]).norm()).item() * moved
            assert gap <= lim, f"{what}: position {r}: codes {int(a[r])} vs {int(b[r])}, distance gap {gap:.3e} > {lim:.3e}"
        return len(bad)

    # (a) the argmin kernel alone on the ORACLE's latents (free of convolution round-off): ties inside the rounding of d only
    k_idx = qv.net_q.indices(z_e.cuda()).cpu().flatten()
    n_kernel = audit(k_idx, O.vq_indices(z_e, c).flatten(), "argmin kernel on the oracle's latents", 0)
    # (b) the whole encode (split-bf16 convolutions + argmin) against the oracle's codes
    n_e2e = audit(got, ref, "encode", 1)
    top2 = d.topk(2, dim=1, largest=False).values
    print(f"\nBAIR default-init codebook (U(+-1/1024)): {n_kernel} / 128 positions differ in the argmin kernel alone, "
          f"{n_e2e} / 128 through the whole encode, all inside the audited bound (median top-2 gap "
          f"{float((top2[:, 1] - top2[:, 0]).median()):.2e}, median 4 eps max|d| {float(tol.median()):.2e}, max |dz| {float(dz.max()):.2e})")


@pytest.fixture(scope="module")
def big_gpt():
    from ccvs_amd.models.skip_vid_generator.models.mingpt import GPT
    torch.manual_seed(0)
    net = GPT(vocab_size=1024, block_size=1024, num_blocks=16, n_layer=24, n_head=16, n_embd=1024, emb_mode="temporal", shape=[8, 8]).cuda().eval()
    g = torch.Generator().manual_seed(3)
    with torch.no_grad():
        net.s_emb.copy_(torch.randn(net.s_emb.shape, generator=g) * 0.02)
        net.t_emb.copy_(torch.randn(net.t_emb.shape, generator=g) * 0.02)
        for blk in net.blocks:       # non-trivial LayerNorm parameters and biases: the folded-LN algebra must hold for them
            for ln in (blk.ln1, blk.ln2):
                ln.weight.copy_(1 + 0.1 * torch.randn(1024, generator=g))
                ln.bias.copy_(0.05 * torch.randn(1024, generator=g))
            for lin in (blk.attn.key, blk.attn.query, blk.attn.value, blk.attn.proj, blk.mlp[0], blk.mlp[3]):
                lin.bias.copy_(0.02 * torch.randn(lin.bias.shape, generator=g))
    cfg = O.namespace(z_shape=[8, 8], emb_mode="temporal", n_layer=24, n_head=16, z_len=1024, num_blocks=16, state_size=0)
    return net, cpu_sd(net), cfg


def test_full_size_gpt_logits_vs_oracle(big_gpt):
    """24 layers x 1024 x 16 heads (the BAIR / Kinetics transformer): teacher-forced logits for B = 2, T = 128."""
    net, sd, cfg = big_gpt
    idx = torch.randint(0, 1024, (2, 128), generator=torch.Generator().manual_seed(4))
    got = net(idx.cuda())
    with torch.no_grad():
        want = O.gpt_forward(sd, cfg, idx)
    assert got.shape == want.shape == (2, 128, 1024)
    d = maxdiff(got, want)
    print(f"\nfull-size GPT logits: max|diff| = {d:.2e} (logit std {want.std().item():.3f})")
    assert d < 2e-4, d


def test_full_size_gpt_p2p_prefix_and_decode_steps_vs_oracle(big_gpt):
    """The same network with the point-to-point conditioning prefix (64 tokens at delta_length 15), and the KV-cached
    decode steps against the oracle's full re-forward: logits of 3 successive positions."""
    net, sd, cfg = big_gpt
    g = torch.Generator().manual_seed(6)
    idx = torch.randint(0, 1024, (2, 96), generator=g)
    cond = torch.randint(0, 1024, (2, 64), generator=g)
    dl = torch.tensor([15, 15])
    got = net(idx.cuda(), cond_idx=cond.cuda(), delta_length_cond=dl)
    with torch.no_grad():
        want = O.gpt_forward(sd, cfg, idx, cond, dl)
    assert got.shape == want.shape == (2, 96, 1024)
    assert maxdiff(got, want) < 2e-4, maxdiff(got, want)
    # decode engine: prefill 93 tokens, then 3 single-token steps == rows 92..95 of the teacher-forced logits
    net.begin(2, 64 + 96)
    lg = net.prefill(idx[:, :93].cuda(), cond.cuda(), dl)
    assert maxdiff(lg, want[:, 92]) < 2e-4
    for t in range(93, 96):
        lg = net.step(idx[:, t:t + 1].cuda())
        assert maxdiff(lg, want[:, t]) < 2e-4, t


@slow
def test_bair_batch16_free_running_host_noise_audit_vs_oracle():
    """The BENCHED configuration against the oracle: BAIR geometry, batch 16, sampled (top-k 100), host-drawn noise -- the
    reference's seeded torch.multinomial stream -- on the default schedule (token groups of 4 batches x 2 chains beside two decode
    streams), 8 batches free-running.  For 9 sampled (batch, clip, position) triples at T in {64, 512, 1023} -- the first pick of
    a clip (eager, block 0 of its noise stream), a mid pick and the last (graph replays reading blocks 448 and 959) -- the
    oracle's `gpt_forward` runs on the HIP-generated prefix: its logits agree with the HIP path's teacher-forced logits within
    2e-4, and the token the HIP loop picked at T is the oracle's `get_icode` pick from those logits on the SAME noise row
    (torch.multinomial = argmax(probs / Exp(1) block), checked against torch.multinomial itself below)."""
    from ccvs_amd.tools.options import Options, BAIR_ARGV
    from ccvs_amd.helpers.generator import Generator
    opt = Options().parse(load_qvid_generator=True, load_transformer=True,
                          argv=list(BAIR_ARGV) + ["--batch_size_vid", "16", "--x_sample_noise", "host", "--rec_pass", "false"])
    xopt = opt["transformer"]
    torch.manual_seed(0)
    gen = Generator(opt).build_models()
    net = gen.transformer_model.net_t
    g = torch.Generator().manual_seed(3)
    with torch.no_grad():
        net.s_emb.copy_(torch.randn(net.s_emb.shape, generator=g) * 0.02)
        net.t_emb.copy_(torch.randn(net.t_emb.shape, generator=g) * 0.02)
    _calibrate(gen.vid_model, gen.synthetic_batch(2, seed=1)["vid"][:, :1])
    n_batches, B, V, steps = 8, 16, 1024, 960
    batches = [{"vid": gen.synthetic_batch(B, seed=1 + i)["vid"].cuda()} for i in range(n_batches)]
    torch.manual_seed(77)
    res = gen.run_pipelined(iter(batches))
    torch.cuda.synchronize()
    assert gen.last_lanes == 4 and gen.last_chains == 2 and [n for n, _ in gen.pipeline_token_groups()] == [4, 4]
    codes = [r["fake"]["code"].cpu() for r in res]
    assert all(c.shape == (B, 1024) for c in codes)

    # the generator stream of the run, re-drawn: batch after batch, one [16, 1024] block per new token
    picks = [(0, 0), (3, 7), (6, 15)]                       # (batch, clip): first group / its last member / the other chain
    torch.manual_seed(77)
    rows = {}
    for i in range(n_batches):
        for s in range(steps):
            blk = torch.empty(B, V).exponential_(1)
            for bi, clip in picks:
                if bi == i and 64 + s in (64, 512, 1023):
                    rows[(bi, clip, 64 + s)] = blk[clip].clone()
    # torch.multinomial(p, 1) IS argmax(p / Exp(1) block of the generator): the formula the audit applies per row
    p16 = torch.softmax(torch.randn(16, V, generator=torch.Generator().manual_seed(5)), dim=-1)
    g1, g2 = torch.Generator().manual_seed(6), torch.Generator().manual_seed(6)
    assert torch.equal(torch.multinomial(p16, 1, generator=g1)[:, 0], torch.argmax(p16 / torch.empty_like(p16).exponential_(1, generator=g2), dim=-1))

    sd = cpu_sd(net)
    cfg = O.namespace(z_shape=[8, 8], emb_mode="temporal", n_layer=24, n_head=16, z_len=1024, num_blocks=xopt.num_blocks, state_size=0)
    t0 = time.time()
    worst = 0.0
    for T in (64, 512, 1023):
        prefix = torch.stack([codes[bi][clip, :T] for bi, clip in picks])             # [3, T] tokens the HIP loop generated
        got = net(prefix.cuda())[:, -1].cpu()                                          # HIP teacher-forced logits at T - 1
        with torch.no_grad():
            want = O.gpt_forward(sd, cfg, prefix)[:, -1]
        worst = max(worst, maxdiff(got, want))
        assert maxdiff(got, want) < 2e-4, (T, maxdiff(got, want))
        logits = O.top_k_logits(want / xopt.temperature, xopt.top_k)
        probs = torch.softmax(logits, dim=-1)
        for j, (bi, clip) in enumerate(picks):
            pick = int(torch.argmax(probs[j] / rows[(bi, clip, T)]))
            assert pick == int(codes[bi][clip, T]), f"batch {bi} clip {clip} position {T}: HIP picked {int(codes[bi][clip, T])}, the oracle {pick}"
    print(f"\nBAIR B=16 free-running audit: 9 (batch, clip, T) picks equal the oracle's on the same noise rows; logits max|diff| {worst:.2e} "
          f"({time.time() - t0:.0f}s of oracle)")
