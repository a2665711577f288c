"""-m gpu: the rows of SURVEY 8f around the path -- state stream (StateModel), step-by-step driver, rec pass, output stage,
checkpoint round trip -- and the multi-GPU sampling invariants, against the oracle / the reference's golden vectors."""
import os

import numpy as np
import pytest
import torch

from oracle import ccvs_oracle as O
from tests.test_e2e_gpu import TINY_ARGV, PIX_TOL, _sd, _load, maxdiff, tiny, _audit_tokens  # noqa: F401  (tiny is a fixture)

pytestmark = pytest.mark.gpu

TINY_STATEMODEL_ARGV = TINY_ARGV + [
    "--x_state", "--x_z_len", "264", "--x_z_chunk", "66", "--x_top_k_state", "5",
    "--s_state_size", "2", "--s_state_num", "24", "--s_state_hsize", "8",
]


@pytest.fixture(scope="module")
def state_model(golden_dir):
    from ccvs_amd.tools.options import Options
    from ccvs_amd.models.skip_vid_generator.models.state_model import StateModel
    gold = np.load(os.path.join(golden_dir, "tiny_statemodel.npz"))
    opt = Options().parse(load_qvid_generator=True, load_transformer=True, load_state_estimator=True, argv=TINY_STATEMODEL_ARGV)
    sm = StateModel(opt["state_estimator"], is_train=False, is_main=True).eval()
    _load(sm.net_s, _sd(gold, "s"))
    _load(sm.net_q, _sd(gold, "sq"))
    return dict(gold=gold, sm=sm, opt=opt, nets={"s": _sd(gold, "s"), "sq": _sd(gold, "sq")})


def test_state_model_golden(state_model):
    """StateModel.encode / decode (state_model.py:109-124) against the reference's own outputs: estimated state within
    1e-5, state tokens (scalar VQ, e_dim = 1) bit-exact, decode exact."""
    g, sm = state_model["gold"], state_model["sm"]
    z = torch.from_numpy(g["z"])
    assert maxdiff(sm.net_s(z.cuda()), torch.from_numpy(g["state"])) < 1e-5
    enc = sm({"z": z.clone()}, mode="vid_encoder")["state_code"]
    assert torch.equal(enc.cpu(), torch.from_numpy(g["state_code"]))
    given = sm({"z": z.clone(), "state": torch.from_numpy(g["given"])}, mode="vid_encoder")["state_code"]
    assert torch.equal(given.cpu(), torch.from_numpy(g["given_code"]))
    dec = sm({"state_code": torch.from_numpy(g["state_code"])}, mode="vid_decoder")["state"]
    assert dec.shape == (2, 3, 2) and maxdiff(dec, torch.from_numpy(g["state_dec"])) == 0
    with pytest.raises(ValueError):
        sm({}, mode="nope")


def test_scalar_vq_ties_and_edges():
    """e_dim = 1 quantiser: exact ties go to the lowest index like torch.argmin, values outside the codebook range clamp to
    the end codes, and the association (z^2 + e^2) - 2ze of quantize.py:46-48 is reproduced bit for bit."""
    from ccvs_amd.models.skip_vid_generator.modules.quantize import VectorQuantizer
    q = VectorQuantizer(128, 1, beta=0.25).cuda()
    cb = torch.rand(128, 1, generator=torch.Generator().manual_seed(3))
    cb[5] = cb[90]                                     # duplicate code: index 5 must win
    with torch.no_grad():
        q.embedding.weight.copy_(cb.cuda())
    z = torch.rand(4, 7, 2, generator=torch.Generator().manual_seed(4)) * 1.4 - 0.2
    z[0, 0, 0] = cb[90, 0]
    z[0, 0, 1] = 0.5 * (cb[10, 0] + cb[11, 0])       # midpoint of two codes
    got = q(z.cuda())[2][2].view(-1).cpu()
    assert torch.equal(got, O.vq_indices(z, cb))
    assert got[0] == 5


def test_state_conditioned_generator_vs_oracle(tiny, state_model):
    """scripts/bairhd/save_videos_state_off.sh in miniature: frames encoded, state estimated from the quantised latents and
    tokenised, [state | frame] tokens predicted jointly (greedy), clip decoded, predicted state decoded back to (x, y)."""
    from ccvs_amd.helpers.generator import Generator
    from ccvs_amd.models.skip_vid_generator.models.transformer_model import Transformer
    opt = state_model["opt"]
    xopt = opt["transformer"]
    xopt.sample, xopt.top_k, xopt.sample_state = False, 10, False
    torch.manual_seed(0)
    tr = Transformer(xopt, is_train=False, is_main=True).eval()
    with torch.no_grad():
        for name in ("s_emb", "t_emb", "state_s_emb"):
            getattr(tr.net_t, name).normal_(0, 0.02)
    gen = Generator(opt)
    gen.vid_model, gen.transformer_model, gen.state_model = tiny["qv"], tr, state_model["sm"]
    data = gen.synthetic_batch(2, seed=41)
    out = gen.generate_vid({"vid": data["vid"].clone()})
    nets = dict(tiny["nets"])
    nets.update(state_model["nets"])
    nets["t"] = {k: v.detach().cpu() for k, v in tr.net_t.state_dict().items()}
    with torch.no_grad():
        enc = O.qvid_encode(nets, tiny["qopt"], data["vid"])
        state_all = O.state_encode(nets, opt["state_estimator"], enc["z"])
        assert torch.equal(out["enc_code"].cpu(), enc["code"])
        trace = []
        code, state = O.generate_fake(nets["t"], xopt, enc["code"][:, :64], 4 * 64 + 4 * 2, state_code=state_all[:, :2], trace=trace)
        want_vid = O.qvid_decode(nets, tiny["qopt"], code, [f[:, :1].contiguous() for f in enc["inter"]])
    assert torch.equal(out["fake"]["state_code"].cpu(), state)
    assert torch.equal(out["fake"]["code"].cpu(), code)
    assert maxdiff(out["fake"]["vid"], want_vid) < PIX_TOL
    assert maxdiff(out["fake"]["state"], O.state_decode(nets, opt["state_estimator"], state)) == 0
    assert maxdiff(out["real_state"], O.state_decode(nets, opt["state_estimator"], state_all)) == 0


def test_step_by_step_driver_vs_oracle(tiny):
    """`--step_by_step` (helpers/generator.py:132-159): per frame predict -> decode -> re-encode -> overwrite the tokens."""
    from ccvs_amd.helpers.generator import Generator
    xopt = tiny["xopt"]
    xopt.sample, xopt.top_k, xopt.step_by_step = False, 10, True
    try:
        gen = Generator(tiny["opt"])
        gen.vid_model, gen.transformer_model = tiny["qv"], tiny["tr"]
        data = gen.synthetic_batch(2, seed=51)
        out = gen.generate_vid({"vid": data["vid"].clone()})
        with torch.no_grad():
            want = O.generate_vid_step_by_step(tiny["nets"], tiny["qopt"], xopt, data["vid"])
        assert out["fake"]["vid"].shape == want["vid"].shape == (2, 4, 3, 32, 32)
        assert torch.equal(out["fake"]["code"].cpu(), want["code"])
        assert maxdiff(out["fake"]["vid"], want["vid"]) < PIX_TOL
    finally:
        xopt.step_by_step = False


def test_rec_only_and_rec_pass_flag(tiny):
    """`--rec_only` yields the reconstruction alone; `--rec_pass false` (the bench setting) skips it."""
    from ccvs_amd.helpers.generator import Generator
    xopt = tiny["xopt"]
    gen = Generator(tiny["opt"])
    gen.vid_model, gen.transformer_model = tiny["qv"], tiny["tr"]
    data = gen.synthetic_batch(2, seed=61)
    xopt.sample = False
    try:
        xopt.rec_only = True
        out = gen.generate_vid({"vid": data["vid"].clone()})
        assert out["fake"] is None and out["rec"]["vid"].shape == (2, 4, 3, 32, 32)
        with torch.no_grad():
            enc = O.qvid_encode(tiny["nets"], tiny["qopt"], data["vid"])
            want = O.qvid_decode(tiny["nets"], tiny["qopt"], enc["code"], [f[:, :1].contiguous() for f in enc["inter"]])
        assert maxdiff(out["rec"]["vid"], want) < PIX_TOL
        xopt.rec_only, xopt.rec_pass = False, False
        out = gen.generate_vid({"vid": data["vid"].clone()})
        assert out["rec"] is None and out["fake"] is not None
    finally:
        xopt.rec_only, xopt.rec_pass = False, True


def test_save_video_batch_and_state_marker(tmp_path):
    """Output stage (helpers/generator.py:285-333): clamp / rescale / uint8 / channels-last pack equal to the oracle's, file
    naming `vid_{bs*iter+i:05d}`, and the 3x3 state marker of the state scripts."""
    from ccvs_amd.helpers.generator import save_video_batch, draw_cross
    vid = torch.rand(2, 3, 3, 256, 256, generator=torch.Generator().manual_seed(1)) * 2.4 - 1.2
    u8 = save_video_batch(vid.cuda(), 2, 3, str(tmp_path / "fake"), 4, True, False, [-1, 1], "bairhd")
    assert torch.equal(u8, O.pack_u8(vid))
    names = sorted(os.listdir(tmp_path / "fake"))
    assert [n.split(".")[0] for n in names] == ["vid_00006", "vid_00007"]
    state = torch.tensor([[[0.5, 0.25]] * 3, [[0.999, 0.0]] * 3])
    marked = save_video_batch(vid.cuda(), 2, 0, str(tmp_path / "state"), 4, True, False, [-1, 1], "bairhd", state=state)
    want = O.mark_state(O.pack_u8(vid), state, "bairhd")   # the oracle's own restatement of the marker (generator.py:311-359)
    assert torch.equal(marked, want)
    assert not torch.equal(marked, u8)
    # imagenet_norm branch: de-normalise with the ImageNet statistics, clamp to [0, 1]
    inet = save_video_batch(vid.cuda(), 2, 0, str(tmp_path / "inet"), 4, True, True, [-1, 1], "kinetics600")
    assert torch.equal(inet, O.pack_u8_imagenet(vid))     # byte work: exact (mul, add, clamp, x255, truncate in the reference's order)


def test_checkpoint_round_trip(tiny, tmp_path):
    """save_network / load_network (models/__init__.py:5-132): file naming, `latest` replacement, `block_delta` key shift,
    non-strict loading, `attn.mask` buffers of reference checkpoints dropped."""
    from ccvs_amd.models import save_network, load_network
    from ccvs_amd.models.skip_vid_generator.models.mingpt import GPT
    from types import SimpleNamespace
    opt = SimpleNamespace(checkpoint_path=str(tmp_path), load_path=str(tmp_path), which_iter="latest", not_strict=False)
    net = tiny["tr"].net_t
    save_network(net, "transformer_t", 100, opt, latest=True)
    save_network(net, "transformer_t", 200, opt, latest=True)
    assert os.listdir(tmp_path) == ["transformer_t_latest_net_200.pth"]
    # a reference checkpoint also carries the 1024^2 causal-mask buffers: add one, it must be ignored on load
    path = os.path.join(tmp_path, "transformer_t_latest_net_200.pth")
    sd = torch.load(path)
    sd["blocks.0.attn.mask"] = torch.ones(1, 1, 8, 8)
    torch.save(sd, path)
    torch.manual_seed(123)
    fresh = GPT(vocab_size=32, block_size=256, num_blocks=4, n_layer=2, n_head=2, n_embd=32, emb_mode="temporal", shape=[8, 8]).cuda()
    load_network(fresh, "transformer_t", opt)
    for (ka, a), (kb, b) in zip(sorted(net.state_dict().items()), sorted(fresh.state_dict().items())):
        assert ka == kb and torch.equal(a.cpu(), b.cpu()), ka
    # and the loaded weights are what the kernels then use (packed-weight caches keyed by parameter version)
    code = torch.randint(0, 32, (2, 70), generator=torch.Generator().manual_seed(2)).cuda()
    assert torch.equal(fresh(code), net(code))
    # block_delta: decoder checkpoints saved with fewer levels load shifted (quantized_video_model.py:198)
    g = tiny["qv"].net_g
    opt2 = SimpleNamespace(checkpoint_path=str(tmp_path), load_path=str(tmp_path), which_iter=7, not_strict=True)
    shifted = {}
    for k, v in g.state_dict().items():
        if k.startswith("blocks.") and int(k.split(".")[1]) >= 1:
            parts = k.split(".")
            parts[1] = str(int(parts[1]) - 1)
            shifted[".".join(parts)] = v
    torch.save(shifted, os.path.join(tmp_path, "qvid_g_net_7.pth"))
    import copy
    g2 = copy.deepcopy(g)
    with torch.no_grad():
        for p in g2.parameters():
            p.zero_()
    load_network(g2, "qvid_g", opt2, block_delta=1)
    for k, v in g.state_dict().items():
        if k.startswith("blocks.") and int(k.split(".")[1]) >= 1:
            assert torch.equal(g2.state_dict()[k], v), k


def test_decode_graph_follows_weight_updates(tiny):
    """A captured decode hipGraph must not outlive the packed weights it points at: after a parameter update the next
    generate() re-captures and the tokens equal an eager (graph-free) run on the new weights."""
    tr, xopt = tiny["tr"], tiny["xopt"]
    xopt.sample, xopt.top_k = False, 10
    code = torch.from_numpy(tiny["gold"]["gen_code_greedy"])[:, :64].clone()
    first = tr({"code": code.clone()}, mode="inference", total_len=128)["code"].cpu()
    saved = tr.net_t.head.weight.detach().clone()
    try:
        with torch.no_grad():
            tr.net_t.head.weight.copy_(saved.flip(0))           # permutes the vocabulary
            tr.net_t.blocks[0].mlp[3].weight.mul_(1.5)
        graph_run = tr({"code": code.clone()}, mode="inference", total_len=128)["code"].cpu()
        xopt.use_graph = False
        eager_run = tr({"code": code.clone()}, mode="inference", total_len=128)["code"].cpu()
        assert torch.equal(graph_run, eager_run)
        assert not torch.equal(graph_run, first)
    finally:
        xopt.use_graph = True
        with torch.no_grad():
            tr.net_t.head.weight.copy_(saved)
            tr.net_t.blocks[0].mlp[3].weight.div_(1.5)
    again = tr({"code": code.clone()}, mode="inference", total_len=128)["code"].cpu()
    assert torch.equal(again, first)


def test_sampled_tokens_do_not_depend_on_sharding(tiny):
    """in-kernel Philox noise keyed by (seed, iteration) with the GLOBAL clip index as counter row: a batch of 4
    generated at once == the same 4 clips generated as two 'ranks' of 2 (row offsets 0 and 2), token for token."""
    from ccvs_amd.helpers.generator import Generator
    from types import SimpleNamespace
    xopt = tiny["xopt"]
    xopt.sample, xopt.top_k = True, 10
    old_noise = tiny["tr"].sample_noise
    tiny["tr"].sample_noise = "device"
    try:
        gen = Generator(tiny["opt"])
        gen.vid_model, gen.transformer_model = tiny["qv"], tiny["tr"]
        xopt.rec_pass = False
        data = gen.synthetic_batch(4, seed=71)["vid"]
        full = gen.generate_vid({"vid": data.clone()}, global_iter=3)["fake"]["code"].cpu()
        parts = []
        for rank in range(2):
            gen.engine = SimpleNamespace(rank=rank)
            parts.append(gen.generate_vid({"vid": data[2 * rank: 2 * rank + 2].clone()}, global_iter=3)["fake"]["code"].cpu())
        gen.engine = None
        assert torch.equal(torch.cat(parts), full)
        other = gen.generate_vid({"vid": data.clone()}, global_iter=4)["fake"]["code"].cpu()
        assert not torch.equal(other[:, 64:], full[:, 64:])       # another iteration draws other noise
        assert not torch.equal(full[0, 64:], full[1, 64:])
    finally:
        xopt.sample, xopt.rec_pass = False, True
        tiny["tr"].sample_noise = old_noise


def test_beam_search_golden(tiny, golden_dir):
    """Beam search on the KV cache (hypotheses = cache rows, pruning = row gather) against the reference's own token streams:
    one continuation per hypothesis, expand-and-prune (`--x_no_sample`), and multinomial proposals from the seeded CPU stream."""
    g = np.load(os.path.join(golden_dir, "tiny_beam.npz"))
    tr, xopt = tiny["tr"], tiny["xopt"]
    code = torch.from_numpy(g["code"])
    old = (xopt.sample, xopt.top_k, getattr(xopt, "beam_size", None), getattr(xopt, "no_sample", False), tr.sample_noise)
    empty = torch.tensor([])
    try:
        xopt.beam_size, xopt.top_k = 3, 10
        tr.sample_noise, tr.generator = "host", None
        for name, sample, no_sample, add_len in (("greedy", False, False, 12), ("prune", False, True, 12), ("sampled", True, False, 8),
                                                 ("sampled_prune", True, True, 8)):
            xopt.sample, xopt.no_sample = sample, no_sample
            torch.manual_seed(5)
            got = tr.fill_code(code.clone().cuda(), empty.cuda(), empty.cuda(), None, empty, add_len=add_len)[0]
            assert torch.equal(got.cpu(), torch.from_numpy(g["beam_" + name])), name
        with pytest.raises(NotImplementedError):
            tr.fill_code(code.clone().cuda(), empty.cuda(), code[:, :64].cuda(), torch.tensor([3, 3]), empty, add_len=4)
    finally:
        xopt.sample, xopt.top_k, xopt.beam_size, xopt.no_sample, tr.sample_noise = old


def test_start_and_label_tokens_golden(golden_dir):
    """`--x_use_start_token --x_cat`: label and start embeddings in front of the sequence (mingpt.py:136-141,289-297): same
    parameter tree / initialiser stream as the reference, teacher-forced logits (prefix positions included) and a greedy
    KV-cached continuation against the reference's outputs."""
    from ccvs_amd.tools.options import Options
    from ccvs_amd.models.skip_vid_generator.models.transformer_model import Transformer
    g = np.load(os.path.join(golden_dir, "tiny_prefix.npz"))
    opt = Options().parse(load_qvid_generator=True, load_transformer=True,
                          argv=TINY_ARGV + ["--x_use_start_token", "--x_cat", "--categories", "a", "b", "c", "--x_top_k", "10"])
    xopt = opt["transformer"]
    torch.manual_seed(0)
    tr = Transformer(xopt, is_train=False, is_main=True).eval()
    ref = _sd(g, "t")
    own = tr.net_t.state_dict()
    assert set(own) == set(ref)
    for k in ("start_tok_emb", "lbl_emb.weight", "tok_emb.weight", "head.weight"):      # same initialiser stream as the reference
        assert torch.equal(own[k].cpu(), ref[k]), k
    _load(tr.net_t, ref)
    assert tr.net_t.get_block_size() == 256 + 2
    code, lbl = torch.from_numpy(g["code"]), torch.from_numpy(g["lbl"])
    logits = tr.net_t(code.cuda(), lbl_idx=lbl.cuda())
    assert logits.shape == (3, 72, 32) and maxdiff(logits, torch.from_numpy(g["logits"])) < 1e-4
    xopt.sample = False
    empty = torch.tensor([])
    out = tr.fill_code(code[:, :64].clone().cuda(), empty.cuda(), empty.cuda(), None, lbl.cuda(), add_len=10)[0]
    assert torch.equal(out.cpu(), torch.from_numpy(g["greedy"]))
    out = tr({"code": code[:, :64].clone(), "vid_lbl": lbl.clone()}, mode="inference", total_len=74)["code"]
    assert torch.equal(out.cpu(), torch.from_numpy(g["greedy"]))


def test_unconditional_generation_vs_oracle():
    """scripts/bairhd/save_videos_unc.sh: `--x_cond_len 0 --x_use_start_token` -- NO conditioning frame: the token loop starts
    from the start token alone (an empty code tensor), the decoder's first frame has no context (`has_ctx=False`) and the ring
    fills from the synthesized frames.  Greedy tokens and pixels against the oracle, fresh weights."""
    from ccvs_amd.tools.options import Options
    from ccvs_amd.helpers.generator import Generator
    opt = Options().parse(load_qvid_generator=True, load_transformer=True,
                          argv=TINY_ARGV + ["--x_use_start_token", "--x_cond_len", "0", "--x_top_k", "10"])
    xopt, qopt = opt["transformer"], opt["qvid_generator"]
    xopt.sample, xopt.rec_pass = False, False
    torch.manual_seed(5)
    gen = Generator(opt).build_models()
    with torch.no_grad():
        net = gen.transformer_model.net_t
        for p_ in (net.s_emb, net.t_emb):
            p_.normal_(0, 0.02)
        vid = gen.synthetic_batch(2, seed=71)["vid"]
        z_e, _ = gen.vid_model.net_e(vid[:, :1].cuda())
        cb = gen.vid_model.net_q.embedding.weight
        cb.copy_(torch.randn(cb.shape, generator=torch.Generator().manual_seed(4)).cuda() * z_e.std())
    out = gen.generate_vid({"vid": vid.clone()})
    cpu = lambda m: {k: v.detach().cpu() for k, v in m.state_dict().items()}
    nets = {"e": cpu(gen.vid_model.net_e), "q": cpu(gen.vid_model.net_q), "g": cpu(gen.vid_model.net_g), "t": cpu(gen.transformer_model.net_t)}
    with torch.no_grad():
        want = O.generate_vid(nets, qopt, xopt, vid)
    assert out["fake"]["code"].shape == want["code"].shape == (2, xopt.vid_len * 64)
    assert torch.equal(out["fake"]["code"].cpu(), want["code"]), "greedy tokens from the start token alone differ from the oracle"
    assert out["fake"]["vid"].shape == want["vid"].shape
    assert maxdiff(out["fake"]["vid"], want["vid"]) < PIX_TOL



def test_encode_conditioning_frames_only(tiny):
    """`--encode_all false`: with the rec pass off the encoder sees only the frames the conditioning crop keeps
    (helpers/generator.py:93-99) -- the synthesized clip and its tokens are the same bits as with the reference's encode of the
    whole clip (frames are encoded independently), `enc_code` holds the kept frames' codes; serial and pipelined schedule; with
    the rec pass on, or an end frame to read (point-to-point), every frame is still encoded."""
    from ccvs_amd.helpers.generator import Generator
    xopt = tiny["xopt"]
    size = int(np.prod(tiny["qopt"].z_shape))                     # tokens per frame
    xopt.sample, xopt.rec_pass = False, False
    try:
        gen = Generator(tiny["opt"])
        gen.vid_model, gen.transformer_model = tiny["qv"], tiny["tr"]
        data = gen.synthetic_batch(2, seed=91)
        full = gen.generate_vid({"vid": data["vid"].clone()}, 7)
        xopt.encode_all = False
        n_keep = gen._frames_to_encode(xopt.vid_len, size)
        assert 1 <= n_keep < xopt.vid_len and n_keep == -(-xopt.cond_len // size)
        cond = gen.generate_vid({"vid": data["vid"].clone()}, 7)
        assert cond["enc_code"].shape == (2, n_keep * size) and torch.equal(cond["enc_code"], full["enc_code"][:, :n_keep * size])
        assert torch.equal(cond["fake"]["code"], full["fake"]["code"]) and torch.equal(cond["fake"]["vid"], full["fake"]["vid"])
        assert cond["real"].is_cuda and cond["real"].shape == full["real"].shape
        res = gen.run_pipelined(({"vid": data["vid"].clone()} for _ in range(2)), first_iter=7, lanes=2, chains=1)
        assert torch.equal(res[0]["fake"]["vid"], full["fake"]["vid"]) and res[0]["enc_code"].shape == (2, n_keep * size)
        xopt.rec_pass = True
        assert gen._frames_to_encode(xopt.vid_len, size) == xopt.vid_len      # the rec pass reads every frame's codes
        xopt.rec_pass, xopt.p2p = False, True
        assert gen._frames_to_encode(xopt.vid_len, size) == xopt.vid_len      # the end frame is read
    finally:
        xopt.sample, xopt.rec_pass, xopt.encode_all, xopt.p2p = False, True, True, False
