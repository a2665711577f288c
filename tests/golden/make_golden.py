"""Generate the golden fixtures under tests/golden/ from the imported reference.

Run in the build container only (needs /root/reference):
    python tests/golden/make_golden.py
It imports the upstream reference through `ref_harness` (stubs, CPU), runs the
reference's own modules on seeded inputs and stores inputs, weights (reference
`state_dict()` layouts) and outputs as `.npz`.  The fixtures are DATA; no reference
source travels.  While generating it also cross-checks the oracle restatement and
prints the max abs difference per item.
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, HERE)
sys.path.insert(0, ROOT)

import ref_harness as rh  # noqa: E402
from oracle import ccvs_oracle as O  # noqa: E402


def sd_np(prefix, module):
    return {f"{prefix}/{k}": v.detach().cpu().numpy() for k, v in module.state_dict().items()
            if not k.endswith(".mask") and not k.endswith(".kernel")}


def report(name, a, b):
    d = (a.float() - b.float()).abs().max().item() if a.numel() else 0.0
    print(f"  oracle vs reference  {name:40s} max|diff| = {d:.3e}")
    return d


def tiny_end_to_end(ns):
    opt = rh.parse_reference_options(rh.TINY_ARGV)
    qopt, xopt = opt["qvid_generator"], opt["transformer"]
    torch.manual_seed(0)
    qv = ns.qvm.QVidModel(qopt, is_train=False, is_main=True).eval()
    tr = ns.tm.Transformer(xopt, is_train=False, is_main=True).eval()
    # give the positional tables non-zero values (they are zero-initialised, mingpt.py:152-153)
    g = torch.Generator().manual_seed(3)
    with torch.no_grad():
        tr.net_t.s_emb.normal_(0, 0.02, generator=g)
        tr.net_t.t_emb.normal_(0, 0.02, generator=g)
    torch.manual_seed(1)
    vid = torch.rand(2, 4, 3, 32, 32) * 2 - 1
    out = {}
    with torch.no_grad():
        enc0 = qv({"vid": vid.clone()}, mode="vid_encoder")
        # documented synthetic codebook: randn * std(z_e) (SURVEY section 7 hard part 2)
        z_e, _ = qv.net_e(vid)
        torch.manual_seed(4)
        qv.net_q.embedding.weight.copy_(torch.randn_like(qv.net_q.embedding.weight) * z_e.std())
        enc = qv({"vid": vid.clone()}, mode="vid_encoder")
        out["z_e"] = z_e
        out["enc_code"] = enc["code"]
        for i, f in enumerate(enc["inter"]):
            out[f"enc_inter{i}"] = f
        # top-2 distance gap audit for the argmin
        zf = z_e.transpose(-3, -1).transpose(-3, -2).reshape(-1, z_e.shape[2])
        cb = qv.net_q.embedding.weight
        d = (zf ** 2).sum(1, keepdim=True) + (cb ** 2).sum(1) - 2 * zf @ cb.t()
        top2 = torch.topk(d, 2, dim=1, largest=False)[0]
        out["vq_min_gap"] = (top2[:, 1] - top2[:, 0]).min()

        # generate_vid sequence (helpers/generator.py:83-164), greedy sampling for determinism
        size = 64
        crop_prop = xopt.cond_len / (size * xopt.vid_len)
        code_c = enc["code"][:, :int(crop_prop * enc["code"].size(1))]
        inter_c = [f[:, :int(crop_prop * f.size(1))].contiguous() for f in enc["inter"]]
        total_len = xopt.vid_len * size
        xopt.sample = False
        xopt.top_k = 10
        fake_enc = tr({"code": code_c.clone()}, mode="inference", total_len=total_len)
        out["gen_code_greedy"] = fake_enc["code"]
        # teacher-forced logits on the generated stream
        out["gpt_logits"] = tr.net_t(fake_enc["code"][:, :-1])
        with rh.patched_overlapping_shift():
            fake = qv({"code": fake_enc["code"].clone(), "inter": [f.clone() for f in inter_c]}, mode="vid_decoder")
        out["fake_vid"] = fake["vid"]
        # reconstruction of the conditioning frame alone through the decoder graph
        rec0, _, fl, oc, _ = qv.net_g(enc["z"][:, :1].contiguous(), [[f[:, :1] for f in enc["inter"]]], return_all=True)
        out["dec_cond_vid"] = rec0
        for i, (a, b) in enumerate(zip(fl, oc)):
            out[f"dec_cond_flow{i}"] = a
            out[f"dec_cond_occ{i}"] = b
        # sampled path: logits -> multinomial with a seeded generator
        xopt.sample = True
        torch.manual_seed(7)
        samp = tr({"code": code_c.clone()}, mode="inference", total_len=64 + 8)
        out["gen_code_sampled_seed7"] = samp["code"]
        # step decoder: one frame
        with rh.patched_overlapping_shift():
            step = qv({"code": fake_enc["code"][:, 64:128].clone(), "inter": [f.clone() for f in inter_c]},
                      mode="vid_step_decoder")
        out["step_vid"] = step["vid"]
        out["step_code"] = step["code"]

    # oracle cross-check
    nets = {"e": qv.net_e.state_dict(), "q": qv.net_q.state_dict(), "g": qv.net_g.state_dict(), "t": tr.net_t.state_dict()}
    xo = O.namespace(**vars(xopt))
    xo.sample, xo.top_k = False, 10
    with torch.no_grad():
        oenc = O.qvid_encode(nets, qopt, vid)
        report("tiny/enc_code (mismatches)", (oenc["code"] != enc["code"]).float(), torch.zeros(1))
        for i, f in enumerate(oenc["inter"]):
            report(f"tiny/enc_inter{i}", f, enc["inter"][i])
        ocode = O.generate_fake(nets["t"], xo, code_c, total_len)
        report("tiny/gen_code_greedy (mismatches)", (ocode != fake_enc["code"]).float(), torch.zeros(1))
        report("tiny/gpt_logits", O.gpt_forward(nets["t"], xo, fake_enc["code"][:, :-1]), out["gpt_logits"])
        ofake = O.qvid_decode(nets, qopt, fake_enc["code"], [f.clone() for f in inter_c])
        report("tiny/fake_vid", ofake, fake["vid"])
        xo.sample = True
        torch.manual_seed(7)
        osamp = O.generate_fake(nets["t"], xo, code_c, 64 + 8)
        report("tiny/gen_code_sampled (mismatches)", (osamp != samp["code"]).float(), torch.zeros(1))
        ostep = O.qvid_step_decode(nets, qopt, fake_enc["code"][:, 64:128], [f.clone() for f in inter_c])
        report("tiny/step_vid", ostep["vid"], step["vid"])

    arrays = {"vid": vid.numpy()}
    arrays.update({k: v.detach().cpu().numpy() for k, v in out.items()})
    arrays.update(sd_np("e", qv.net_e))
    arrays.update(sd_np("q", qv.net_q))
    arrays.update(sd_np("g", qv.net_g))
    arrays.update(sd_np("t", tr.net_t))
    np.savez_compressed(os.path.join(HERE, "tiny_e2e.npz"), **arrays)
    print("  wrote tiny_e2e.npz", sum(a.nbytes for a in arrays.values()) / 1e6, "MB raw")
    return qopt, xopt


def tiny_state_stream(ns):
    """Ancillary-token stream (BASELINE config 5 in miniature): StftModel.encode, GPT.forward with the
    state / frame interleave, and Transformer.generate_fake with given and with predicted STFT tokens."""
    opt = rh.parse_reference_options(rh.TINY_STATE_ARGV)
    xopt, aopt = opt["transformer"], opt["stft_ae"]
    torch.manual_seed(0)
    tr = ns.tm.Transformer(xopt, is_train=False, is_main=True).eval()
    sm = ns.stft_model.StftModel(aopt, is_train=False, is_main=True).eval()
    g = torch.Generator().manual_seed(3)
    with torch.no_grad():
        tr.net_t.s_emb.normal_(0, 0.02, generator=g)
        tr.net_t.t_emb.normal_(0, 0.02, generator=g)
        tr.net_t.state_s_emb.normal_(0, 0.02, generator=g)
    out = {}
    with torch.no_grad():
        torch.manual_seed(11)
        stft = torch.rand(2, 5, 1, 16, 8) * 2 - 1
        z = sm.net_e(stft)
        torch.manual_seed(12)
        sm.net_q.embedding.weight.copy_(torch.randn_like(sm.net_q.embedding.weight) * z.std())
        state_code = sm({"stft": stft.clone()}, mode="vid_encoder")["state_code"]          # [2, 5 * 2]
        out["stft"], out["stft_z"], out["state_code"] = stft, z, state_code
        torch.manual_seed(13)
        code = torch.randint(0, 32, (2, 64 * 3 + 10))
        # teacher-forced logits, interleaved stream (3 full frames + a partial one, states of 4 frames)
        out["tf_code"] = code
        out["tf_logits"] = tr.net_t(code, state_idx=state_code[:, :8])
        xopt.sample, xopt.top_k, xopt.sample_state = False, 10, False
        # (a) STFT tokens of every frame given (keep_state): only frame tokens are predicted
        total = 4 * 64 + 4 * 2
        fa = tr({"code": code[:, :64].clone(), "state_code": state_code[:, :8].clone()}, mode="inference", total_len=total)
        out["given_code"], out["given_state"] = fa["code"], fa["state_code"]
        # (b) STFT tokens of the first frame only: the others are predicted too
        fb = tr({"code": code[:, :64].clone(), "state_code": state_code[:, :2].clone()}, mode="inference", total_len=total)
        out["pred_code"], out["pred_state"] = fb["code"], fb["state_code"]
        # (c) 5 frames through the 4-frame window (one slide), STFT tokens given
        total5 = 5 * 64 + 5 * 2
        fc = tr({"code": code[:, :64].clone(), "state_code": state_code.clone()}, mode="inference", total_len=total5)
        out["slide_code"], out["slide_state"] = fc["code"], fc["state_code"]
        # (d) sampled, both streams (seeded multinomial)
        xopt.sample, xopt.sample_state = True, True
        torch.manual_seed(7)
        fd = tr({"code": code[:, :64].clone(), "state_code": state_code[:, :2].clone()}, mode="inference", total_len=64 + 2 + 2 + 6)
        out["samp_code"], out["samp_state"] = fd["code"], fd["state_code"]
        xopt.sample, xopt.sample_state = False, False

    nets = {"t": tr.net_t.state_dict(), "ae": sm.net_e.state_dict(), "aq": sm.net_q.state_dict()}
    xo = O.namespace(**vars(xopt))
    with torch.no_grad():
        report("state/stft_code (mismatches)", (O.stft_encode(nets, aopt, stft) != state_code).float(), torch.zeros(1))
        report("state/tf_logits", O.gpt_forward(nets["t"], xo, code, state_idx=state_code[:, :8]), out["tf_logits"])
        oc, os_ = O.generate_fake(nets["t"], xo, code[:, :64], total, state_code=state_code[:, :8])
        report("state/given (mismatches)", (oc != fa["code"]).float(), torch.zeros(1))
        oc, os_ = O.generate_fake(nets["t"], xo, code[:, :64], total, state_code=state_code[:, :2])
        report("state/pred code (mismatches)", (oc != fb["code"]).float(), torch.zeros(1))
        report("state/pred state (mismatches)", (os_ != fb["state_code"]).float(), torch.zeros(1))
        oc, os_ = O.generate_fake(nets["t"], xo, code[:, :64], total5, state_code=state_code)
        report("state/slide (mismatches)", (oc != fc["code"]).float(), torch.zeros(1))
        xo.sample, xo.sample_state = True, True
        torch.manual_seed(7)
        oc, os_ = O.generate_fake(nets["t"], xo, code[:, :64], 64 + 2 + 2 + 6, state_code=state_code[:, :2])
        report("state/sampled code (mismatches)", (oc != fd["code"]).float(), torch.zeros(1))
        report("state/sampled state (mismatches)", (os_ != fd["state_code"]).float(), torch.zeros(1))

    arrays = {k: v.detach().cpu().numpy() for k, v in out.items()}
    arrays.update(sd_np("t", tr.net_t))
    arrays.update(sd_np("ae", sm.net_e))
    arrays.update(sd_np("aq", sm.net_q))
    np.savez_compressed(os.path.join(HERE, "tiny_state.npz"), **arrays)
    print("  wrote tiny_state.npz", sum(a.nbytes for a in arrays.values()) / 1e6, "MB raw")


def tiny_state_front(ns):
    """`--x_state_front` (mingpt.py:261-263): every ancillary token in front of the frame tokens.  Same networks as
    tiny_state.npz (their weights are read from it, not stored again): teacher-forced logits and Transformer.generate_fake
    with given, with predicted and with sampled ancillary tokens."""
    opt = rh.parse_reference_options(rh.TINY_STATE_ARGV + ["--x_state_front"])
    xopt = opt["transformer"]
    assert xopt.state_front
    base = np.load(os.path.join(HERE, "tiny_state.npz"))
    torch.manual_seed(0)
    tr = ns.tm.Transformer(xopt, is_train=False, is_main=True).eval()
    sd_t = {k[2:]: torch.from_numpy(base[k]) for k in base.files if k.startswith("t/")}
    missing = tr.net_t.load_state_dict(sd_t, strict=False)
    assert not missing.unexpected_keys, missing
    state_code = torch.from_numpy(base["state_code"])
    code = torch.from_numpy(base["tf_code"])
    out = {}
    with torch.no_grad():
        out["tf_logits"] = tr.net_t(code, state_idx=state_code[:, :8])
        xopt.sample, xopt.top_k, xopt.sample_state = False, 10, False
        total = 2 * 64 + 4 * 2 + 20          # the state slots of the interleaved count decide which stream a pick goes to
        fa = tr({"code": code[:, :64].clone(), "state_code": state_code[:, :8].clone()}, mode="inference", total_len=total)
        out["given_code"], out["given_state"] = fa["code"], fa["state_code"]
        total_b = 64 + 2 + 2 + 40
        fb = tr({"code": code[:, :64].clone(), "state_code": state_code[:, :2].clone()}, mode="inference", total_len=total_b)
        out["pred_code"], out["pred_state"] = fb["code"], fb["state_code"]
        xopt.sample, xopt.sample_state = True, True
        torch.manual_seed(7)
        fd = tr({"code": code[:, :64].clone(), "state_code": state_code[:, :2].clone()}, mode="inference", total_len=64 + 2 + 2 + 6)
        out["samp_code"], out["samp_state"] = fd["code"], fd["state_code"]
        xopt.sample, xopt.sample_state = False, False
    xo = O.namespace(**vars(xopt))
    nets_t = tr.net_t.state_dict()
    with torch.no_grad():
        report("state_front/tf_logits", O.gpt_forward(nets_t, xo, code, state_idx=state_code[:, :8]), out["tf_logits"])
        oc, os_ = O.generate_fake(nets_t, xo, code[:, :64], total, state_code=state_code[:, :8])
        report("state_front/given (mismatches)", (oc != fa["code"]).float(), torch.zeros(1))
        oc, os_ = O.generate_fake(nets_t, xo, code[:, :64], total_b, state_code=state_code[:, :2])
        report("state_front/pred code (mismatches)", (oc != fb["code"]).float(), torch.zeros(1))
        report("state_front/pred state (mismatches)", (os_ != fb["state_code"]).float(), torch.zeros(1))
        xo.sample, xo.sample_state = True, True
        torch.manual_seed(7)
        oc, os_ = O.generate_fake(nets_t, xo, code[:, :64], 64 + 2 + 2 + 6, state_code=state_code[:, :2])
        report("state_front/sampled code (mismatches)", (oc != fd["code"]).float(), torch.zeros(1))
        report("state_front/sampled state (mismatches)", (os_ != fd["state_code"]).float(), torch.zeros(1))
    print("  shapes:", {k: tuple(v.shape) for k, v in out.items()})
    arrays = {k: v.detach().cpu().numpy() for k, v in out.items()}
    np.savez_compressed(os.path.join(HERE, "tiny_state_front.npz"), **arrays)
    print("  wrote tiny_state_front.npz", sum(a.nbytes for a in arrays.values()) / 1e6, "MB raw")


def tiny_state_model(ns):
    """StateModel (state_model.py:109-124) in miniature: StateEstimator on a quantised latent map, the scalar (e_dim = 1)
    quantiser, and the code -> state decode."""
    opt = rh.parse_reference_options(rh.TINY_STATEMODEL_ARGV)
    sopt = opt["state_estimator"]
    torch.manual_seed(0)
    sm = ns.state_model.StateModel(sopt, is_train=False, is_main=True).eval()
    out = {}
    with torch.no_grad():
        torch.manual_seed(31)
        z = torch.randn(2, 3, sopt.z_size, *sopt.z_shape)
        sm.net_s.fc.bias.normal_(0, 0.3)
        for conv in sm.net_s.convs:
            conv[1].bias.normal_(0, 0.1)
        state = sm.net_s(z)
        enc = sm({"z": z.clone()}, mode="vid_encoder")["state_code"]
        dec = sm({"state_code": enc.clone()}, mode="vid_decoder")["state"]
        given = torch.rand(2, 3, 2)
        enc_given = sm({"z": z.clone(), "state": given.clone()}, mode="vid_encoder")["state_code"]
        out.update(z=z, state=state, state_code=enc, state_dec=dec, given=given, given_code=enc_given)
    nets = {"s": sm.net_s.state_dict(), "sq": sm.net_q.state_dict()}
    with torch.no_grad():
        report("statemodel/state", O.state_estimator_forward(nets["s"], sopt, z), state)
        report("statemodel/code (mismatches)", (O.state_encode(nets, sopt, z) != enc).float(), torch.zeros(1))
        report("statemodel/given code (mismatches)", (O.state_encode(nets, sopt, z, given) != enc_given).float(), torch.zeros(1))
        report("statemodel/decode", O.state_decode(nets, sopt, enc), dec)
    arrays = {k: v.detach().cpu().numpy() for k, v in out.items()}
    arrays.update(sd_np("s", sm.net_s))
    arrays.update(sd_np("sq", sm.net_q))
    np.savez_compressed(os.path.join(HERE, "tiny_statemodel.npz"), **arrays)
    print("  wrote tiny_statemodel.npz", sum(a.nbytes for a in arrays.values()) / 1e6, "MB raw")


def tiny_beam(ns):
    """Beam search of Transformer.fill_code (transformer_model.py:358-391) on the tiny transformer of tiny_e2e.npz: greedy
    continuation per hypothesis, expand-and-prune (`--x_no_sample`), and seeded multinomial proposals."""
    opt = rh.parse_reference_options(rh.TINY_ARGV + ["--x_beam_size", "3", "--x_top_k", "10"])
    xopt = opt["transformer"]
    torch.manual_seed(0)
    ns.qvm.QVidModel(opt["qvid_generator"], is_train=False, is_main=True)   # same construction order as tiny_end_to_end
    tr = ns.tm.Transformer(xopt, is_train=False, is_main=True).eval()
    ref = np.load(os.path.join(HERE, "tiny_e2e.npz"))
    with torch.no_grad():
        for k, v in tr.net_t.state_dict().items():
            if k in ("s_emb", "t_emb"):
                v.copy_(torch.from_numpy(ref["t/" + k]))
            elif not k.endswith(".mask"):
                assert np.array_equal(v.numpy(), ref["t/" + k]), k
    code = torch.from_numpy(ref["gen_code_greedy"])[:, :64].clone()
    empty = torch.tensor([])
    out = {"code": code}
    xo = O.namespace(**vars(xopt))
    sd = tr.net_t.state_dict()
    with torch.no_grad():
        for name, sample, no_sample, add_len in (("greedy", False, False, 12), ("prune", False, True, 12), ("sampled", True, False, 8),
                                                 ("sampled_prune", True, True, 8)):
            xopt.sample, xopt.no_sample = sample, no_sample
            xo.sample, xo.no_sample = sample, no_sample
            torch.manual_seed(5)
            got = tr.fill_code(code.clone(), empty, empty, None, empty, add_len=add_len)[0]
            out["beam_" + name] = got
            torch.manual_seed(5)
            report(f"beam/{name} (mismatches)", (O.fill_code_beam(sd, xo, code.clone(), add_len) != got).float(), torch.zeros(1))
    np.savez_compressed(os.path.join(HERE, "tiny_beam.npz"), **{k: v.numpy() for k, v in out.items()})
    print("  wrote tiny_beam.npz")


def tiny_prefix_tokens(ns):
    """Start token + class-label token in front of the sequence (mingpt.py:136-141,289-297; `--x_use_start_token --x_cat`):
    teacher-forced logits and a greedy continuation from the reference."""
    opt = rh.parse_reference_options(rh.TINY_ARGV + ["--x_use_start_token", "--x_cat", "--categories", "a", "b", "c", "--x_top_k", "10"])
    xopt = opt["transformer"]
    torch.manual_seed(0)
    tr = ns.tm.Transformer(xopt, is_train=False, is_main=True).eval()
    g = torch.Generator().manual_seed(3)
    with torch.no_grad():
        tr.net_t.s_emb.normal_(0, 0.02, generator=g)
        tr.net_t.t_emb.normal_(0, 0.02, generator=g)
    torch.manual_seed(17)
    code = torch.randint(0, 32, (3, 70))
    lbl = torch.tensor([2, 0, 1])
    empty = torch.tensor([])
    out = {"code": code, "lbl": lbl}
    with torch.no_grad():
        out["logits"] = tr.net_t(code, lbl_idx=lbl)
        xopt.sample = False
        out["greedy"] = tr.fill_code(code[:, :64].clone(), empty, empty, None, lbl, add_len=10)[0]
    sd = tr.net_t.state_dict()
    xo = O.namespace(**vars(xopt))
    with torch.no_grad():
        report("prefix/logits", O.gpt_forward(sd, xo, code, lbl_idx=lbl), out["logits"])
        report("prefix/greedy (mismatches)", (O.fill_code(sd, xo, code[:, :64].clone(), 10, lbl=lbl) != out["greedy"]).float(), torch.zeros(1))
    arrays = {k: v.detach().cpu().numpy() for k, v in out.items()}
    arrays.update(sd_np("t", tr.net_t))
    np.savez_compressed(os.path.join(HERE, "tiny_prefix.npz"), **arrays)
    print("  wrote tiny_prefix.npz", sum(a.nbytes for a in arrays.values()) / 1e6, "MB raw")


def tiny_keep_first(ns):
    """Decoder context ring with `--q_keep_first --q_n_first 1` (quantized_video_model.py:896-898; the drums script uses
    n_first 8): 6 frames through a 3-slot ring.  Same seed / construction order as tiny_end_to_end, so the weights are the
    ones stored in tiny_e2e.npz."""
    opt = rh.parse_reference_options(rh.TINY_ARGV + ["--q_keep_first", "--q_n_first", "1", "--vid_len", "6"])
    qopt = opt["qvid_generator"]
    torch.manual_seed(0)
    qv = ns.qvm.QVidModel(qopt, is_train=False, is_main=True).eval()
    ref = np.load(os.path.join(HERE, "tiny_e2e.npz"))
    for k, v in qv.net_g.state_dict().items():
        if not k.endswith(".kernel"):
            assert np.array_equal(v.numpy(), ref["g/" + k]), k
    with torch.no_grad():
        qv.net_q.embedding.weight.copy_(torch.from_numpy(ref["q/embedding.weight"]))
        torch.manual_seed(21)
        vid = torch.rand(2, 6, 3, 32, 32) * 2 - 1
        code = torch.randint(0, 32, (2, 6 * 64))
        enc = qv({"vid": vid.clone()}, mode="vid_encoder")
        inter = [f[:, :1].contiguous() for f in enc["inter"]]
        with rh.patched_overlapping_shift():
            fake = qv({"code": code.clone(), "inter": [f.clone() for f in inter]}, mode="vid_decoder")["vid"]
        nets = {"e": qv.net_e.state_dict(), "q": qv.net_q.state_dict(), "g": qv.net_g.state_dict()}
        report("keep_first/vid", O.qvid_decode(nets, qopt, code, [f.clone() for f in inter]), fake)
    np.savez_compressed(os.path.join(HERE, "tiny_keepfirst.npz"), vid=vid.numpy(), code=code.numpy(), fake_vid=fake.numpy())
    print("  wrote tiny_keepfirst.npz")


def op_fixtures(ns):
    """Per-op vectors at a few real channel counts on small maps."""
    sae = ns.sae
    arrays = {}
    g = torch.Generator().manual_seed(11)

    def rnd(*s):
        return torch.randn(*s, generator=g)

    with torch.no_grad():
        # Blur / upfirdn2d (modules/upfirdn2d.py:162-203): down pad, up pad (x4), up=2
        x = rnd(2, 5, 12, 10)
        arrays["blur/x"] = x.numpy()
        k1 = sae.make_kernel([1, 3, 3, 1])
        for name, kw in {"down3": dict(pad=(2, 2)), "down1": dict(pad=(1, 1)), "up3": dict(pad=(1, 1), gain=4.0),
                         "up1": dict(pad=(2, 2), gain=4.0), "upsample2": dict(pad=(2, 1), up=2, gain=4.0)}.items():
            gain = kw.pop("gain", 1.0)
            y = ns.upfirdn2d.upfirdn2d(x, k1 * gain, **kw)
            arrays[f"blur/{name}"] = y.numpy()
            report(f"blur/{name}", O.upfirdn2d(x, O.make_fir_kernel(gain=gain), **kw), y)

        # ConvLayer variants (skip_autoencoder.py:66-102)
        torch.manual_seed(21)
        for name, (cin, cout, k, kw) in {
            "plain3": (24, 40, 3, {}), "plain1": (96, 24, 1, {}), "down3": (16, 32, 3, dict(downsample=True)),
            "down1": (16, 32, 1, dict(downsample=True, activate=False, bias=False)),
            "up3": (32, 16, 3, dict(upsample=True)), "up1": (32, 16, 1, dict(upsample=True, activate=False, bias=False)),
            "head9": (32, 2, 9, dict(activate=False)), "head5": (32, 1, 5, dict(activate=False)),
        }.items():
            m = sae.ConvLayer(cin, cout, k, **kw)
            for p in m.parameters():
                if p.ndim == 1:
                    p.normal_(0, 0.5)
            x = rnd(2, cin, 16, 16)
            y = m(x)
            arrays[f"conv/{name}/x"] = x.numpy()
            arrays[f"conv/{name}/y"] = y.numpy()
            sd = {f"m.{kk}": v for kk, v in m.state_dict().items()}
            for kk, v in sd.items():
                if not kk.endswith("kernel"):
                    arrays[f"conv/{name}/{kk}"] = v.numpy()
            okw = {kk: v for kk, v in kw.items() if kk != "bias"}
            report(f"conv/{name}", O.conv_layer(sd, "m", x, **okw), y)

        # ResBlock down / up (skip_autoencoder.py:105-117)
        for name, kw in {"down": dict(downsample=True), "up": dict(upsample=True)}.items():
            m = sae.ResBlock(16, 24, **kw)
            x = rnd(2, 16, 16, 16)
            y = m(x)
            arrays[f"res/{name}/x"] = x.numpy()
            arrays[f"res/{name}/y"] = y.numpy()
            sd = {f"m.{kk}": v for kk, v in m.state_dict().items()}
            for kk, v in sd.items():
                if not kk.endswith("kernel"):
                    arrays[f"res/{name}/{kk}"] = v.numpy()
            report(f"res/{name}", O.res_block(sd, "m", x, **kw), y)

        # backwarp (skip_autoencoder.py:120-128)
        x = rnd(2, 6, 12, 16)
        flow = rnd(2, 2, 12, 16) * 3
        grid = sae.get_backwarp_grid(12, 16)
        y = sae.backwarp(x, flow, grid)
        arrays["warp/x"], arrays["warp/flow"], arrays["warp/y"] = x.numpy(), flow.numpy(), y.numpy()
        report("warp", O.backwarp(x, flow, O.backwarp_grid(12, 16)), y)

        # VQ: default init (U(+-1/n_e), tie-prone) and scaled-normal codebook
        for name, scale_normal in {"default": False, "randn": True}.items():
            q = ns.quantize.VectorQuantizer(64, 32, beta=0.25)
            z = rnd(3, 32, 8, 8) * 0.5
            if scale_normal:
                q.embedding.weight.copy_(rnd(64, 32) * z.std())
            zq, _, info = q(z)
            arrays[f"vq/{name}/z"] = z.numpy()
            arrays[f"vq/{name}/codebook"] = q.embedding.weight.numpy()
            arrays[f"vq/{name}/idx"] = info[2].view(-1).numpy()
            arrays[f"vq/{name}/zq"] = zq.numpy()
            report(f"vq/{name} (mismatches)", (O.vq_indices(z, q.embedding.weight) != info[2].view(-1)).float(), torch.zeros(1))
            report(f"vq/{name}/zq", O.vq_quantize(z, q.embedding.weight)[0], zq)
            arrays[f"vq/{name}/embed"] = q.embed_code(info[2].view(3, 8, 8)).numpy()

        # InterBlock at decoder level 3 geometry (5x5 heads, stride-2 correlation, carried flow), k = 2
        opt = O.namespace(no_corr=False, use_masked_flow=False, use_deformed_conv=False, use_tradeoff=False, no_proj=False)
        torch.manual_seed(31)
        ib = sae.InterBlock(opt, 16, 16, flow_mult=8, kernel=5, feat_size=24, corr_stride=2, first=False)
        inp = rnd(2, 24, 16, 16)
        inters = [rnd(2, 24, 16, 16), rnd(2, 24, 16, 16)]
        flows = rnd(4, 2, 8, 8) * 0.2
        occs = rnd(4, 1, 8, 8)
        y, f2, o2, _ = ib(inp, inters, flows, occs)
        arrays["ib/inp"], arrays["ib/inter0"], arrays["ib/inter1"] = inp.numpy(), inters[0].numpy(), inters[1].numpy()
        arrays["ib/flows"], arrays["ib/occs"] = flows.numpy(), occs.numpy()
        arrays["ib/y"], arrays["ib/flows_out"], arrays["ib/occs_out"] = y.numpy(), f2.numpy(), o2.numpy()
        sd = {f"m.{kk}": v for kk, v in ib.state_dict().items()}
        for kk, v in sd.items():
            if not kk.endswith("kernel"):
                arrays[f"ib/{kk}"] = v.numpy()
        oy, of, oo = O.inter_block_forward(sd, "m", 3, 24, inp, inters, flows, occs, O.backwarp_grid(16, 16), opt)
        report("ib/y", oy, y)
        report("ib/flows", of, f2)
        report("ib/occs", oo, o2)

        # correlation: harness brute force vs oracle (parity unpinned by the reference, see docstrings)
        a, b = rnd(2, 24, 9, 11), rnd(2, 24, 9, 11)
        for s in (1, 2):
            y = rh.correlation_bruteforce(a, b, s)
            arrays[f"corr/s{s}"] = y.numpy()
            report(f"corr/s{s}", O.correlation(a, b, s), y)
        arrays["corr/a"], arrays["corr/b"] = a.numpy(), b.numpy()

        # GPT block + full GPT with p2p conditioning prefix (mingpt.py:232-305)
        torch.manual_seed(41)
        gpt = ns.mingpt.GPT(vocab_size=50, block_size=192, num_blocks=4, n_layer=2, n_head=4, n_embd=64,
                            emb_mode="temporal", shape=[4, 4])
        gpt.s_emb.normal_(0, 0.02)
        gpt.t_emb.normal_(0, 0.02)
        idx = torch.randint(0, 50, (3, 37), generator=g)
        cond = torch.randint(0, 50, (3, 16), generator=g)
        dl = torch.tensor([3, 3, 3])
        arrays["gpt/idx"], arrays["gpt/cond"], arrays["gpt/delta"] = idx.numpy(), cond.numpy(), dl.numpy()
        y0 = gpt(idx)
        y1 = gpt(idx, cond_idx=cond, delta_length_cond=dl)
        arrays["gpt/logits"], arrays["gpt/logits_p2p"] = y0.numpy(), y1.numpy()
        sd = gpt.state_dict()
        for kk, v in sd.items():
            if not kk.endswith("mask"):
                arrays[f"gpt/w/{kk}"] = v.numpy()
        cfg = O.namespace(z_shape=[4, 4], emb_mode="temporal", n_layer=2, n_head=4, z_len=192)
        report("gpt/logits", O.gpt_forward(sd, cfg, idx), y0)
        report("gpt/logits_p2p", O.gpt_forward(sd, cfg, idx, cond, dl), y1)

        # top-k masking + greedy pick (transformer_model.py:256-260,395-409)
        logits = rnd(4, 1, 50)
        tr = ns.tm.Transformer.__new__(ns.tm.Transformer)
        masked = ns.tm.Transformer.top_k_logits(tr, logits[:, -1], 7)
        icode, _ = ns.tm.Transformer.get_icode(tr, logits, 0.7, 7, False)
        arrays["topk/logits"], arrays["topk/masked"], arrays["topk/icode"] = logits.numpy(), masked.numpy(), icode.numpy()
        report("topk/masked", torch.nan_to_num(O.top_k_logits(logits[:, -1], 7), neginf=-1e30), torch.nan_to_num(masked, neginf=-1e30))

    np.savez_compressed(os.path.join(HERE, "ops.npz"), **arrays)
    print("  wrote ops.npz", sum(a.nbytes for a in arrays.values()) / 1e6, "MB raw")


if __name__ == "__main__":
    ns = rh.load_reference()
    which = sys.argv[1:] or ["ops", "tiny", "state", "statefront", "keepfirst", "statemodel", "beam", "prefix"]
    if "ops" in which:
        print("== op fixtures")
        op_fixtures(ns)
    if "tiny" in which:
        print("== tiny end-to-end")
        tiny_end_to_end(ns)
    if "state" in which:
        print("== tiny ancillary-token stream")
        tiny_state_stream(ns)
    if "prefix" in which:
        print("== tiny start / label tokens")
        tiny_prefix_tokens(ns)
    if "beam" in which:
        print("== tiny beam search")
        tiny_beam(ns)
    if "statefront" in which:
        print("== tiny state_front")
        tiny_state_front(ns)
    if "statemodel" in which:
        print("== tiny state model")
        tiny_state_model(ns)
    if "keepfirst" in which:
        print("== tiny keep_first ring")
        tiny_keep_first(ns)
