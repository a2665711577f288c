"""not gpu, build container only: the oracle against the REAL reference imported through
tests/golden/ref_harness.py (skipped where /root/reference does not exist, e.g. the GPU box)."""
import os
import sys

import pytest
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "golden"))
import ref_harness as rh  # noqa: E402
from oracle import ccvs_oracle as O  # noqa: E402

pytestmark = pytest.mark.skipif(not rh.reference_available(), reason="upstream reference tree not present")


@pytest.fixture(scope="module")
def ref():
    return rh.load_reference()


def test_encoder_decoder_fresh_weights(ref):
    """New seed, new input: reference modules vs the oracle driven by their state dicts."""
    opt = rh.parse_reference_options(rh.TINY_ARGV)["qvid_generator"]
    torch.manual_seed(123)
    enc, dec = ref.sae.SkipGANEncoder(opt), ref.sae.SkipGANDecoder(opt)
    vid = torch.rand(1, 2, 3, 32, 32) * 2 - 1
    with torch.no_grad():
        z, inter = enc(vid)
        oz, ointer = O.encoder_forward(enc.state_dict(), opt, vid)
        assert torch.equal(z, oz) and all(torch.equal(a, b) for a, b in zip(inter, ointer))
        ctx = [[f[:, :1] for f in inter], [f[:, 1:] for f in inter]]
        rgb, _, flows, occs, _ = dec(z[:, :1].contiguous(), ctx, return_all=True, inter_pre_warping=False)
        orgb, oflows, ooccs = O.decoder_forward(dec.state_dict(), opt, z[:, :1].contiguous(), ctx, return_all=True)
        assert (rgb - orgb).abs().max() < 1e-6
        for a, b in zip(flows + occs, oflows + ooccs):
            assert (a - b).abs().max() < 1e-6


def test_gpt_spatio_temporal_and_flat_positions(ref):
    torch.manual_seed(5)
    for mode in ("spatio-temporal", None):
        gpt = ref.mingpt.GPT(vocab_size=30, block_size=48, num_blocks=3, n_layer=1, n_head=2, n_embd=32, emb_mode=mode, shape=[4, 4])
        with torch.no_grad():
            for n, p in gpt.named_parameters():
                if n.endswith("_emb"):
                    p.normal_(0, 0.1)
            idx = torch.randint(0, 30, (2, 21))
            cfg = O.namespace(z_shape=[4, 4], emb_mode=mode, n_layer=1, n_head=2, z_len=48)
            assert (gpt(idx) - O.gpt_forward(gpt.state_dict(), cfg, idx)).abs().max() < 1e-6


def test_output_stage_vs_reference_save_video_batch(ref, tmp_path):
    """The oracle's output stage (clamp / rescale / x255 / uint8, the imagenet de-normalisation branch, the 3x3 state marker)
    against the reference's own `save_video_batch` (helpers/generator.py:285-333), with `torchvision.io.write_video` replaced
    by a recorder: byte for byte, including a marker clipped at the frame border."""
    import importlib
    import torchvision
    g = importlib.import_module("helpers.generator")
    got = []
    torchvision.io.write_video = lambda fn, v, fps: got.append((os.path.basename(fn), v.clone()))
    g.torchvision = torchvision
    vid = torch.rand(2, 3, 3, 64, 64, generator=torch.Generator().manual_seed(1)) * 2.4 - 1.2
    state = torch.tensor([[[0.5, 0.25]] * 3, [[0.999, 0.0]] * 3])      # the second marker sits in a corner: clipped cross
    g.save_video_batch(vid.clone(), 2, 3, str(tmp_path), 4, True, False, [-1, 1], "bair", state=state)
    assert [n for n, _ in got] == ["vid_00006.mp4", "vid_00007.mp4"]
    assert torch.equal(torch.stack([v for _, v in got]), O.mark_state(O.pack_u8(vid), state, "bair"))
    got.clear()
    g.save_video_batch(vid.clone(), 2, 0, str(tmp_path), 4, True, False, [-1, 1], "bairhd")
    assert torch.equal(torch.stack([v for _, v in got]), O.pack_u8(vid))
    got.clear()
    g.save_video_batch(vid.clone(), 2, 0, str(tmp_path), 4, True, True, [-1, 1], "kinetics600")
    assert torch.equal(torch.stack([v for _, v in got]), O.pack_u8_imagenet(vid))


def _script_argv(path):
    import re
    import shlex
    body = open(path).read().split("helpers/generator.py", 1)[1].replace("\\\n", " ")
    return shlex.split(re.sub(r"\$\{GPU_IDS\}", "0", body))


def test_every_reference_save_script_parses_like_the_reference(ref):
    """The flags of all nine `scripts/*/save_videos*.sh` launch lines: the product's `Options` must accept them (training-only
    flags ignored) and give the fields the synthesis path reads the values the reference's own parser gives."""
    import glob
    from ccvs_amd.tools.options import Options
    scripts = sorted(glob.glob(os.path.join(rh.REF_ROOT, "scripts", "*", "save_videos*.sh")))
    assert len(scripts) >= 9
    x_fields = ["vid_len", "cond_len", "z_len", "z_chunk", "z_num", "z_shape", "state_size", "state_num", "num_blocks", "top_k", "temperature",
                "sample", "p2p", "stft", "state", "keep_state", "use_start_token", "cat", "n_layer", "n_head", "n_embd", "emb_mode",
                "sample_state", "top_k_state", "temperature_state", "batch_size_vid", "max_dim", "aspect_ratio", "fps", "imagenet_norm"]
    q_fields = ["necf", "necf_mult", "z_num", "z_size", "z_shape", "skip_context", "skip_memory", "keep_first", "n_first", "use_inter", "inter_p",
                "max_dim", "aspect_ratio", "vid_len", "use_ema"]
    for path in scripts:
        argv = _script_argv(path)
        want = rh.parse_reference_options(argv)
        got = Options().parse(load_qvid_generator=True, load_transformer=True, load_state_estimator=True, load_stft_ae=True, argv=argv)
        for key, fields in (("transformer", x_fields), ("qvid_generator", q_fields)):
            for f in fields:
                if hasattr(want[key], f):
                    assert getattr(got[key], f) == getattr(want[key], f), (os.path.basename(path), key, f, getattr(got[key], f), getattr(want[key], f))
        if want.get("stft_ae") is not None and "--x_stft" in argv:
            for f in ("stft_num", "stft_size", "stft_hsize", "stft_shape"):
                assert getattr(got["stft_ae"], f) == getattr(want["stft_ae"], f), (path, f)
