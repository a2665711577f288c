"""-m gpu: every `scripts/*/save_videos*.sh` launch line of the reference (tests/golden/reference_launch_lines.json, written by
tests/golden/make_launch_lines.py from the reference tree) drives `Generator.generate_vid` at its FULL geometry on untrained
weights: prediction, point-to-point, state-conditioned (given / predicted), unconditional (start token), audio-conditioned
(given / predicted STFT tokens), Kinetics (V = 16384, imagenet_norm), UCF-101.  One batch of 2 clips each: shapes, token ranges,
finite pixels, and the uint8 pack of the output stage -- "a user of the reference can run each of its scripts"."""
import json
import os

import pytest
import torch

pytestmark = pytest.mark.gpu

HERE = os.path.dirname(os.path.abspath(__file__))
LINES = json.load(open(os.path.join(HERE, "golden", "reference_launch_lines.json")))
slow = pytest.mark.skipif(os.environ.get("CCVS_SKIP_SLOW", "0") == "1", reason="CCVS_SKIP_SLOW=1")


@slow
@pytest.mark.parametrize("script", sorted(LINES))
def test_reference_launch_line_runs(script):
    from ccvs_amd.tools.options import Options
    from ccvs_amd.helpers.generator import Generator
    from ccvs_amd import ops
    argv, skip = [], False
    for a in LINES[script]:              # the checkpoints the scripts name do not exist here: untrained weights (the reference's own
        if skip:                         # behaviour without --*_load_path), everything else of the launch line as it is
            skip = False
        elif a.endswith("_load_path") or a.endswith("_which_iter"):
            skip = True
        else:
            # (sic) the Drums scripts pass --dataset "drum", which matches no preset of tools/options.py:431-439 ("drums": square frames,
            # 30 fps); with the default aspect ratio 2 the 8 x 8 token grid of --q_z_shape cannot hold: the preset's name is used
            argv.append("drums" if a == "drum" else a)
    opt = Options().parse(load_qvid_generator=True, load_transformer=True, load_state_estimator=True, load_stft_ae=True,
                          argv=argv + ["--x_sample_noise", "device", "--rec_pass", "false"])
    xopt, qopt = opt["transformer"], opt["qvid_generator"]
    if xopt.vid_len > 20:                 # Drums: 45 frames = 30 slides of the 1280-token window; three slides are enough here
        xopt.vid_len = qopt.vid_len = 18
        for key in ("state_estimator", "stft_ae"):
            if opt.get(key) is not None:
                opt[key].vid_len = 18
    torch.manual_seed(0)
    gen = Generator(opt).build_models()
    data = gen.synthetic_batch(2, seed=5)
    h = xopt.max_dim
    assert data["vid"].shape == (2, xopt.vid_len, 3, h, int(h * xopt.aspect_ratio))
    if xopt.stft:
        data["stft"] = torch.rand(2, xopt.vid_len, 1, 64, 16, generator=torch.Generator().manual_seed(6)) * 2 - 1
    out = gen.generate_vid(data)
    fake = out["fake"]
    size = qopt.z_shape[0] * qopt.z_shape[1]
    assert fake["vid"].shape == data["vid"].shape and torch.isfinite(fake["vid"]).all()
    n_frames = xopt.vid_len - (1 if xopt.p2p else 0)
    assert fake["code"].shape == (2, n_frames * size)
    assert int(fake["code"].min()) >= 0 and int(fake["code"].max()) < xopt.z_num
    assert out["enc_code"].shape == (2, xopt.vid_len * size)
    if xopt.state or xopt.stft:
        sc = fake["state_code"]
        assert sc is not None and sc.shape[0] == 2 and int(sc.max()) < xopt.state_num
    if xopt.cond_len > 0:                 # the conditioning tokens pass through unchanged
        assert torch.equal(fake["code"][:, :xopt.cond_len], out["enc_code"][:, :xopt.cond_len])
    if xopt.imagenet_norm:
        u8 = ops.pack_u8_norm(fake["vid"], (0.229, 0.224, 0.225), (0.485, 0.456, 0.406))
    else:
        u8 = ops.pack_u8(fake["vid"])
    assert u8.dtype == torch.uint8 and u8.shape == (2, xopt.vid_len, h, int(h * xopt.aspect_ratio), 3)
