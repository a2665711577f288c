"""-m gpu: the evaluation metrics on the produced clips (tools/pytorch_metrics/metrics.py:15-25, 115-124) --
ccvs_psnr / ccvs_ssim / ccvs_resize_bilinear through the C ABI against the oracle's restatements."""
import numpy as np
import pytest
import torch

from oracle import ccvs_oracle as O
from tests.test_e2e_gpu import tiny  # noqa: F401  (fixture)

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def tiny_generator_clips(tiny):
    """real / fake uint8 clips [N, T, H, W, 3] of one tiny generate_vid call, packed by the output stage (helpers/generator.py:306-309)."""
    from ccvs_amd import ops as _ops
    from ccvs_amd.helpers.generator import Generator
    xopt = tiny["xopt"]
    xopt.sample = False
    gen = Generator(tiny["opt"])
    gen.vid_model, gen.transformer_model = tiny["qv"], tiny["tr"]
    data = gen.synthetic_batch(2, seed=71)
    out = gen.generate_vid({"vid": data["vid"].clone()})
    return _ops.pack_u8(data["vid"].cuda()).cpu(), _ops.pack_u8(out["fake"]["vid"]).cpu()


@pytest.fixture(scope="module")
def ops():
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    from ccvs_amd import ops as _ops
    return _ops


@pytest.fixture(scope="module")
def M():
    from ccvs_amd.tools.pytorch_metrics import metrics
    return metrics


def _pair(shape, seed, noise=0.1):
    g = torch.Generator().manual_seed(seed)
    x = torch.rand(*shape, generator=g)
    y = (x + noise * torch.randn(*shape, generator=g)).clamp(0, 1)
    return x, y


@pytest.mark.parametrize("hw", [(7, 7), (8, 39), (37, 70), (64, 64), (161, 161), (256, 256)])
def test_ssim_planes_vs_oracle(ops, hw):
    """Every 2-D plane: float64 result within 1e-12 of the filter-based restatement (same window sums, other summation order);
    ragged sizes exercise partial tiles, 7 x 7 is one window."""
    x, y = _pair((2, 3) + hw, hw[0] * 1000 + hw[1])
    got = ops.ssim_planes(x.cuda(), y.cuda()).cpu()
    assert got.dtype == torch.float64 and got.shape == (2, 3)
    for i in range(2):
        for c in range(3):
            assert abs(got[i, c].item() - O.ssim_plane(x[i, c].numpy(), y[i, c].numpy())) < 1e-12, (i, c)
    same = ops.ssim_planes(x.cuda(), x.cuda()).cpu()
    assert torch.equal(same, torch.ones_like(same))
    again = ops.ssim_planes(x.cuda(), y.cuda()).cpu()
    assert torch.equal(again, got), "no atomics: identical bits on every run"


def test_ssim_errors_and_data_range(ops):
    x, y = _pair((1, 6, 20), 5)
    with pytest.raises(Exception, match="7 x 7"):
        ops.ssim_planes(x.cuda(), y.cuda())
    x, y = _pair((4, 20, 20), 6)
    got = ops.ssim_planes((x * 255).floor().cuda(), (y * 255).floor().cuda(), data_range=255.0).cpu()
    for i in range(4):
        want = O.ssim_plane((x[i] * 255).floor().numpy(), (y[i] * 255).floor().numpy(), data_range=255.0)
        assert abs(got[i].item() - want) < 1e-12


def test_psnr_vs_oracle(ops, M):
    for shape, noise in (((5, 3, 64, 64), 0.1), ((3, 3, 37, 41), 0.01), ((2, 1, 9, 9), 0.5)):   # 37 x 41 x 3: unaligned images, scalar tail
        x, y = _pair(shape, shape[2], noise)
        got = ops.psnr(x.cuda(), y.cuda()).cpu()
        assert torch.allclose(got, O.psnr(x, y), atol=2e-5, rtol=0), (got, O.psnr(x, y))
    x = torch.rand(2, 3, 16, 16)
    assert torch.allclose(ops.psnr(x.cuda(), x.cuda()).cpu(), torch.full((2,), 80.0), atol=1e-4)      # the 1e-8 floor
    x, y = _pair((4, 3, 32, 32), 9)
    assert abs(M.get_psnr(x.cuda(), y.cuda()).item() - O.psnr(x, y).mean().item()) < 2e-5


def test_resize_bilinear_vs_torch(ops, M):
    """`upscale` (metrics.py:115-124): torch's bilinear, align_corners False."""
    for shape, size in (((2, 3, 64, 64), (161, 161)), ((1, 3, 64, 128), (161, 322)), ((2, 3, 100, 75), (214, 161)), ((1, 1, 5, 7), (5, 7))):
        x = torch.rand(*shape, generator=torch.Generator().manual_seed(shape[-1]))
        got = ops.resize_bilinear(x.cuda(), size).cpu()
        want = torch.nn.functional.interpolate(x, size=list(size), mode="bilinear")
        assert (got - want).abs().max() < 1e-5, (shape, size)     # fp32 source coordinates: 64 / 161 * (o + 0.5) rounds differently on the two sides
    x = torch.rand(2, 3, 100, 75)
    assert M.upscale(x.cuda()).shape == O.upscale(x).shape == (2, 3, 214, 161)
    big = torch.rand(1, 3, 161, 170).cuda()
    assert M.upscale(big) is big


def test_metrics_from_videos_vs_oracle_loop(M):
    """The aggregate of metrics.py:27-78 on uint8 clips [N, T, H, W, 3]: batches of 16 clips, every frame (idx = []) and per frame
    index with the up-scaling to 161 pixels (idx given), against the same loop over the oracle's functions."""
    g = torch.Generator().manual_seed(4)
    real = torch.randint(0, 256, (32, 3, 24, 24, 3), generator=g, dtype=torch.uint8)
    fake = (real.float() + 12 * torch.randn(real.shape, generator=g)).clamp(0, 255).to(torch.uint8)
    lp, ssim, psnr = M.metrics_from_videos(real.numpy(), fake.numpy())
    assert lp is None
    want_s, want_p = [], []
    for i in range(2):
        r = (real[16 * i:16 * i + 16] / 255).view(-1, 24, 24, 3).permute(0, 3, 1, 2)
        f = (fake[16 * i:16 * i + 16] / 255).view(-1, 24, 24, 3).permute(0, 3, 1, 2)
        want_s.append(O.get_ssim(r, f))
        want_p.append(O.psnr(r, f).mean())
    assert abs(ssim.item() - torch.stack(want_s).mean().item()) < 1e-8      # torch's `/ 255` on the device is not the CPU's correctly rounded division
    assert abs(psnr.item() - torch.stack(want_p).mean().item()) < 2e-5
    _, ssim_k, psnr_k = M.metrics_from_videos(real, fake, idx=[0, 2])
    for k, t in enumerate([0, 2]):
        r = O.upscale((real[:16, t] / 255).permute(0, 3, 1, 2))
        f = O.upscale((fake[:16, t] / 255).permute(0, 3, 1, 2))
        r2 = O.upscale((real[16:, t] / 255).permute(0, 3, 1, 2))
        f2 = O.upscale((fake[16:, t] / 255).permute(0, 3, 1, 2))
        assert abs(ssim_k[k].item() - (O.get_ssim(r, f) + O.get_ssim(r2, f2)).item() / 2) < 1e-5      # the fp32 resize differs in the last bit
        assert abs(psnr_k[k].item() - (O.psnr(r, f).mean() + O.psnr(r2, f2).mean()).item() / 2) < 1e-4


def test_unavailable_parts_raise(M):
    with pytest.raises(NotImplementedError):
        M.get_lpips(None, None)
    with pytest.raises(RuntimeError, match="decoder"):
        M.metrics_from_files(["a.mp4"] * 16, ["b.mp4"] * 16, None, 1, False, [])


def test_generated_clips_score_against_their_inputs(tiny_generator_clips, M):
    """End of the path: the uint8 clips `save_video_batch` packs (fake vs real) go straight into the metrics -- values finite,
    PSNR of a clip against itself at the floor, SSIM 1."""
    real, fake = tiny_generator_clips
    _, ssim, psnr = M.metrics_from_videos(real, fake, batch_size=real.shape[0])
    assert np.isfinite(ssim.item()) and np.isfinite(psnr.item()) and -1 <= ssim.item() <= 1
    _, ssim1, psnr1 = M.metrics_from_videos(real, real, batch_size=real.shape[0])
    assert ssim1.item() == 1.0 and abs(psnr1.item() - 80.0) < 1e-3
