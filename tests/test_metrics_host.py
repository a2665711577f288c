"""not gpu: the oracle's PSNR / SSIM restatements (tools/pytorch_metrics/metrics.py:15-25).  piq 0.5.4 and scikit-image 0.17.2 are
absent here (parity unpinned, see the oracle's header): the pins are known answers and an independent direct evaluation of the
SSIM definition, window by window in plain loops, that shares no code with the filter-based restatement."""
import math

import numpy as np
import torch

from oracle import ccvs_oracle as O


def _ssim_direct(a, b, data_range=2.0):
    """SSIM as defined (Wang et al. 2004) with the options skimage 0.17.2 defaults to: every 7 x 7 window inside the plane, sample
    (co)variances (divisor 48), K1 = 0.01, K2 = 0.03, arithmetic in float64, plain mean over the windows."""
    a, b = a.astype(np.float64), b.astype(np.float64)
    c1, c2 = (0.01 * data_range) ** 2, (0.03 * data_range) ** 2
    vals = []
    for i in range(a.shape[0] - 6):
        for j in range(a.shape[1] - 6):
            wa, wb = a[i:i + 7, j:j + 7].ravel(), b[i:i + 7, j:j + 7].ravel()
            ma, mb = wa.mean(), wb.mean()
            va, vb = ((wa - ma) ** 2).sum() / 48, ((wb - mb) ** 2).sum() / 48
            cab = ((wa - ma) * (wb - mb)).sum() / 48
            vals.append(((2 * ma * mb + c1) * (2 * cab + c2)) / ((ma * ma + mb * mb + c1) * (va + vb + c2)))
    return float(np.mean(vals))


def test_ssim_restatement_equals_the_definition():
    rng = np.random.default_rng(3)
    for hw in ((7, 7), (9, 24), (24, 20)):
        a = rng.random(hw, dtype=np.float32)
        b = np.clip(a + 0.2 * rng.standard_normal(hw).astype(np.float32), 0, 1)
        assert abs(O.ssim_plane(a, b) - _ssim_direct(a, b)) < 1e-12, hw
        assert abs(O.ssim_plane(a, b) - O.ssim_plane(b, a)) < 1e-15
    a = rng.random((16, 16), dtype=np.float32)
    assert O.ssim_plane(a, a) == 1.0
    assert abs(O.ssim_plane(a, 1 - a) - _ssim_direct(a, 1 - a)) < 1e-12     # anti-correlated: negative values
    assert O.ssim_plane(a, 1 - a) < 0
    # uint8 planes: data_range = 255 (the dtype's span)
    u, v = (a * 255).astype(np.uint8), (np.clip(a + 0.1, 0, 1) * 255).astype(np.uint8)
    assert abs(O.ssim_plane(u, v) - _ssim_direct(u, v, 255.0)) < 1e-12


def test_psnr_known_answers():
    x = torch.full((2, 3, 8, 8), 0.5)
    y = x.clone()
    y[0] += 0.1
    got = O.psnr(x, y)
    assert abs(got[0].item() - (-10 * math.log10(0.1 ** 2 + 1e-8))) < 1e-4            # 20 dB
    assert abs(got[1].item() - 80.0) < 1e-4                                           # identical images: the 1e-8 floor
    got255 = O.psnr(x * 255, y * 255, data_range=255.0)
    assert torch.allclose(got, got255, atol=1e-4)


def test_get_ssim_is_the_mean_over_images_and_channels():
    g = torch.Generator().manual_seed(0)
    x, y = torch.rand(3, 3, 12, 10, generator=g), torch.rand(3, 3, 12, 10, generator=g)
    want = np.mean([[O.ssim_plane(x[i, c].numpy(), y[i, c].numpy()) for c in range(3)] for i in range(3)])
    assert abs(O.get_ssim(x, y).item() - want) < 1e-12
    assert O.get_ssim(x, y).dtype == torch.float64


def test_upscale_rule():
    assert O.upscale(torch.zeros(1, 3, 161, 200)).shape[-2:] == (161, 200)
    assert O.upscale(torch.zeros(1, 3, 64, 64)).shape[-2:] == (161, 161)
    assert O.upscale(torch.zeros(1, 3, 64, 128)).shape[-2:] == (161, 322)
    assert O.upscale(torch.zeros(1, 3, 128, 64)).shape[-2:] == (322, 161)
