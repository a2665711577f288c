"""PSNR / SSIM of generated clips against the real ones, on the GPU that produced them
(mirror of the reference's tools/pytorch_metrics/metrics.py).

Same function names and argument meaning as the reference:

  get_psnr(x, y)      piq.psnr(x, y, data_range=1., reduction='mean')                       metrics.py:24-25 -> ccvs_psnr
  get_ssim(x, y)      mean over images of the mean over the 3 channel planes of
                      skimage.metrics.structural_similarity (scikit-image 0.17.2 defaults)  metrics.py:15-22 -> ccvs_ssim
  upscale(videos)     F.interpolate(..., mode='bilinear') up to 161 pixels                  metrics.py:115-124 -> ccvs_resize_bilinear
  metrics_from_files  the batching / aggregation of metrics.py:27-78 over a caller-supplied mp4 decoder (the reference's is
                      OpenCV, absent here); `metrics_from_videos` is the same loop on uint8 tensors [N, T, H, W, 3] such as
                      `save_video_batch` packs.  The folder / CLI glue of metrics.py:99-186 is not mirrored.
  get_lpips           NOT available: piq.LPIPS is a pretrained VGG16 plus learned linear heads whose weights are downloaded;
                      no such file exists offline and the network is not on the synthesis path.  Raises NotImplementedError;
                      the aggregate functions return None in its place.

x, y: [N, 3, H, W] fp32 CUDA tensors in [0, 1].  No CPU fallback: the kernels live in libccvs_hip.so.
"""
import torch

from ccvs_amd import ops


def get_lpips(x, y):
    raise NotImplementedError("LPIPS needs piq's pretrained VGG16 + linear-head weights (downloaded by piq.LPIPS); not available offline")


def get_ssim(x, y):
    """metrics.py:15-22.  Returns a 0-dim float64 tensor like the reference's `torch.tensor(ssim)`."""
    s = ops.ssim_planes(x, y, data_range=2.0)     # [N, 3] float64; float planes and no data_range: skimage 0.17.2 takes dmax - dmin = 2
    return s.mean(dim=1).mean().cpu()              # (sum of three planes) / 3 per image, then / N


def get_psnr(x, y):
    """metrics.py:24-25."""
    return ops.psnr(x, y, data_range=1.0).mean()


def upscale(videos, min_size=161):
    """metrics.py:115-124."""
    h, w = videos.shape[-2:]
    if h >= min_size and w >= min_size:
        return videos
    size = [min_size, int(min_size * w / h)] if h < w else [int(min_size * h / w), min_size]
    return ops.resize_bilinear(videos, size)


def _scores(real_videos, generated_videos, idx, lpips, ssim, psnr):
    """One batch of metrics.py:46-59: real_videos / generated_videos [B, T, H, W, 3] in [0, 1]."""
    if len(idx) == 0:
        real = real_videos.view(-1, *real_videos.shape[2:]).permute(0, 3, 1, 2).contiguous()
        fake = generated_videos.view(-1, *generated_videos.shape[2:]).permute(0, 3, 1, 2).contiguous()
        ssim.append(get_ssim(real, fake).cpu())
        psnr.append(get_psnr(real, fake).cpu())
    else:
        for k, frame_idx in enumerate(idx):
            real = upscale(real_videos[:, frame_idx].permute(0, 3, 1, 2).contiguous())
            fake = upscale(generated_videos[:, frame_idx].permute(0, 3, 1, 2).contiguous())
            ssim[k].append(get_ssim(real, fake).cpu())
            psnr[k].append(get_psnr(real, fake).cpu())


def _aggregate(ssim, psnr, print_256, idx, batch_size):
    """metrics.py:61-78 (LPIPS left out)."""
    if print_256 and len(idx) == 0:
        ssim = torch.stack(ssim).view(-1, 256 // batch_size)
        psnr = torch.stack(psnr).view(-1, 256 // batch_size)
        ssim_m, psnr_m = ssim.mean(), psnr.mean()
        print(f"(256) SSIM is {ssim_m} (+- {ssim.mean(1).std()}), PSNR is {psnr_m} (+- {psnr.mean(1).std()})")
    elif len(idx) == 0:
        ssim_m, psnr_m = torch.stack(ssim).mean(), torch.stack(psnr).mean()
    else:
        ssim_m = [torch.stack(e).mean() for e in ssim]
        psnr_m = [torch.stack(e).mean() for e in psnr]
    return None, ssim_m, psnr_m


def metrics_from_videos(real_videos_u8, generated_videos_u8, print_256=False, idx=(), batch_size=16, device="cuda"):
    """The loop of `metrics_from_files` on clips already in memory: uint8 [N, T, H, W, 3] (numpy or torch), batches of 16 as in the
    reference (a ragged tail is dropped like its `total_size // batch_size`).  Returns (None, ssim, psnr)."""
    idx = list(idx)
    total = len(real_videos_u8)
    assert len(generated_videos_u8) == total
    ssim, psnr = ([], []) if len(idx) == 0 else ([[] for _ in idx], [[] for _ in idx])
    with torch.no_grad():
        for i in range(total // batch_size):
            sl = slice(i * batch_size, min((i + 1) * batch_size, total))
            real = torch.as_tensor(real_videos_u8[sl]).to(device) / 255
            fake = torch.as_tensor(generated_videos_u8[sl]).to(device) / 255
            _scores(real, fake, idx, None, ssim, psnr)
    return _aggregate(ssim, psnr, print_256, idx, batch_size)


def metrics_from_files(real_video_files, generated_video_files, resize, num_workers, print_256, idx, loader=None):
    """metrics.py:27-78 with the decoding of the mp4 files left to `loader(files, resize) -> uint8 [B, T, H, W, 3]`: the reference
    decodes with OpenCV (metrics.py:80-97), which this image does not have, and writing / reading mp4 is outside the path
    (SURVEY 8 f4).  Without a loader the call raises; clips already in memory go to `metrics_from_videos`."""
    if loader is None:
        raise RuntimeError("metrics_from_files: no mp4 decoder here (the reference uses OpenCV); pass loader=..., or call "
                           "metrics_from_videos on the uint8 clips")
    batch_size = 16
    total = len(real_video_files)
    ssim, psnr = ([], []) if len(idx) == 0 else ([[] for _ in idx], [[] for _ in idx])
    with torch.no_grad():
        for i in range(total // batch_size):
            sl = slice(i * batch_size, min((i + 1) * batch_size, total))
            real = torch.as_tensor(loader(real_video_files[sl], resize)).cuda() / 255
            fake = torch.as_tensor(loader(generated_video_files[sl], resize)).cuda() / 255
            _scores(real, fake, idx, None, ssim, psnr)
    return _aggregate(ssim, psnr, print_256, idx, batch_size)
