"""Worker of tests/test_pipeline_gpu.py::test_gemm_block_tile_switch_is_bit_identical: the decode GEMMs of a few ragged shapes under
whatever CCVS_GEMM_TILE2 the parent set (the library reads the switch once per process), results to the .npz named on the
command line."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ccvs_amd import ops  # noqa: E402

g = torch.Generator().manual_seed(17)
out = {}
for m in (48, 64, 37):
    for n, k in ((200, 1024), (72, 4096), (1000, 256)):     # N not a multiple of 32 (ragged column block), deep K (split over workgroups)
        x = torch.randn(m, k, generator=g).cuda()
        w = (torch.randn(n, k, generator=g) * 0.03).cuda()
        b = torch.randn(n, generator=g).cuda()
        res = torch.randn(m, n, generator=g).cuda()
        out[f"nt_{m}_{n}_{k}"] = ops.gemm_nt(x, w, b, ops.EPI_RESIDUAL, residual=res)
        out[f"plain_{m}_{n}_{k}"] = ops.gemm_nt(x, w, b)
        gamma, beta = (1 + 0.1 * torch.randn(k, generator=g)).cuda(), (0.1 * torch.randn(k, generator=g)).cuda()
        out[f"ln_{m}_{n}_{k}"] = ops.gemm_ln(x, *ops.pack_ln_linear(w, b, gamma, beta), epilogue=ops.EPI_GELU)
torch.cuda.synchronize()
np.savez(sys.argv[1], **{k: v.cpu().numpy() for k, v in out.items()})
