"""Worker for tests/test_distributed_cpu.py: world_size-2 `gloo` run of the sharding / gather
logic bench.py uses on RCCL (Engine.shard_batch, all_gather_clips, max-over-ranks timing)."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from ccvs_amd.tools.engine import Engine  # noqa: E402
from ccvs_amd.tools.options import Options  # noqa: E402
from ccvs_amd.helpers.generator import Generator  # noqa: E402
from tests.test_capi_host import TINY_ARGV  # noqa: E402


def main():
    with Engine(backend="gloo") as eng:
        assert eng.distributed and eng.world_size == 2
        gen = Generator(Options().parse(True, True, argv=TINY_ARGV))
        gen.engine = eng
        lo, hi = eng.shard_batch(4)
        assert (lo, hi) == (2 * eng.rank, 2 * eng.rank + 2)
        local = gen.synthetic_batch(hi - lo, seed=3, first_clip=lo)["vid"]
        # stand-in for the decoded clip: a deterministic function of the local shard, packed to uint8
        clips = ((local.clamp(-1, 1) + 1) / 2 * 255).to(torch.uint8)
        gathered = eng.all_gather_clips(clips)
        full = gen.synthetic_batch(4, seed=3)["vid"]
        want = ((full.clamp(-1, 1) + 1) / 2 * 255).to(torch.uint8)
        assert gathered.shape == want.shape and torch.equal(gathered, want), "all-gather != concatenation of per-rank outputs"
        info = gen.get_data_info("valid", "vid")
        assert info["batch_size_per_gpu"] == 1  # batch_size_vid 2 split over 2 ranks
        t = eng.all_reduce_max(1.0 + eng.rank)
        assert t == 2.0
        eng.barrier()
        if eng.is_main:
            print("DIST_OK")


if __name__ == "__main__":
    main()
