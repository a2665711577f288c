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
    # per-rank CPU placement, before anything else (bench.py does the same before its first GPU call): disjoint equal blocks
    from ccvs_amd.tools.affinity import pin_rank
    allowed = sorted(os.sched_getaffinity(0))
    cores = pin_rank(numa=False)
    with Engine(backend="gloo") as eng:
        assert eng.distributed and eng.world_size == 2
        import torch.distributed as dist
        assert cores is not None and set(cores) == os.sched_getaffinity(0)
        both = [None, None]
        dist.all_gather_object(both, cores)
        if len(allowed) >= 2:
            per = len(allowed) // 2
            assert both[0] == allowed[:per] and both[1] == allowed[per:2 * per], (both, allowed)
            assert not set(both[0]) & set(both[1])
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
        # sampling-noise derivation: one Philox key per iteration shared by all ranks, counter row = GLOBAL clip
        # index -> identical to what a single rank running the whole batch would use (world-size invariance)
        import torch.distributed as dist
        key = torch.tensor(gen.noise_key(5), dtype=torch.int64)
        keys = [torch.zeros_like(key) for _ in range(2)]
        dist.all_gather(keys, key)
        assert torch.equal(keys[0], keys[1]), "Philox key differs between ranks"
        rows = gen.first_clip(2) + torch.arange(2)
        allrows = [torch.zeros_like(rows) for _ in range(2)]
        dist.all_gather(allrows, rows)
        solo = Generator(Options().parse(True, True, argv=TINY_ARGV))          # the same job on one rank
        assert solo.noise_key(5) == gen.noise_key(5) and solo.noise_key(6) != gen.noise_key(5)
        assert torch.equal(torch.cat(allrows), solo.first_clip(4) + torch.arange(4)), "global clip rows differ from the 1-rank job"
        # host-drawn noise (the default sampler): bench.py seeds rank r's process generator with NOISE_SEED + r, so the first Exp(1)
        # block torch.multinomial would consume (transformer_model.py:395-409) differs between the ranks (round <= 5: every rank
        # NOISE_SEED itself -> rank 1's clips sampled with rank 0's noise); rank 0 keeps the single-GPU stream
        import bench
        assert bench.noise_seed_of_rank(0) == bench.NOISE_SEED and bench.noise_seed_of_rank(eng.rank) == bench.NOISE_SEED + eng.rank
        torch.manual_seed(bench.noise_seed_of_rank(eng.rank))
        block = torch.empty(2, 64).exponential_(1)
        blocks = [torch.zeros_like(block) for _ in range(2)]
        dist.all_gather(blocks, block)
        assert not torch.equal(blocks[0], blocks[1]), "the two ranks draw the same host noise"
        torch.manual_seed(bench.NOISE_SEED)
        assert torch.equal(blocks[0], torch.empty(2, 64).exponential_(1)), "rank 0 no longer draws the single-GPU stream"
        h = eng.all_gather_clips_async(clips)          # the side-stream form degrades to the blocking call on gloo
        assert torch.equal(h.wait(), want)
        t = eng.all_reduce_max(1.0 + eng.rank)
        assert t == 2.0
        eng.barrier()
        if eng.is_main:
            print("DIST_OK")


if __name__ == "__main__":
    main()
