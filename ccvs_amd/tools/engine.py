"""Process / device set-up for one-process-per-GPU generation (reference tools/engine.py:17-140).

The reference initialises NCCL, wraps models in apex DDP (gradient sync: training only) and
shards the dataset with a DistributedSampler; inference then runs as pure replicas with no
collective.  Here: `torch.distributed` on RCCL (backend "nccl" on ROCm),
the generation batch sharded along B, and ONE collective -- the all-gather of the decoded clips
over xGMI (`all_gather_clips`).  A `gloo` backend is accepted for the CPU tests of the
sharding / gather logic.
"""
import os

import torch
import torch.distributed as dist


class _Gathered:
    """Result of `Engine.all_gather_clips_async`."""

    def __init__(self, tensor, event):
        self.tensor, self.event = tensor, event

    def wait(self):
        if self.event is not None:
            torch.cuda.current_stream().wait_event(self.event)
            self.event = None
        return self.tensor


class Engine(object):
    def __init__(self, opt=None, backend=None):
        self.opt = opt
        self.world_size = int(os.environ.get("WORLD_SIZE", 1))
        self.rank = int(os.environ.get("RANK", 0))
        self.local_rank = int(os.environ.get("LOCAL_RANK", getattr(opt, "local_rank", 0) if opt is not None else 0))
        self.distributed = self.world_size > 1
        self.backend = backend or ("nccl" if torch.cuda.is_available() else "gloo")
        self.devices = None
        self._own_group = False
        self._gather_stream = None
        if self.backend == "nccl":
            torch.cuda.set_device(self.local_rank)
        if self.distributed:
            if not dist.is_initialized():
                os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
                os.environ.setdefault("MASTER_PORT", "29500")
                kw = {"device_id": torch.device("cuda", self.local_rank)} if self.backend == "nccl" else {}  # eager RCCL init
                dist.init_process_group(backend=self.backend, init_method="env://", rank=self.rank, world_size=self.world_size, **kw)
                self._own_group = True
            self.devices = list(range(self.world_size))
        self.is_main = self.rank == 0

    # reference API: models are used as they are (no gradient sync at inference)
    def data_parallel(self, model):
        return model

    def shard_batch(self, global_batch):
        """[lo, hi) clip indices of this rank: global batch split along B, like
        `batch_size // world_size` in tools/engine.py:86-89."""
        assert global_batch % self.world_size == 0, "global batch must divide evenly across ranks"
        per = global_batch // self.world_size
        return self.rank * per, (self.rank + 1) * per

    def all_gather_clips(self, clips):
        """RCCL all-gather of the decoded clips [B_loc, ...] -> [world*B_loc, ...] (rank order)."""
        return self.all_gather_clips_async(clips).wait()

    def all_gather_clips_async(self, clips):
        """The same collective issued on a SIDE stream: it starts when the work already queued on the current
        stream (the uint8 pack) is done and the current stream carries on with the next batch meanwhile.  Returns a
        handle; `handle.wait()` orders the current stream after the gather and returns the gathered tensor."""
        if not self.distributed:
            return _Gathered(clips, None)
        clips = clips.contiguous()
        out = torch.empty(self.world_size * clips.shape[0], *clips.shape[1:], dtype=clips.dtype, device=clips.device)
        if self.backend != "nccl":
            dist.all_gather_into_tensor(out, clips)
            return _Gathered(out, None)
        if self._gather_stream is None:
            self._gather_stream = torch.cuda.Stream()
        side = self._gather_stream
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            dist.all_gather_into_tensor(out, clips)
            done = torch.cuda.Event()
            done.record(side)
        clips.record_stream(side)   # the allocator must not hand `clips` out again before the gather has read it
        out.record_stream(side)
        return _Gathered(out, done)

    def all_reduce_max(self, value):
        """max over ranks of a python float (benchmark timing)."""
        if not self.distributed:
            return value
        dev = torch.device("cuda", torch.cuda.current_device()) if self.backend == "nccl" else torch.device("cpu")
        t = torch.tensor([value], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    def barrier(self):
        if self.distributed:
            dist.barrier()

    def __enter__(self):
        return self

    def __exit__(self, type, value, tb):
        if self._own_group and dist.is_initialized():
            dist.destroy_process_group()
        if type is not None:
            print("An exception occurred during Engine initialization, give up running process")
            return False
