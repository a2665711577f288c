"""not gpu: host-side logic of the path that needs no device -- the sliding token windows of `Transformer.generate_fake`
(transformer_model.py:301-326), the token-group sizing of `Generator.run_pipelined`, the merged frame / ancillary token order of
`GPT.stream_kinds` (mingpt.py:246-282) against the oracle's merge, and the reference launch-line presets."""
import os
import types

import pytest
import torch

from oracle import ccvs_oracle as O


def _transformer_stub(z_len, z_chunk):
    from ccvs_amd.models.skip_vid_generator.models.transformer_model import Transformer
    t = Transformer.__new__(Transformer)          # no networks: only the window arithmetic is exercised
    t.opt = types.SimpleNamespace(z_len=z_len, z_chunk=z_chunk)
    return t


@pytest.mark.parametrize("z_len,z_chunk,total", [(1280, 80, 3600), (1024, 64, 1024 + 64 * 3 + 10), (256, 66, 300), (192, 64, 193)])
def test_token_windows_match_the_reference_loop(z_len, z_chunk, total):
    """`token_windows` against the reference's own bookkeeping (curr_len / add_len / i of transformer_model.py:301-326)."""
    want, curr, i = [], z_len, 1
    while curr < total:
        add = total - curr if total - curr < z_chunk else None
        want.append((i, add))
        curr += add if add is not None else z_chunk
        i += 1
    assert list(_transformer_stub(z_len, z_chunk).token_windows(total)) == want
    assert sum(z_chunk if a is None else a for _, a in want) == total - z_len


def test_token_group_size_rules(monkeypatch):
    """Stacked rows stay within the row cap (128 by default, CCVS_PIPELINE_MAX_ROWS; never more than one decode step's 256); beam
    search is never stacked; host-drawn noise is stacked when a batch's draws are one plain stream of [B, V] blocks (no ancillary
    token stream, no sliding window): each batch of the group then reads its own pre-drawn stream."""
    from ccvs_amd.helpers.generator import Generator
    g = Generator.__new__(Generator)
    g.opt = types.SimpleNamespace(sample=True, beam_size=None, state=False, stft=False, vid_len=16, z_len=1024, cond_len=64, p2p=False, z_num=1024)
    g.qvid_opt = types.SimpleNamespace(z_shape=[8, 8])
    g.transformer_model = types.SimpleNamespace(sample_noise="device")
    monkeypatch.delenv("CCVS_PIPELINE_MAX_ROWS", raising=False)
    assert g._token_group_size(16, 3) == 3 and g._token_group_size(16, 4) == 4 and g._token_group_size(16, 12) == 8
    assert g._token_group_size(64, 3) == 2 and g._token_group_size(5, 3) == 3 and g._token_group_size(100, 3) == 1 and g._token_group_size(200, 3) == 1
    monkeypatch.setenv("CCVS_PIPELINE_MAX_ROWS", "64")
    assert g._token_group_size(16, 8) == 4 and g._token_group_size(64, 3) == 1
    monkeypatch.setenv("CCVS_PIPELINE_MAX_ROWS", "1024")
    assert g._token_group_size(16, 32) == 16          # one decode step takes 256 rows
    monkeypatch.delenv("CCVS_PIPELINE_MAX_ROWS")
    g.transformer_model.sample_noise = "host"
    assert g._host_noise_streams_ok() and g._token_group_size(16, 3) == 3   # one pre-drawn generator stream per batch, in the reference's order
    g.opt.vid_len = 45                                # the token window slides (Drums): several fill_code calls per batch
    assert not g._host_noise_streams_ok() and g._token_group_size(16, 3) == 1
    g.opt.vid_len, g.opt.stft = 16, True              # ancillary picks draw blocks of another width
    assert not g._host_noise_streams_ok() and g._token_group_size(16, 3) == 1
    g.opt.stft, g.opt.use_graph = False, False        # eager steps draw for themselves
    assert g._token_group_size(16, 3) == 1
    g.opt.use_graph = True
    g.opt.z_num, g.opt.cond_len = 16384, 320          # Kinetics: 704 x 64 x 16384 floats = 3 GB of noise per batch -- not held, drawn pick by pick
    assert g._host_noise_streams_ok() and not g._host_noise_streams_ok(64) and g._token_group_size(64, 2) == 1 and g._host_noise_streams_ok(8)
    g.opt.z_num, g.opt.cond_len = 1024, 64
    g.opt.sample = False
    assert g._token_group_size(16, 3) == 3            # greedy: nothing is drawn
    g.opt.beam_size = 4
    assert g._token_group_size(16, 3) == 1


def test_lanes_that_fit_the_device_memory():
    """Batches in flight = lanes x (chains + 2); the lanes shrink until they fit 0.7 of the device memory."""
    from ccvs_amd.helpers.generator import Generator
    total = 288 * 2 ** 30
    assert Generator._lanes_that_fit(4, 2, 16, 256, 256, total) == 4          # the benchmark: 16 batches, ~168 GB
    assert Generator._lanes_that_fit(4, 2, 32, 256, 256, total) == 2          # twice the clips: 8 batches
    assert Generator._lanes_that_fit(4, 2, 64, 256, 256, total) == 1
    assert Generator._lanes_that_fit(2, 3, 64, 64, 64, total) == 2            # Kinetics: small frames
    assert Generator._lanes_that_fit(4, 2, 16, 256, 256, 64 * 2 ** 30) == 1   # a smaller device
    assert Generator._lanes_that_fit(1, 2, 512, 256, 256, total) == 1         # never below one
    assert Generator._lanes_that_fit(4, 2, 16, 256, 256, total, taken=60 * 2 ** 30) == 3   # another process holds 60 GB of the device
    assert Generator._lanes_that_fit(4, 2, 16, 256, 256, total, dec_streams=4) == 4 and Generator._lanes_that_fit(4, 2, 16, 256, 256, 235 * 2 ** 30, dec_streams=6) == 3
    assert Generator._lanes_that_fit(4, 2, 16, 256, 256, 235 * 2 ** 30, dec_streams=2) == 4
    # the pre-drawn host noise stream of every batch in flight: BAIR's 63 MB changes nothing, 1 GB per batch does
    assert Generator._lanes_that_fit(4, 2, 16, 256, 256, total, noise_bytes_per_batch=960 * 16 * 1024 * 4.0) == 4
    assert Generator._lanes_that_fit(4, 2, 16, 256, 256, 235 * 2 ** 30, noise_bytes_per_batch=2.0 ** 30) == 3


def test_frames_the_encoder_has_to_see():
    """`--encode_all false`: only the frames the conditioning crop keeps (helpers/generator.py:93-99) -- and only when nothing else
    reads the rest of the clip; the default is the reference's encode of every frame."""
    from ccvs_amd.helpers.generator import Generator
    g = Generator.__new__(Generator)
    base = dict(encode_all=False, rec_pass=False, rec_only=False, state=False, stft=False, p2p=False, gen_from_img=False, cond_len=64, vid_len=16)
    g.opt = types.SimpleNamespace(**base)
    assert g._frames_to_encode(16, 64) == 1                                   # BAIR: one conditioning frame of 64 tokens
    g.opt.cond_len = 320
    assert g._frames_to_encode(16, 64) == 5                                   # Kinetics: five
    g.opt.cond_len = 100
    assert g._frames_to_encode(16, 64) == 2                                   # a partial frame of tokens keeps its frame
    g.opt.cond_len = 0
    assert g._frames_to_encode(16, 64) == 1                                   # unconditional: the encoder still needs an input
    g.opt.cond_len = 64
    for key in ("encode_all", "rec_pass", "rec_only", "state", "stft", "p2p", "gen_from_img"):
        g.opt = types.SimpleNamespace(**dict(base, **{key: True}))
        assert g._frames_to_encode(16, 64) == 16, key                          # someone reads the rest of the clip: the reference's encode
    g.opt = types.SimpleNamespace(**{k: v for k, v in base.items() if k not in ("encode_all", "rec_pass")})
    assert g._frames_to_encode(16, 64) == 16                                   # options from a reference launch line: reference behaviour


@pytest.mark.parametrize("n_code,n_state", [(64 * 3, 16 * 3), (64 * 2 + 5, 16 * 3), (64 * 15 + 5, 16 * 16), (64, 16)])
def test_stream_kinds_equal_the_oracle_merge(n_code, n_state):
    """The merged sequence order the KV-cached engine walks (`GPT.stream_kinds`) is the oracle's `gpt_merge_state` order
    (mingpt.py:258-282): checked by merging index-carrying embeddings."""
    from ccvs_amd.models.skip_vid_generator.models.mingpt import GPT
    net = GPT.__new__(GPT)
    net.config = types.SimpleNamespace(shape=[8, 8], state_size=16, num_blocks=16, state_front=False)
    kinds = GPT.stream_kinds(net, n_code, n_state)
    cfg = O.namespace(z_shape=[8, 8], state_size=16, state_front=False)
    emb = torch.arange(n_code, dtype=torch.float32).view(1, n_code, 1)                 # frame token i carries +i
    semb = -(torch.arange(n_state, dtype=torch.float32) + 1).view(1, n_state, 1)        # ancillary token i carries -(i+1)
    merged = O.gpt_merge_state(cfg, emb, semb)[0, :, 0].tolist()
    got = [float(i) if k == 0 else -(float(i) + 1) for k, i in kinds]
    assert got == merged


def test_reference_launch_line_presets():
    from ccvs_amd.tools.options import Options, DRUMS_ARGV, BAIR_P2P_ARGV, KINETICS_ARGV
    d = Options().parse(load_qvid_generator=True, load_transformer=True, load_stft_ae=True, argv=list(DRUMS_ARGV))
    x, q, a = d["transformer"], d["qvid_generator"], d["stft_ae"]
    assert (x.vid_len, x.cond_len, x.z_len, x.z_chunk, x.state_size, x.num_blocks) == (45, 960, 1280, 80, 16, 16) and x.stft and x.keep_state
    assert q.keep_first and q.n_first == 8 and q.max_dim == 128 and q.aspect_ratio == 1 and a.stft_shape == [8, 2] and a.stft_hsize == 512
    p = Options().parse(load_qvid_generator=True, load_transformer=True, argv=list(BAIR_P2P_ARGV))["transformer"]
    assert p.p2p and p.vid_len == 16 and p.cond_len == 64
    k = Options().parse(load_qvid_generator=True, load_transformer=True, argv=list(KINETICS_ARGV))["transformer"]
    assert k.z_num == 16384 and k.cond_len == 320


def test_frame_feed_hands_over_whole_frames(monkeypatch):
    """`_FrameFeed` (run_pipelined): the token loop reports how many columns of its buffer are final; every frame those complete is
    copied out once, gets an event, and its flag is set -- partial frames wait, release() wakes whoever waits for the rest."""
    import torch
    from ccvs_amd.helpers import generator as G

    class FakeEvent:
        n = 0

        def record(self):
            FakeEvent.n += 1

    monkeypatch.setattr(torch.cuda, "Event", FakeEvent)
    feed = G._FrameFeed(rows=3, frames=4, frame_tokens=5, device="cpu")
    live = torch.arange(3 * 20).view(3, 20)                 # the loop's own buffer, final up to column n
    feed.on_tokens(6, live)                                  # frame 0 complete, one token of frame 1
    assert [f.is_set() for f in feed.flags] == [True, False, False, False] and feed.sent == 1 and FakeEvent.n == 1
    assert torch.equal(feed.codes[:, :5], live[:, :5])
    feed.on_tokens(9, live)                                  # still inside frame 1: nothing happens
    assert feed.sent == 1 and FakeEvent.n == 1
    feed.on_tokens(16, live)                                 # frames 1 and 2 at once: one copy, one event for both
    assert [f.is_set() for f in feed.flags] == [True, True, True, False] and FakeEvent.n == 2
    assert feed.events[1] is feed.events[2] and feed.events[0] is not feed.events[1]
    assert torch.equal(feed.codes[:, :15], live[:, :15])
    feed.on_tokens(400, live)                                # more columns than the clip has frames: capped
    assert feed.sent == 4 and torch.equal(feed.codes, live)
    short = G._FrameFeed(rows=1, frames=4, frame_tokens=5, device="cpu")
    short.on_tokens(5, live[:1])
    short.release()                                          # a failed stage: the flags wake the waiter, the events say "never came"
    assert all(f.is_set() for f in short.flags) and short.events[0] is not None and short.events[1:] == [None] * 3


def test_decode_frames_asks_for_tokens_frame_by_frame():
    """`QVidModel.decode_frames` yields, before each piece of work, how many leading frames of tokens it is about to read: first the
    conditioning frames, then one more per synthesized frame -- and reads exactly those through `code_of`."""
    import torch
    from ccvs_amd.models.skip_vid_generator.models.quantized_video_model import QVidModel

    class Stub(QVidModel):
        def __init__(self):
            torch.nn.Module.__init__(self)
            self.opt = types.SimpleNamespace(vid_len=5, skip_memory=3, skip_context=[1, 2, 3], skip_mode="enc", z_shape=(2, 2), z_size=4,
                                             n_first=1, keep_first=False)
            self.asked = []

        def _embed(self, code, frames):
            return code.float().view(code.size(0), frames, 1, 2, 2)

        def net_g(self, z, inters, has_ctx=True):
            return z.sum(dim=(2, 3, 4)).view(z.size(0), z.size(1), 1, 1, 1).expand(-1, -1, 3, 2, 2).clone(), None

        def encode(self, data, layout, dtype, log, suffix, global_iter, quantize=True):
            return {"inter": [torch.zeros(data.size(0), 1, 2, 2, 2)]}

    m = Stub()
    code = torch.arange(2 * 5 * 4).view(2, 20)
    inter = [torch.zeros(2, 2, 2, 2, 2)]                       # two conditioning frames

    def code_of(lo, hi):
        m.asked.append((lo, hi))
        return code[:, lo * 4:hi * 4]

    gen = m.decode_frames(code_of, inter, [])
    needs = []
    while True:
        try:
            needs.append(next(gen))
        except StopIteration as fin:
            vid = fin.value
            break
    assert needs == [2, 3, 4, 5] and m.asked == [(0, 2), (2, 3), (3, 4), (4, 5)]
    assert vid.shape == (2, 5, 3, 2, 2)
    assert torch.equal(vid[:, :, 0, 0, 0], code.float().view(2, 5, 4).sum(-1))


def test_rank_cpu_sets():
    """Per-rank core blocks (ccvs_amd/tools/affinity.py): contiguous equal blocks of the allowed cores; NUMA-local blocks when the
    GPU -> node map is known (the ranks of a node split that node's cores); never empty."""
    from ccvs_amd.tools import affinity as A
    assert A._parse_cpulist("0-3,8,10-11\n") == [0, 1, 2, 3, 8, 10, 11]
    allowed = list(range(16))
    assert [A.rank_cpu_set(r, 4, allowed) for r in range(4)] == [[0, 1, 2, 3], [4, 5, 6, 7], [8, 9, 10, 11], [12, 13, 14, 15]]
    assert A.rank_cpu_set(0, 1, allowed) == allowed
    assert A.rank_cpu_set(5, 8, [0, 1, 2]) == [2]                       # fewer cores than ranks: shared round-robin
    text = "GPU[0]\t\t: (Topology) Numa Node: 0\nGPU[0]\t\t: (Topology) Numa Affinity: 0\nGPU[1]\t\t: (Topology) Numa Node: 0\n" \
           "GPU[2]\t\t: (Topology) Numa Node: 1\nGPU[3]\t\t: (Topology) Numa Node: 1\n"
    gpu_node = A.parse_showtoponuma(text)
    assert gpu_node == {0: 0, 1: 0, 2: 1, 3: 1}
    node_cpus = {0: list(range(0, 8)) + list(range(16, 24)), 1: list(range(8, 16)) + list(range(24, 32))}
    sets = [A.rank_cpu_set(r, 4, list(range(32)), gpu_node, node_cpus) for r in range(4)]
    assert sets[0] == [0, 1, 2, 3, 4, 5, 6, 7] and sets[1] == [16, 17, 18, 19, 20, 21, 22, 23]
    assert sets[2] == [8, 9, 10, 11, 12, 13, 14, 15] and sets[3] == [24, 25, 26, 27, 28, 29, 30, 31]
    assert A.rank_cpu_set(2, 4, list(range(32)), {0: 0, 1: 0}, node_cpus) == list(range(16, 24))   # incomplete map: contiguous blocks
    assert A.parse_showtoponuma("garbage") == {}


def test_noise_feed_streams_follow_the_generator(monkeypatch):
    """`NoiseFeed` (ccvs_amd/helpers/pipeline.py): requests served by parallel drawer threads behind a skipper that counts every
    stream's values off on a copy of the generator == the same blocks drawn one after the other from the generator itself (what
    torch.multinomial does, pick after pick, batch after batch); the process generator ends where the serial loop would leave it;
    the single-thread fallback gives the same.  (The device pieces -- pinned memory, copy stream, events -- are stubbed: CPU test.)"""
    import contextlib
    import torch
    from ccvs_amd.helpers import pipeline as P

    class _Stub:
        def __init__(self, *a, **k):
            pass

        def record(self):
            pass

    real_empty = torch.empty
    monkeypatch.setattr(torch.cuda, "Stream", _Stub)
    monkeypatch.setattr(torch.cuda, "Event", _Stub)
    monkeypatch.setattr(torch.cuda, "stream", lambda s: contextlib.nullcontext())
    monkeypatch.setattr(torch.cuda, "set_device", lambda d: None)
    monkeypatch.setattr(torch, "empty", lambda *a, **k: real_empty(*a, **{kk: v for kk, v in k.items() if kk != "pin_memory"}))
    ref = torch.Generator().manual_seed(7)
    shapes = [(4, 30, 50), (4, 30, 50), (3, 7, 33), (4, 30, 50), (1, 1, 1), (5, 12, 64)]      # (rows, steps, width), ragged on purpose
    want = [torch.stack([real_empty(r, w).exponential_(1, generator=ref) for _ in range(n)]) for r, n, w in shapes]
    tail = real_empty(5).exponential_(1, generator=ref)
    for drawers in (3, 0):
        g = torch.Generator().manual_seed(7)
        feed = P.NoiseFeed(g, "cpu", drawers=drawers)
        assert feed.parallel == (drawers > 0)
        tickets = [feed.request(r, n, w) for r, n, w in shapes[:4]]
        feed.drain()                                   # another consumer draws here (class labels): the generator must stand behind request 3
        mid = torch.Generator()
        mid.set_state(g.get_state())
        feed.resync()
        tickets += [feed.request(r, n, w) for r, n, w in shapes[4:]]
        for t in tickets:
            assert t["done"].wait(60) and t["error"] is None
        feed.close()
        for t, w in zip(tickets, want):
            assert torch.equal(t["noise"], w)
        assert torch.equal(real_empty(5).exponential_(1, generator=g), tail)
        chk = torch.Generator().manual_seed(7)
        for r, n, w in shapes[:4]:
            real_empty(n * r * w).exponential_(1, generator=chk)
        assert torch.equal(mid.get_state(), chk.get_state())


def _stub_device_pieces(monkeypatch):
    import contextlib
    import torch

    class _Stub:
        def __init__(self, *a, **k):
            pass

        def record(self):
            pass

    real_empty = torch.empty
    monkeypatch.setattr(torch.cuda, "Stream", _Stub)
    monkeypatch.setattr(torch.cuda, "Event", _Stub)
    monkeypatch.setattr(torch.cuda, "stream", lambda s: contextlib.nullcontext())
    monkeypatch.setattr(torch.cuda, "set_device", lambda d: None)
    monkeypatch.setattr(torch, "empty", lambda *a, **k: real_empty(*a, **{kk: v for kk, v in k.items() if kk != "pin_memory"}))
    return real_empty


def test_noise_feed_drain_waits_for_a_slow_skipper(monkeypatch):
    """`pending == 0` said "every stream is drawn", not "the skipper has counted the last one off": with the
    drawers faster than the skipper drain() published a mid-stream cursor and the class labels that `condition()` draws next came
    from the wrong place.  Here the skipper is slowed down until every drawer has finished first, and the schedule of a
    class-conditional run without given labels -- labels(i), noise(i), labels(i+1), ... -- must reproduce the serial order."""
    import time
    import torch
    from ccvs_amd.helpers import pipeline as P
    real_empty = _stub_device_pieces(monkeypatch)
    orig_skip = P.NoiseFeed._skip

    def slow_skip(self, n):
        time.sleep(0.15)                 # the drawers (tiny streams) are done long before
        orig_skip(self, n)

    monkeypatch.setattr(P.NoiseFeed, "_skip", slow_skip)
    shape = (4, 9, 31)                   # rows, steps, width
    ref = torch.Generator().manual_seed(11)
    want = []
    for _ in range(3):                   # the serial loop: labels of batch i (generator.py:124), then its picks' noise
        lbl = torch.randint(0, 101, (shape[0],), generator=ref)
        noise = torch.stack([real_empty(shape[0], shape[2]).exponential_(1, generator=ref) for _ in range(shape[1])])
        want.append((lbl, noise))
    g = torch.Generator().manual_seed(11)
    feed = P.NoiseFeed(g, "cpu", drawers=3)
    assert feed.parallel
    got = []
    for i in range(3):
        feed.drain()                     # `_submit_group`: behind the previous batch's noise ...
        lbl = torch.randint(0, 101, (shape[0],), generator=g)
        feed.resync()                    # ... and this batch's noise behind the labels
        got.append((lbl, feed.request(*shape)))
    for (lbl, t), (wl, wn) in zip(got, want):
        assert t["done"].wait(60) and t["error"] is None
        assert torch.equal(lbl, wl)
        assert torch.equal(t["noise"], wn)
    feed.close()
    assert torch.equal(torch.randint(0, 101, (7,), generator=g), torch.randint(0, 101, (7,), generator=ref))


def test_noise_feed_failure_is_sticky(monkeypatch):
    """a skip that fails leaves the cursor in an unknown place -- every later request must carry the error (not
    noise from a misplaced stream), and drain() / close() must raise instead of publishing the cursor."""
    import pytest as _pytest
    import torch
    from ccvs_amd.helpers import pipeline as P
    _stub_device_pieces(monkeypatch)
    calls = {"n": 0}
    orig_skip = P.NoiseFeed._skip

    def failing_skip(self, n):
        calls["n"] += 1
        if calls["n"] == 2:
            raise MemoryError("host memory exhausted (test)")
        orig_skip(self, n)

    monkeypatch.setattr(P.NoiseFeed, "_skip", failing_skip)
    g = torch.Generator().manual_seed(5)
    before = g.get_state().clone()
    feed = P.NoiseFeed(g, "cpu", drawers=2)
    t1, t2, t3 = feed.request(2, 3, 8), feed.request(2, 3, 8), feed.request(2, 3, 8)
    for t in (t1, t2, t3):
        assert t["done"].wait(60)
    assert t1["error"] is None
    assert isinstance(t3["error"], MemoryError)          # served behind the failure: refused
    with _pytest.raises(RuntimeError):
        feed.drain()
    with _pytest.raises(RuntimeError):
        feed.request(2, 3, 8)
    with _pytest.raises(RuntimeError):
        feed.close()
    assert torch.equal(g.get_state(), before)            # the process generator was not moved to a wrong position


def test_group_sizes_the_warm_up_prepares():
    """`PipelinedRun._group_sizes_ahead`: with the number of batches known the warm-up captures the decode step of exactly the group sizes
    `_submit_group` will form (ramp, then `lanes`, then the remainder) -- every size up to `lanes` otherwise (each is a KV cache)."""
    from ccvs_amd.helpers.pipeline import PipelinedRun

    class _Run:
        _group_sizes_ahead = PipelinedRun._group_sizes_ahead

    r = _Run()
    r.lanes, r.ramp, r.warm_counts = 4, (), (None,)
    assert list(r._group_sizes_ahead()) == [1, 2, 3, 4]
    r.warm_counts = (20,)
    assert list(r._group_sizes_ahead()) == [4]
    r.warm_counts = (5,)
    assert list(r._group_sizes_ahead()) == [1, 4]
    r.lanes, r.ramp, r.warm_counts = 8, (4, 4, 4, 8), (20,)
    assert list(r._group_sizes_ahead()) == [4, 8]
    r.lanes, r.ramp, r.warm_counts = 10, (), (20,)
    assert list(r._group_sizes_ahead()) == [10]
    r.lanes, r.ramp, r.warm_counts = 4, (1, 2), (6,)
    assert list(r._group_sizes_ahead()) == [1, 2, 3]
    # a warm-up run told the length of the run that follows (bench.py: --warmup 1 --steps 3; --warmup 5 --steps 20) captures that run's sizes too
    r.lanes, r.ramp, r.warm_counts = 4, (), (1, 3)
    assert list(r._group_sizes_ahead()) == [1, 3]
    r.warm_counts = (5, 20)
    assert list(r._group_sizes_ahead()) == [1, 4]
    r.warm_counts = (2, 6)
    assert list(r._group_sizes_ahead()) == [2, 4]
    r.warm_counts = (2, None)
    assert list(r._group_sizes_ahead()) == [1, 2, 3, 4]


def test_power_trace_reads_rocm_smi_records():
    """tools/power_trace.py: the record rocm-smi printed on the GPU box (profiles/r06_power_trace.txt was reduced from such samples) -> power,
    cap, clocks, junction temperature; and the percentile helper."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("power_trace", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "power_trace.py"))
    pt = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(pt)
    card = {"Temperature (Sensor junction) (C)": "44.0", "Temperature (Sensor memory) (C)": "32.0", "fclk clock speed:": "(1250Mhz)", "fclk clock level:": "0",
            "mclk clock speed:": "(2000Mhz)", "mclk clock level:": "0", "sclk clock speed:": "(2209Mhz)", "sclk clock level:": "S", "socclk clock speed:": "(38Mhz)",
            "socclk clock level:": "S", "Max Graphics Package Power (W)": "1400.0", "Current Socket Graphics Package Power (W)": "1319.0"}
    rec = pt.parse(card)
    assert rec == {"tj_c": 44.0, "mclk_mhz": 2000.0, "sclk_mhz": 2209.0, "cap_w": 1400.0, "power_w": 1319.0}
    assert pt.pct([5, 1, 3, 2, 4], 0.5) == 3 and pt.pct([], 0.5) is None
