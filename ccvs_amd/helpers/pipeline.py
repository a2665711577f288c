"""Several generation batches in flight on one GPU: the schedule behind `Generator.run_pipelined` and `Generator.run`.

The work of one batch is exactly `Generator.generate_vid`'s (reference helpers/generator.py:57-230: encode -> crop -> token
synthesis -> flow-guided decode); this module only decides WHEN each piece is enqueued and on which HIP stream:

  * the token loops of `lanes` consecutive batches run as ONE loop over their stacked rows (a *token group*: one KV cache of
    lanes x B rows, one captured decode step with per-group sampler state -- `ccvs_gpt_decode.groups`), so the GPT weights are
    streamed once per token for all of them;
  * `chains` such loops run beside each other, each on its own stream, each fed by its own worker thread (a hipGraph launch
    blocks its caller once the stream's queue is a few dozen steps deep, and the decoder's launches must not wait behind that);
  * the encode stream encodes the batches of the groups to come (chains + 1 groups in front of the decoder);
  * `dec_streams` decode streams decode -- FRAME BY FRAME, as the tokens arrive: the decoder needs the tokens of frame t only
    for frame t (`QVidModel.decode_frames`), so a running token loop hands every finished frame to a `FrameFeed` and the decode
    of a batch is a generator advanced one frame at a time, the oldest batch whose next frame is there first; the decode of
    batch i runs on decode stream i % D;
  * host-drawn sampling noise (`--x_sample_noise host`: the stream torch.multinomial consumes under the reference's process
    seed, transformer_model.py:395-409) is pre-drawn batch after batch, in the reference's order, by a `NoiseFeed` thread and
    uploaded on a copy stream: a token group then reads every batch's own stream from inside the captured step
    (`ccvs_gpt_decode.noise_stream`) -- reference-seed sampling costs the schedule nothing.

`PipelinedRun` holds that state as fields (streams, queues, the jobs and tasks in flight) with one method per stage; results
are bit-identical to the serial schedule (tests/test_pipeline_gpu.py).
"""
import os
import queue
import sys
import threading
import time
from collections import deque

import torch

from ccvs_amd import ops
from ccvs_amd.models.skip_vid_generator.models.skip_autoencoder import prepare_packed_modules


class FrameFeed:
    """The tokens of one token group on their way to the decoder, frame by frame.  The token stream copies every finished
    frame out of the loop's own buffer (which the chain's next group overwrites) and records an event behind the copy; the
    decoder's thread sees `flags[f]` once `events[f]` exists and waits for it on its stream."""

    def __init__(self, rows, frames, frame_tokens, device):
        self.size = frame_tokens
        self.codes = torch.empty(rows, frames * frame_tokens, dtype=torch.int64, device=device)
        self.events = [None] * frames
        self.flags = [threading.Event() for _ in range(frames)]
        self.sent = 0

    def on_tokens(self, n, codes):
        """Columns [0, n) of `codes` are final in the order of the current (token) stream: pass on the frames they complete."""
        f = min(n // self.size, len(self.events))
        if f > self.sent:
            lo, hi = self.sent * self.size, f * self.size
            self.codes[:, lo:hi].copy_(codes[:, lo:hi])
            ev = torch.cuda.Event()
            ev.record()
            for i in range(self.sent, f):
                self.events[i] = ev
                self.flags[i].set()
            self.sent = f

    def release(self):
        """End of the token stage (also a failed one): nobody waits for a frame that will not come."""
        for flag in self.flags:
            flag.set()


class NoiseFeed:
    """Host-drawn sampling noise, one request per batch: request i is the `steps` blocks [rows, V] of Exp(1) values that
    `torch.multinomial` would draw for batch i's picks from the process generator (transformer_model.py:395-409 --
    `empty_like(probs).exponential_(1, gen)` per pick), so a run with many batches in flight consumes the generator exactly like the
    reference's one-batch-at-a-time loop.  The generator is a serial stream (mt19937), but WHERE batch i's draws start is known as soon
    as the draws before it have been counted off: one *skipper* thread walks a private copy of the generator from request to request
    by drawing the same number of raw 64-bit values (`random_()` on int64: one `random64()` per element, like `exponential_`, at half
    its cost -- checked against `exponential_` when the feed starts; if the states ever disagreed the feed falls back to drawing
    every stream on the skipper itself) and hands the state at each boundary to a pool of *drawer* threads, which produce the
    streams in parallel with torch's own `exponential_`, block by block, into pinned memory and upload them on a copy stream.
    `ticket["done"]` is set when the upload is enqueued, `ticket["event"]` is recorded behind it.  The process generator itself is moved to
    the end of everything requested in `drain()` / `close()`; nobody else may draw from it while requests are pending."""

    SKIP_CHUNK = 1 << 22   # raw values per skip call (32 MB of int64)

    def __init__(self, generator, device, drawers=None):
        self.real = generator if generator is not None else torch.default_generator
        self.device = device
        # (device "cpu": the host half only -- draw into ordinary memory, no upload; tools/host_stress.py measures N ranks' host work
        # side by side that way, without a GPU)
        self.on_gpu = torch.device(device).type == "cuda"
        self.copy_stream = torch.cuda.Stream(device=device) if self.on_gpu else None
        self.cursor = torch.Generator()          # the skipper's copy: always at the start of the next request
        self.cursor.set_state(self.real.get_state())
        n = int(os.environ.get("CCVS_NOISE_DRAWERS", "4")) if drawers is None else int(drawers)
        self.parallel = n > 0 and self._skip_matches_draw()
        self.requests, self.work = queue.Queue(), queue.Queue()
        self.pending = 0          # requests not yet drawn (uploaded): decremented by whoever drew them
        self.unskipped = 0        # requests whose values the skipper has not yet counted off on `cursor`
        self.failed = None        # sticky: the exception that left `cursor` in an unknown place
        self.timeout = float(os.environ.get("CCVS_PIPELINE_TIMEOUT", "600"))
        self.idle = threading.Condition()
        self.threads = [threading.Thread(target=self._skipper, name="ccvs-noise-skipper", daemon=True)]
        if self.parallel:
            self.threads += [threading.Thread(target=self._drawer, name=f"ccvs-noise-drawer-{k}", daemon=True) for k in range(n)]
        self.n_drawers = len(self.threads) - 1
        for th in self.threads:
            th.start()

    def _skip_matches_draw(self):
        """`random_()` on n int64 values leaves a generator where `exponential_` on n floats leaves it (ragged n included)."""
        for n in (1, 17, 4099):
            a, b = torch.Generator(), torch.Generator()
            a.set_state(self.cursor.get_state())
            b.set_state(self.cursor.get_state())
            torch.empty(n, dtype=torch.float32).exponential_(1, generator=a)
            torch.empty(n, dtype=torch.int64).random_(generator=b)
            if not torch.equal(a.get_state(), b.get_state()):
                return False
        return True

    def request(self, rows, steps, width):
        ticket = {"rows": rows, "steps": steps, "width": width, "done": threading.Event(), "event": None, "noise": None, "error": None}
        with self.idle:
            if self.failed is not None:    # the cursor stands in the wrong place since a skip failed: serve nothing from it
                raise RuntimeError("NoiseFeed: an earlier request failed while its values were counted off; the noise stream is lost") from self.failed
            self.pending += 1
            self.unskipped += 1
        self.requests.put(ticket)
        return ticket

    def _settled(self):
        """Every request so far has been DRAWN (`pending`: the drawers) and COUNTED OFF on the cursor (`unskipped`: the skipper).  The two
        finish in either order -- a drawer runs `exponential_`, the skipper the cheaper `random_`, but with pinned cores, launch threads
        and torch's pool competing nothing says the skipper wins -- and the cursor is only at the start of the next request once the
        skipper is through (`pending == 0` alone let drain() publish a mid-stream cursor)."""
        return self.pending == 0 and self.unskipped == 0

    def drain(self):
        """Wait until every request so far has been drawn and counted off, and move the process generator behind them (another
        consumer of the generator is about to draw)."""
        with self.idle:
            if not self.idle.wait_for(lambda: self._settled() or self.failed is not None, timeout=self.timeout):
                raise RuntimeError(f"NoiseFeed.drain: requests still pending after {self.timeout:.0f} s")
            if self.failed is not None:
                raise RuntimeError("NoiseFeed: a request failed while its values were counted off; the generator's position is unknown") from self.failed
            self.real.set_state(self.cursor.get_state())

    def resync(self):
        """After `drain()` and the other consumer's draws: the next request starts where the process generator now stands."""
        with self.idle:
            assert self._settled()
            self.cursor.set_state(self.real.get_state())

    def close(self):
        self.requests.put(None)
        self.threads[0].join(60.0)
        for _ in range(self.n_drawers):
            self.work.put(None)
        for th in self.threads[1:]:
            th.join(60.0)
        if self.failed is not None:
            raise RuntimeError("NoiseFeed: a request failed while its values were counted off; the generator was left where the run found it") from self.failed
        if self.threads[0].is_alive():     # still counting: its cursor is mid-stream
            raise RuntimeError("NoiseFeed.close: the skipper thread did not finish within 60 s; the generator was left where the run found it")
        self.real.set_state(self.cursor.get_state())

    def _finish(self, t):
        t["done"].set()
        with self.idle:
            self.pending -= 1
            self.idle.notify_all()

    def _skipped(self):
        with self.idle:
            self.unskipped -= 1
            self.idle.notify_all()

    def _draw(self, t, gen):
        """One stream from `gen` (positioned at its first value): block after block, exactly the calls torch.multinomial makes."""
        try:
            buf = torch.empty(t["steps"], t["rows"], t["width"], dtype=torch.float32, pin_memory=self.on_gpu)
            for i in range(t["steps"]):
                buf[i].exponential_(1, generator=gen)
            if not self.on_gpu:
                t["noise"] = buf
            else:
                with torch.cuda.stream(self.copy_stream):
                    t["noise"] = buf.to(self.device, non_blocking=True)
                    t["event"] = torch.cuda.Event()
                    t["event"].record()
        except BaseException as exc:
            t["error"] = exc
        finally:
            self._finish(t)

    def _skipper(self):
        if self.on_gpu:
            torch.cuda.set_device(self.device)
        while True:
            t = self.requests.get()
            if t is None:
                return
            if self.failed is not None:       # sticky: every later ticket carries the error instead of noise from a misplaced cursor
                t["error"] = self.failed
                self._finish(t)
                self._skipped()
                continue
            if not self.parallel:
                self._draw(t, self.cursor)
                if t["error"] is not None:
                    self.failed = t["error"]
                self._skipped()
                continue
            state = self.cursor.get_state()
            left = t["steps"] * t["rows"] * t["width"]
            try:
                self.work.put((t, state))
                while left > 0:           # count the stream's values off on the cursor: where the next request starts
                    n = min(left, self.SKIP_CHUNK)
                    self._skip(n)
                    left -= n
            except BaseException as exc:   # (out of host memory ...): the tickets behind this one would start in the wrong place
                self.failed = exc
            finally:
                self._skipped()

    def _skip(self, n):
        torch.empty(n, dtype=torch.int64).random_(generator=self.cursor)

    def _drawer(self):
        if self.on_gpu:
            torch.cuda.set_device(self.device)
        gen = torch.Generator()
        while True:
            item = self.work.get()
            if item is None:
                return
            t, state = item
            gen.set_state(state)
            self._draw(t, gen)


class PipelinedRun:
    """One call of `Generator.run_pipelined`: the configuration, the streams, the token groups (`jobs`) and decodes (`tasks`)
    in flight, and one method per stage of the schedule."""

    def __init__(self, gen, batches, first_iter=0, cu_limit=None, finish=None, lanes=None, chains=None, ramp=None, rec_pass=False, consume=None,
                 n_batches=None):
        opt = gen.opt
        if opt.step_by_step or opt.rec_only:
            raise NotImplementedError("run_pipelined covers the plain synthesis schedule (use generate_vid for step_by_step / rec_only)")
        self.gen, self.opt, self.finish, self.consume, self.first_iter = gen, opt, finish, consume, first_iter
        self.rec_pass = bool(rec_pass) and not opt.gen_from_img      # the teacher-forced decode of the clip's own codes, behind its synthesis
        env = os.environ.get
        self.dev = torch.device("cuda", torch.cuda.current_device())
        self.lanes = int(env("CCVS_PIPELINE_LANES", "4")) if lanes is None else int(lanes)
        self.chains = max(1, int(env("CCVS_PIPELINE_CHAINS", "2")) if chains is None else int(chains))
        self.ramp = tuple(int(v) for v in env("CCVS_PIPELINE_RAMP", "").split(",") if v) if ramp is None else tuple(ramp)
        self.cu_limit = int(env("CCVS_PIPELINE_CU_LIMIT", "0")) if cu_limit is None else int(cu_limit)   # 0 = no budget
        self.timeout = float(env("CCVS_PIPELINE_TIMEOUT", "600"))
        self.depth = max(1, int(env("CCVS_PIPELINE_DEPTH", "2")))
        self.debug = env("CCVS_PIPELINE_DEBUG", "0") == "1"
        self.stream_frames = env("CCVS_PIPELINE_STREAM", "1") != "0"   # 0: a batch is decoded when its whole token stage is done
        # Batches whose decode has begun and not ended hold their context rings (BAIR: 8.5 GB per batch of 16 clips -- 15 slots x 6
        # levels); with the decoder following the token loops frame by frame EVERY batch of the running token groups would begin at
        # once.  At most this many are in decode at a time, the oldest first; the others' tokens wait in their FrameFeed (0: no cap)
        self.max_decoding = int(env("CCVS_PIPELINE_MAX_DECODING", "0"))
        self.frame_tokens = int(gen.qvid_opt.z_shape[0]) * int(gen.qvid_opt.z_shape[1])
        # host-drawn sampling noise: pre-drawn per batch when the batch's draws are one plain stream of [B, V] blocks; otherwise the
        # token stages draw for themselves and must run one after the other, in batch order (one chain, one batch per group)
        self.host_noise = bool(getattr(opt, "sample", False)) and getattr(gen.transformer_model, "sample_noise", "host") != "device"
        self.noise_feed = None
        # batches of the run when the caller knows (a list; `Generator.run`'s n_iter): the warm-up then captures the decode step of the
        # group sizes that WILL occur instead of every size up to `lanes` (one KV cache each: lanes (lanes + 1) / 2 x 3.2 GB per chain at BAIR)
        # (a tuple: this run's count, then counts of runs to come whose group sizes this run's warm-up should capture as well -- a benchmark's
        #  untimed warm-up run in front of a timed run of another length: the capture of a size met first there would sit inside the clock)
        also = tuple(n_batches[1:]) if isinstance(n_batches, (tuple, list)) else ()
        n_batches = n_batches[0] if isinstance(n_batches, (tuple, list)) else n_batches
        self.n_batches = n_batches if n_batches is not None else (len(batches) if hasattr(batches, "__len__") else None)
        self.warm_counts = (self.n_batches,) + also
        self.it = iter(batches)
        self.held = []            # a batch read ahead: the first one (its size decides below), or one that did not fit its group (ragged size)
        first = next(self.it, None)
        if first is not None:
            self.held.append(first)
        self.noise_streams = self.host_noise and gen._host_noise_streams_ok(first["vid"].shape[0] if first is not None else None)
        if self.host_noise and not self.noise_streams:
            self.chains = 1
        # stream D and its twins: the encodes run on the first, the decode of batch i on stream i % n (two decodes side by side: the
        # small launches of one's coarse pyramid levels run under the other's large ones).  Default: two for frames of 128^2 and
        # more -- there the decoder is the longer stage; one for the 64^2 configurations, whose token loops are
        n_dec = env("CCVS_PIPELINE_DEC_STREAMS", "")
        n_dec = max(1, int(n_dec)) if n_dec else (2 if int(getattr(opt, "max_dim", 256)) >= 128 else 1)
        prio = [int(v) for v in env("CCVS_PIPELINE_PRIORITIES", "0,-1").split(",")]   # (token streams, decode streams)
        have = list(getattr(gen, "_dec_streams", ()))
        if len(have) < n_dec:
            gen._dec_streams = have + [torch.cuda.Stream(device=self.dev, priority=prio[1]) for _ in range(n_dec - len(have))]
            gen._dec_stream = gen._dec_streams[0]
        self.dec_streams = gen._dec_streams[:n_dec]
        self.s_enc = self.dec_streams[0]
        if env("CCVS_PIPELINE_ENC_STREAM", "0") == "1":    # the encodes on a stream of their own (slower: 199 against 214 frames/s)
            if getattr(gen, "_enc_stream", None) is None:
                gen._enc_stream = torch.cuda.Stream(device=self.dev, priority=prio[1])
            self.s_enc = gen._enc_stream
        # CCVS_PIPELINE_ENC_SPREAD=1 (experiments): the encode of batch i on the decode stream that does NOT decode batch i ((i + 1) % D),
        # so that every decode stream carries the same share of encodes.  Measured +-0 (profiles/r05_enc_spread_ab.txt: 203.8 / 207.0
        # against 206.8 / 206.7 frames/s): an encode beside the other stream's decode takes 146-204 ms instead of 83 -- the chip, not the
        # longer stream, is what is full.  Default: all encodes on the first decode stream.
        self.enc_spread = env("CCVS_PIPELINE_ENC_SPREAD", "0") == "1" and self.s_enc is self.dec_streams[0] and len(self.dec_streams) > 1
        self.chain_list = [gen._token_chain(k) for k in range(self.chains)]
        self.entry = torch.cuda.current_stream()
        self.index = first_iter
        self.n_groups = 0
        self.queues = [queue.Queue() for _ in range(self.chains)]
        self.abort = threading.Event()   # set on the error path: queued token stages are dropped instead of run
        self.jobs = deque()       # submitted token groups with undecoded batches, oldest first
        self.tasks = []           # the decodes of their batches, oldest first
        self.in_flight = [deque() for _ in self.dec_streams]   # per decode stream: events behind the pieces of decoder work enqueued last
        self.finished = {}        # batch index -> (task, clip) decoded ahead of an earlier batch: handed out in index order
        self.exhausted, self.next_out, self.budget_now = False, first_iter, 0
        self.results, self.timings = [], []
        gen.last_cu_limit, gen.last_lanes, gen.last_chains, gen.last_dec_streams = self.cu_limit, self.lanes, self.chains, len(self.dec_streams)

    # ------------------------------------------------------------------ the run
    def run(self):
        side = set(self.dec_streams + [self.s_enc] + [st for _, st in self.chain_list])
        for st in side:
            st.wait_stream(self.entry)
        # Several batches in flight: the chip is power-managed then, and the persistent-tile form of the convolutions -- faster alone -- costs
        # every kernel beside it clock (profiles/r06_conv_pt_bench_ab.txt): the per-tile kernels for the duration of such a run.
        # One batch (generate_vid's stream schedule) keeps the library's default.  Process-wide: restored in `finally`.
        many = (self.n_batches or 2) > 1
        pt_prev = ops.conv_persistent_tiles(0) if many else None
        try:
            return self._run(side)
        finally:
            if pt_prev is not None:
                ops.conv_persistent_tiles(pt_prev)

    def _run(self, side):
        gen = self.gen
        self._warm_up()
        if self.noise_streams:
            self.noise_feed = NoiseFeed(gen.transformer_model.generator, self.dev)
        threads = [threading.Thread(target=self._token_worker, args=(k,), name=f"ccvs-token-chain-{k}", daemon=True) for k in range(self.chains)]
        for th in threads:
            th.start()
        try:
            self._top_up()
            while self.tasks:
                self._advance(self._pick())
                self._hand_out()
                self._top_up()
        except BaseException:
            self.abort.set()   # the workers drop what is still queued; the error surfaces now, not after every queued token loop has run
            raise
        finally:
            for q_ in self.queues:
                q_.put(None)
            for th in threads:
                th.join(30.0 if self.abort.is_set() else self.timeout)
            if self.noise_feed is not None:
                self.noise_feed.close()
            self._budget(0)
            # (also on the error path: the caller's stream must not run ahead of work still queued on ours)
            for st in side:
                self.entry.wait_stream(st)
        gen._pipeline_events = self.timings
        return self.results

    def _budget(self, n):
        """CU budget of everything submitted to the decode streams from now on."""
        for st in set(self.dec_streams + [self.s_enc]):
            ops.stream_cu_limit(st, n)

    # ------------------------------------------------------------------ captures, up front
    def _capture_key(self, chain, rows, groups):
        """What a captured decode step depends on (the cache length follows from the options): such a key is warmed once."""
        opt = self.opt
        tr = self.chain_list[chain][0]
        ver = sum(p_._version for p_ in tr.net_t.parameters())
        return (chain, rows, groups, opt.vid_len, opt.z_len, opt.cond_len, bool(opt.p2p), bool(opt.sample), opt.top_k, float(opt.temperature),
                tr.sample_noise, ver)

    def _is_cold(self, c, nb, g):
        # the captured steps live in the engine's cache of that row count: ask the engine, not only the side set (a serial
        # generate_vid with the same rows but another group count or a longer sequence, the eviction at 8 entries or
        # drop_engine_state() rebuild the cache and drop its graphs -- the capture would then happen in a worker thread)
        cache = self.chain_list[c][0].net_t._caches.get(nb * g)
        live = cache is not None and cache.get("G") == g and len(cache.get("graphs", {})) > 0
        key = self._capture_key(c, nb * g, g)
        # ... and the live graphs must be THIS key's: a run with another sampler in between (bench.py's legs: host noise, device
        # noise, host noise again) leaves graphs of its own behind a key that is still in the side set
        if not live or cache.get("warm_key") != key:
            self.gen._warm_keys.discard(key)
        return key not in self.gen._warm_keys

    def _warm(self, c, nb, g, ws):
        """Capture the decode step of chain c for g stacked batches of nb clips (inputs: the working set of one such batch),
        from this thread; nothing is replayed, the sampler words are put back, nothing is drawn from the host generator."""
        gen = self.gen
        tr, s_tok = self.chain_list[c]
        for st in set(self.dec_streams + [self.s_enc]):     # `ws` was encoded on one of them
            s_tok.wait_stream(st)
        with torch.cuda.stream(s_tok):
            gen._seed_sampler_group(nb, list(range(g)), tr.net_t)
            tr.net_t.warm_only = True
            try:
                tr(self._stack_inputs([ws["cropped"]] * g), mode='inference', total_len=ws["total_len"])
            finally:
                tr.net_t.warm_only = False
                tr.net_t.noise_key, tr.net_t.row_offset = None, 0
        s_tok.synchronize()
        key = self._capture_key(c, nb * g, g)
        gen._warm_keys.add(key)
        cache = tr.net_t._caches.get(nb * g)
        if cache is not None:
            cache["warm_key"] = key
        self.s_enc.wait_stream(s_tok)

    def _warm_up(self):
        """Everything a stage builds lazily on first use is built here, from this thread, before any worker exists and before
        two decode streams can touch it:
          * the kernel-ready weights of the encoder / decoder (packed convolution weights, fused heads, split Subpixel weights:
            plain Python caches filled by pack kernels on whichever stream first needs them -- with the decode of batch i on
            stream i % D, stream 1 could otherwise read weights whose pack kernels are still queued on stream 0): packed on the
            encode stream, every decode and token stream waits for that;
          * the decode step of every (chain, group size) this run can use that is not captured yet: first-time launches and graph
            captures never race with another thread's launches, and no capture falls into the steady state.  The first batch is
            encoded once more for it; results are discarded.  (A later batch of another size: `_submit_group` waits until no
            worker is launching and captures then.)"""
        gen = self.gen
        with torch.cuda.stream(self.s_enc):
            gen.vid_model.prepare_packed()
            for model in (gen.state_model, gen.stft_model):
                if model is not None:
                    prepare_packed_modules(model)
            packed = torch.cuda.Event()
            packed.record()
        for st in set(self.dec_streams + [st_ for _, st_ in self.chain_list]):
            st.wait_event(packed)
        if not self.held:
            return
        first = self.held[-1]
        nb = first["vid"].shape[0]
        free, total = torch.cuda.mem_get_info(self.dev)
        foreign = max(0, total - free - torch.cuda.memory_reserved(self.dev))     # held by other processes on this device
        noise_bytes = 0.0
        if self.noise_streams:      # every batch in flight holds its pre-drawn stream [steps, B, V] on the device
            size = self.frame_tokens
            steps = self.opt.vid_len * size - int(self.opt.cond_len) - (size if self.opt.p2p else 0)
            noise_bytes = 4.0 * max(steps, 0) * nb * gen.transformer_model.net_t.head.weight.shape[0]
        fit = gen._lanes_that_fit(self.lanes, self.chains, nb, first["vid"].shape[-2], first["vid"].shape[-1], total,
                                  frac=float(os.environ.get("CCVS_PIPELINE_MEM_FRAC", "0.7")), taken=foreign, dec_streams=len(self.dec_streams),
                                  noise_bytes_per_batch=noise_bytes)
        if fit < self.lanes:
            print(f"[pipeline] {self.lanes} -> {fit} batches per token group: {self.lanes * (self.chains + 2)} batches of {nb} clips in flight would not fit "
                  "the device memory (CCVS_PIPELINE_MEM_FRAC)", file=sys.stderr, flush=True)
            self.lanes = gen.last_lanes = fit
        sizes = sorted({gen._token_group_size(nb, g) for g in self._group_sizes_ahead()})
        cold = [(c, g) for c in range(self.chains) for g in sizes if self._is_cold(c, nb, g)]
        if not cold:
            return
        with torch.cuda.stream(self.s_enc):
            ws = gen.condition({k: (v.clone() if torch.is_tensor(v) else v) for k, v in first.items()}, draw=False)
        for c, g in cold:
            self._warm(c, nb, g, ws)

    def _group_sizes_ahead(self):
        """Sizes of the token groups this run will form: every size up to `lanes` when the number of batches is unknown (a ragged tail can
        be any of them), else exactly the ones `_submit_group` will ask for -- the ramp, then `lanes`, then the remainder."""
        if any(c is None for c in self.warm_counts):
            return range(1, self.lanes + 1)
        sizes = set()
        for count in self.warm_counts:
            left, i = count, 0
            while left > 0:
                want = min(self.ramp[i] if i < len(self.ramp) else self.lanes, self.lanes, left)
                sizes.add(max(1, want))
                left -= max(1, want)
                i += 1
        return sorted(sizes)

    # ------------------------------------------------------------------ stage 1: encode + crop, queue the token stage
    @staticmethod
    def _stack_inputs(cropped):
        tok_in = {}
        for key in ("code", "cond_code", "state_code", "vid_lbl", "delta_length_cond"):
            if key in cropped[0]:
                tok_in[key] = cropped[0][key] if len(cropped) == 1 else torch.cat([c[key] for c in cropped], dim=0)
        return tok_in

    def _submit_group(self):
        """Encode + crop the next group's batches on the encode stream and queue their token stage; None when the input is exhausted."""
        gen, opt = self.gen, self.opt
        members, group = [], None
        chain = self.n_groups % self.chains
        s_tok = self.chain_list[chain][1]
        while group is None or len(members) < group:
            data = self.held.pop() if self.held else next(self.it, None)
            if data is None:
                break
            if group is None:
                want = self.ramp[self.n_groups] if self.n_groups < len(self.ramp) else self.lanes
                group = gen._token_group_size(data["vid"].shape[0], min(want, self.lanes))
            elif data["vid"].shape[0] != members[0]["batch"]:
                self.held.append(data)         # a ragged batch starts the next group
                break
            other_draw = self.noise_feed is not None and opt.cat and "vid_lbl" not in data
            if other_draw:
                self.noise_feed.drain()        # `condition` draws the labels from the same generator: behind the previous batch's noise
            ev = {k: torch.cuda.Event(enable_timing=True) for k in ("e0", "e1", "d0", "d1")}
            s_enc = self.dec_streams[(self.index - self.first_iter + 1) % len(self.dec_streams)] if self.enc_spread else self.s_enc
            with torch.cuda.stream(s_enc):
                ev["e0"].record()
                ws = gen.condition(data)
                ev["e1"].record()
            if other_draw:
                self.noise_feed.resync()       # ... and this batch's noise behind the labels
            for t in (ws["cropped"]["code"], ws["cropped"].get("cond_code"), ws["cropped"].get("state_code")):
                if torch.is_tensor(t) and t.is_cuda:
                    t.record_stream(s_tok)
            m = {"i": self.index, "ws": ws, "ev": ev, "batch": data["vid"].shape[0], "noise": None, "s_enc": s_enc}
            if self.noise_feed is not None:    # this batch's draws, in batch order: one [B, V] block per new token
                n_cond = ws["cropped"]["cond_code"].shape[1] if "cond_code" in ws["cropped"] else 0
                add_len = int(ws["total_len"]) - ws["cropped"]["code"].shape[1] - n_cond
                m["noise"] = self.noise_feed.request(m["batch"], add_len, gen.transformer_model.net_t.head.weight.shape[0])
            members.append(m)
            self.index += 1
        if not members:
            return None
        if any(m["ws"]["total_len"] != members[0]["ws"]["total_len"] for m in members):
            raise RuntimeError("run_pipelined: the batches of a token group must share total_len")
        self.n_groups += 1
        nb = members[0]["batch"]
        if self._is_cold(chain, nb, len(members)):
            # a batch size _warm_up() has not seen (ragged input): the capture must not race with a worker's launches --
            # wait until every queued token stage is enqueued (the workers then sleep on their queues), capture here
            for queued in self.jobs:
                self._wait_job(queued, "the token stage (before a capture for a new batch size)")
            self._warm(chain, nb, len(members), members[0]["ws"])
        with torch.cuda.stream(self.s_enc):
            feed = FrameFeed(len(members) * nb, opt.vid_len, self.frame_tokens, self.dev)
            feed.codes.record_stream(s_tok)
        job = {"members": members, "batch": nb, "total_len": members[0]["ws"]["total_len"], "chain": chain,
               "enc_done": [m["ev"]["e1"] for m in members], "t0": torch.cuda.Event(enable_timing=True), "t1": torch.cuda.Event(enable_timing=True),
               "codes": None, "error": None, "done": threading.Event(), "feed": feed, "left": len(members)}
        self.queues[chain].put(job)
        return job

    def _top_up(self):
        """Keep every chain busy and one more group encoded behind them (at most chains + 2 groups in flight)."""
        while (not self.exhausted and len(self.jobs) < self.chains + 2
               and sum(1 for j in self.jobs if not j["done"].is_set()) < self.chains + 1):
            job = self._submit_group()
            if job is None:
                self.exhausted = True
                break
            self.jobs.append(job)
            self.tasks.extend({"job": job, "k": k, "m": m, "gen": None, "need": None, "sid": (m["i"] - self.first_iter) % len(self.dec_streams)}
                              for k, m in enumerate(job["members"]))

    # ------------------------------------------------------------------ stage 2: the token loop of a group (worker thread of its chain)
    def _token_worker(self, chain):
        gen, opt = self.gen, self.opt
        tr, s_tok = self.chain_list[chain]
        torch.cuda.set_device(self.dev)
        while True:
            job = self.queues[chain].get()
            if job is None:
                return
            feed = job["feed"]
            if self.abort.is_set():
                job["error"] = RuntimeError("run_pipelined: aborted")
                job["done"].set()
                feed.release()
                continue
            try:
                with torch.cuda.stream(s_tok), torch.no_grad():
                    for enc_done in job["enc_done"]:     # every member's encode (they run on different streams)
                        s_tok.wait_event(enc_done)
                    members = job["members"]
                    tok_in = self._stack_inputs([m["ws"]["cropped"] for m in members])
                    gen._seed_sampler_group(job["batch"], [m["i"] for m in members], tr.net_t)
                    if members[0]["noise"] is not None:          # every batch's own pre-drawn noise stream, in group order
                        streams = []
                        for m in members:
                            t = m["noise"]
                            if not t["done"].wait(self.timeout):
                                raise RuntimeError(f"run_pipelined: the sampling noise of batch {m['i']} was not drawn within {self.timeout:.0f} s")
                            if t["error"] is not None:
                                raise t["error"]
                            s_tok.wait_event(t["event"])
                            streams.append(t["noise"])
                            m["noise"] = None
                        tr.net_t.noise_streams = streams
                    # one window of tokens, frame tokens only: the loop reports every finished frame (mingpt `progress`)
                    by_frame = self.stream_frames and int(job["total_len"]) <= opt.z_len and not (opt.state or opt.stft)
                    tr.net_t.progress = feed.on_tokens if by_frame else None
                    job["t0"].record()
                    out = tr(tok_in, mode='inference', total_len=job["total_len"])
                    feed.on_tokens(out["code"].shape[1], out["code"])   # whatever the loop has not handed over itself
                    job["t1"].record()
                    job["codes"] = out
            except BaseException as exc:   # re-raised by the main thread when it collects the job
                job["error"] = exc
            finally:
                tr.net_t.progress = None
                tr.net_t.noise_streams = None
                tr.net_t.noise_key, tr.net_t.row_offset = None, 0
                job["done"].set()
                feed.release()

    def _wait_job(self, job, what):
        if not job["done"].wait(self.timeout):
            import faulthandler
            faulthandler.dump_traceback(file=sys.stderr, all_threads=True)
            raise RuntimeError(f"run_pipelined: {what} of batches {[m['i'] for m in job['members']]} (chain {job['chain']}) was not enqueued within "
                               f"{self.timeout:.0f} s (stacks of all threads are on stderr)")
        if job["error"] is not None:
            raise job["error"]

    # ------------------------------------------------------------------ stage 3: the decode of a batch, piece by piece
    def _start_decode(self, task):
        """The decode of one batch as a generator over its frames (`QVidModel.decode_frames`): it reads the tokens of a frame
        from the group's feed and the whole sequence, as the token stage returns it, when it is done."""
        job, k = task["job"], task["k"]
        nb, feed, ft = job["batch"], job["feed"], self.frame_tokens
        st = self.dec_streams[task["sid"]]

        def code_of(lo, hi):
            return feed.codes[k * nb:(k + 1) * nb, lo * ft:hi * ft]

        def final():
            self._wait_job(job, "the token stage")
            codes = job["codes"]
            state_all = codes.get("state_code")
            if state_all is not None and 0 in state_all.size():
                state_all = None
            for t in (codes["code"], state_all):
                if torch.is_tensor(t):
                    t.record_stream(st)
            st.wait_event(job["t1"])
            return codes["code"][k * nb:(k + 1) * nb], (state_all[k * nb:(k + 1) * nb] if state_all is not None else None)

        m = task["m"]
        m["segs"] = []
        if st is not m["s_enc"]:     # encoded on one stream, decoded on this one
            st.wait_event(m["ev"]["e1"])
            for holder in (m["ws"]["cropped"], m["ws"]["encoded"], m["ws"]["data"]):
                for v in holder.values():
                    for t in (v if isinstance(v, (list, tuple)) else (v,)):
                        if torch.is_tensor(t) and t.is_cuda:
                            t.record_stream(st)
        feed.codes.record_stream(st)
        with torch.cuda.stream(st):
            task["gen"] = self.gen._decode_codes_stream(m["ws"], code_of, final)
            task["need"] = next(task["gen"])

    def _is_ready(self, task):
        if task["gen"] is None:
            self._start_decode(task)
        feed, f = task["job"]["feed"], task["need"] - 1
        if not feed.flags[f].is_set():
            return False
        ev = feed.events[f]
        return ev is None or ev.query()    # None: released without tokens (error path) -- `_advance` raises

    def _has_room(self, task):
        """The host stays `depth` pieces ahead of a decode stream, so that "ready" is judged late."""
        q = self.in_flight[task["sid"]]
        while q and q[0].query():
            q.popleft()
        return len(q) <= self.depth

    def _pick(self):
        """The oldest batch whose next frame of tokens is there and whose stream has room; none: wait for one (the worker's
        error, if that is why)."""
        t_end = time.perf_counter() + self.timeout
        while True:
            blocked = None
            begun = sum(1 for t in self.tasks if t["m"].get("segs")) if self.max_decoding > 0 else 0
            for task in self.tasks:
                if self.max_decoding > 0 and begun >= self.max_decoding and not task["m"].get("segs"):
                    continue          # would be one more batch holding a context ring: wait for one of the begun ones to end
                if self._is_ready(task):
                    if self._has_room(task):
                        return task
                    blocked = blocked or task
            if blocked is not None:       # work is there, its stream is `depth` pieces behind: wait for a piece to finish
                self.in_flight[blocked["sid"]].popleft().synchronize()
                continue
            for job in self.jobs:
                if job["error"] is not None:
                    raise job["error"]
            if time.perf_counter() > t_end:
                first = self.tasks[0]
                self._wait_job(first["job"], "the token stage")   # raises with the stacks if it was never enqueued
                raise RuntimeError(f"run_pipelined: no tokens for frame {first['need'] - 1} of batch {first['m']['i']} within {self.timeout:.0f} s")
            time.sleep(2e-4)

    def _advance(self, task):
        """One piece of a batch's decode (its conditioning frames / one new frame) behind the event of the tokens it reads."""
        job, m = task["job"], task["m"]
        ev = job["feed"].events[task["need"] - 1]
        if ev is None:
            if job["error"] is not None:
                raise job["error"]
            raise RuntimeError(f"run_pipelined: the token stage of batch {m['i']} ended without frame {task['need'] - 1}")
        want = self.cu_limit if any(not j["done"].is_set() for j in self.jobs) else 0
        if want != self.budget_now:
            self._budget(want)
            self.budget_now = want
        clip = None
        st = self.dec_streams[task["sid"]]
        with torch.cuda.stream(st):
            st.wait_event(ev)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            try:
                task["need"] = next(task["gen"])
            except StopIteration as fin:
                clip = fin.value
            e1.record()
        m["segs"].append((e0, e1))
        self.in_flight[task["sid"]].append(e1)
        if clip is not None:
            self.tasks[:] = [t for t in self.tasks if t is not task]
            self.finished[m["i"]] = (task, clip)
            job["left"] -= 1
            if job["left"] == 0:
                keep = [j for j in self.jobs if j is not job]
                self.jobs.clear()
                self.jobs.extend(keep)
            if self.debug:
                print(f"[pipeline] batch {m['i']} (chain {job['chain']}) decoded in {len(m['segs'])} pieces", file=sys.stderr, flush=True)

    # ------------------------------------------------------------------ stage 4: results, in batch order
    def _hand_out(self):
        while self.next_out in self.finished:
            task, fake = self.finished.pop(self.next_out)
            job, m = task["job"], task["m"]
            ws = m["ws"]
            out = {"real": ws["data"]["vid"], "fake": fake, "rec": None, "enc_code": ws["encoded"]["code"],
                   "real_state": ws["data"].get("state") if self.opt.state else None, "index": m["i"]}
            with torch.cuda.stream(self.dec_streams[task["sid"]]):
                if self.rec_pass:
                    out["rec"] = self.gen.reconstruct(ws)
                out["finished"] = self.finish(m["i"], fake) if self.finish is not None else None
                for t in (fake["vid"], fake["code"]) + ((out["rec"]["vid"],) if out["rec"] is not None else ()):   # handed to the caller's stream
                    t.record_stream(self.entry)
                if self.consume is not None:   # a long run: the caller takes every batch as it comes, nothing is kept here
                    self.consume(out)
                    out = {"index": m["i"], "finished": out["finished"]}
            m["ev"]["d0"], m["ev"]["d1"] = m["segs"][0][0], m["segs"][-1][1]
            self.results.append(out)
            if self.consume is not None and len(self.results) > 1:
                self.results[-2]["finished"] = None      # only the last hand-over's handle is kept
                del self.timings[:-64]                   # ... and the events of the last batches
            self.timings.append(dict(m["ev"], t0=job["t0"], t1=job["t1"], group=len(job["members"]), segs=m["segs"],
                                     hand_overs=len({id(e) for e in job["feed"].events if e is not None})))
            m["ws"] = None
            self.next_out += 1
