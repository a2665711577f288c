"""Generator: the inference driver (reference helpers/generator.py:27-394) on the MI355X path.

`Generator(opt).run()` and `.generate_vid(data, global_iter)` keep the reference's meaning:
encode all frames -> crop to the conditioning window -> autoregressive token synthesis ->
flow-guided decoding (-> optional teacher-forced "rec" decode) -> uint8 pack / save.

Differences by design:
  * `generate_vid` RETURNS its results (the reference only writes mp4 files), so the batch can be
    all-gathered over RCCL and benchmarked with nothing leaving HBM;
  * one process per GPU shards the generation batch along B (`Engine.shard_batch`) and the
    decoded clips of all ranks are all-gathered once at the end (`Engine.all_gather_clips`);
  * with no dataset on disk the input is the seeded synthetic tensor BASELINE.md specifies
    (`vid = rand(B,T,3,H,W)*2-1`).
State / STFT / layout / deblurring conditioning are "next" rows and raise.
"""
import os
import sys
import time
from itertools import cycle

import torch
import torch.nn.functional as F

from ccvs_amd.tools.options import Options
from ccvs_amd.tools.engine import Engine
from ccvs_amd.models.skip_vid_generator.models.quantized_video_model import QVidModel
from ccvs_amd.models.skip_vid_generator.models.transformer_model import Transformer
from ccvs_amd import ops


class _FrameFeed:
    """The tokens of one token group on their way to the decoder, frame by frame (`Generator.run_pipelined`).  The token
    stream copies every finished frame out of the loop's own buffer (which the chain's next group overwrites) and records an
    event behind the copy; the decoder's thread sees `flags[f]` once `events[f]` exists and waits for it on its stream."""

    def __init__(self, rows, frames, frame_tokens, device):
        import threading
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


class Generator:
    def __init__(self, opt):
        self.opt = opt["transformer"]
        self.qvid_opt = opt["qvid_generator"]
        self.state_opt = opt.get("state_estimator")
        self.stft_ae_opt = opt.get("stft_ae")
        self.iter_fn = cycle if self.opt.iter_function == "cycle" else iter
        self.engine = None
        self.vid_model = None
        self.transformer_model = None
        self.timings = {}
        self.stft_model = None
        self.state_model = None
        self._dec_stream, self._chains, self._warm_keys = None, [], set()
        self.last_cu_limit, self.last_lanes, self.last_chains = 0, 0, 0
        for flag in ("layout", "deblurring"):
            if getattr(self.opt, flag, False):
                raise NotImplementedError(f"--{flag} is not on the MI355X path (SURVEY 8f); --x_state / --x_stft [--keep_state] / --x_cat are")

    # ------------------------------------------------------------------ models / data
    def build_models(self, is_main=True):
        self.vid_model = QVidModel(self.qvid_opt, is_train=False, is_main=is_main, logger=None).eval()
        if not self.opt.rec_only:
            self.transformer_model = Transformer(self.opt, is_train=False, is_main=is_main, logger=None).eval()
        if self.opt.state:  # generator.py:266-269
            from ccvs_amd.models.skip_vid_generator.models.state_model import StateModel
            self.state_model = StateModel(self.state_opt, is_train=False, is_main=is_main, logger=None).eval()
        if self.opt.stft:  # generator.py:271-274
            from ccvs_amd.models.skip_vid_generator.models.stft_model import StftModel
            self.stft_model = StftModel(self.stft_ae_opt, is_train=False, is_main=is_main, logger=None).eval()
        return self

    def synthetic_batch(self, batch, seed=1, first_clip=0):
        """`vid = rand(B,T,3,H,W)*2-1`, one generator per global clip index so that the data does
        not depend on how the batch is sharded over ranks."""
        o = self.opt
        h = o.max_dim
        w = int(h * o.aspect_ratio)
        frames = []
        for i in range(batch):
            g = torch.Generator().manual_seed(seed * 1000003 + first_clip + i)
            frames.append(torch.rand(o.vid_len, 3, h, w, generator=g) * 2 - 1)
        return {"vid": torch.stack(frames)}

    def get_data_info(self, phase, data_type, fold=None, num_folds=None):
        """Synthetic stand-in for the reference's dataset/loader factory (generator.py:232-246)."""
        bs = self.opt.batch_size_vid * self.opt.batch_size_valid_mult
        lo, hi = self.engine.shard_batch(bs) if self.engine is not None else (0, bs)
        per = hi - lo
        seed = getattr(self.opt, "seed", 0) + 1

        def loader():
            it = 0
            while True:
                yield self.synthetic_batch(per, seed=seed + it, first_clip=lo)
                it += 1
        return {"dataloader": None, "datasampler": None, "epoch": 0, "phase": phase, "data_type": data_type,
                "batch_size_per_gpu": per, "loader_iter": loader(), "fold": fold, "num_folds": num_folds}

    def next_batch(self, data_info):
        return next(data_info["loader_iter"])

    # ------------------------------------------------------------------ the hot path
    def noise_key(self, global_iter):
        """Philox key of the in-kernel sampling noise for batch `global_iter`: splitmix64 of (seed, iteration), the same
        on every rank.  With the rank's first global clip index as row offset (`first_clip`), a clip's token stream
        depends on (seed, iteration, global clip index) only -- not on the world size."""
        z = (int(getattr(self.opt, "seed", 0)) * 0x9E3779B97F4A7C15 + int(global_iter) + 0x632BE59BD9B4E019) & (2**64 - 1)
        z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & (2**64 - 1)
        z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & (2**64 - 1)
        z ^= z >> 31
        return z & 0xffffffff, z >> 32

    def first_clip(self, batch):
        """Global index of this rank's clip 0 when every rank holds `batch` clips."""
        return self.engine.rank * batch if self.engine is not None else 0

    def _seed_sampler(self, batch, global_iter, net_t=None):
        if net_t is None and self.transformer_model is not None:
            net_t = self.transformer_model.net_t
        if net_t is not None:
            net_t.noise_key, net_t.noise_call = self.noise_key(global_iter), 0
            net_t.row_offset = self.first_clip(batch)

    @torch.no_grad()
    def condition(self, data):
        """First half of generate_vid (generator.py:57-130): encode every frame, tokenise the ancillary streams, crop to the
        conditioning window.  Returns the working set of one batch as a dict: `encoded`, `cropped`, `total_len`,
        `cond_len`, `crop_prop`."""
        opt, qopt = self.opt, self.qvid_opt
        if getattr(opt, "layout", False) or getattr(opt, "deblurring", False):
            raise NotImplementedError("layout / deblurring conditioning is not on the MI355X path (SURVEY 8f)")
        if opt.down_size is not None:  # generator.py:60-66
            vid = data["vid"].cuda()
            bs, t = vid.shape[:2]
            img = vid.view(-1, *vid.shape[2:])
            img = F.interpolate(img, size=opt.down_size, mode='bilinear')
            img = F.interpolate(img, size=vid.shape[-2:], mode='bilinear')
            data["vid"] = img.view(bs, t, *vid.shape[2:])
        size = int(torch.prod(torch.tensor(qopt.z_shape)))
        n_keep = self._frames_to_encode(data["vid"].shape[1], size)
        if n_keep < data["vid"].shape[1]:                                   # --encode_all false: the conditioning frames only
            enc_in = {k: v for k, v in data.items() if k != "vid"}
            enc_in["vid"] = data["vid"][:, :n_keep].contiguous()
            encoded_data = self.vid_model(enc_in, mode='vid_encoder')
            data["vid"] = data["vid"].cuda()
        else:
            encoded_data = self.vid_model(data, mode='vid_encoder')        # encode all frames
        n_frames = data["vid"].shape[1]                                     # the crops below are fractions of the WHOLE clip
        if opt.state:                                                       # generator.py:73-77: estimate + quantise the state
            encoded_data.update(self.state_model(encoded_data, mode='vid_encoder'))
            data.update(self.state_model(encoded_data, mode='vid_decoder'))
        if opt.stft:                                                        # generator.py:78-80
            encoded_data.update(self.stft_model(data, mode='vid_encoder'))

        # generator.py:83-102
        cond_step, t_step = (1, opt.vid_len - 1) if opt.p2p else (0, opt.vid_len)
        total_len = (cond_step + t_step) * size
        cond_len = cond_step * size
        if opt.state or opt.stft:                                           # generator.py:91-92
            total_len += t_step * opt.state_size
        if opt.gen_from_img:
            crop_prop = opt.cond_len / size
        else:
            crop_prop = opt.cond_len / (size * opt.vid_len)
        cropped = {}
        cropped["code"] = encoded_data["code"][:, :int(crop_prop * (size * n_frames if n_keep < n_frames else encoded_data["code"].size(1)))]
        encoded_data["inter"] = [feat[:, :int(crop_prop * (n_frames if n_keep < n_frames else feat.size(1)))].contiguous() for feat in encoded_data["inter"]]
        cropped["inter"] = encoded_data["inter"]
        if opt.p2p:
            cropped["cond_code"] = encoded_data["code"][:, -opt.z_chunk:]
            # (sic) the reference reads the ALREADY CROPPED list here (generator.py:101,105), i.e. the
            # skip features of the last conditioning frame, not of the end frame; kept for parity
            cropped["cond_inter"] = [feat[:, -1:].contiguous() for feat in encoded_data["inter"]]
            cropped["delta_length_cond"] = torch.tensor([opt.vid_len - 1]).repeat(cropped["code"].size(0))

        if opt.state or opt.stft:                                           # generator.py:107-117
            if opt.keep_state:
                cropped["state_code"] = encoded_data["state_code"]
            elif opt.custom_state:
                init_state = self.state_model(encoded_data, mode='vid_decoder')["state"][:, [0]]
                custom_state = square_trajectory(init_state, opt.vid_len)
                cropped["state_code"] = self.state_model(custom_state, mode='vid_encoder')["state_code"]
            else:
                cropped["state_code"] = encoded_data["state_code"][:, :int(crop_prop * encoded_data["state_code"].size(1))]
        if opt.cat:                                                         # generator.py:123-126: class-conditional generation
            if "vid_lbl" not in data:
                data["vid_lbl"] = torch.randint(low=0, high=len(opt.categories), size=[encoded_data["code"].size(0)])
            cropped["vid_lbl"] = data["vid_lbl"]
        return {"data": data, "encoded": encoded_data, "cropped": cropped, "total_len": total_len, "cond_len": cond_len,
                "crop_prop": crop_prop}

    def _frames_to_encode(self, n_frames, size):
        """Frames of the input clip the encoder has to see.  The reference encodes all of them (generator.py:69); with
        `--encode_all false` only those the conditioning crop keeps (generator.py:93-99: the first crop_prop of the codes and of
        the skip features) -- when nothing else reads the rest: no rec pass, no state / audio stream estimated from the clip, no
        end frame (point-to-point), no single-image mode."""
        opt = self.opt
        if getattr(opt, "encode_all", True) or getattr(opt, "rec_pass", True) or opt.rec_only or opt.state or opt.stft or opt.p2p or opt.gen_from_img:
            return n_frames
        crop_prop = opt.cond_len / (size * opt.vid_len)
        n_code = -(-int(crop_prop * size * n_frames) // size)             # frames holding the kept tokens
        n_inter = int(crop_prop * n_frames)                                # frames of skip features kept
        return min(n_frames, max(1, n_code, n_inter))

    @torch.no_grad()
    def decode_codes(self, ws, code, state_code=None):
        """Flow-guided decode of a full token sequence `code` for the working set `ws` of `condition()`
        (generator.py:161-169): the second half of the synthesis, also used teacher-forced by the tests."""
        opt = self.opt
        dec_in = dict(ws["cropped"])
        dec_in["code"] = code
        if state_code is not None:
            dec_in["state_code"] = state_code
        fake_data = self.vid_model(dec_in, mode='vid_decoder')
        fake_data["code"], fake_data["state_code"] = code, state_code
        if opt.p2p:
            fake_data["vid"] = torch.cat([fake_data["vid"], ws["data"]["vid"][:, -1:]], dim=1)
        if opt.state and state_code is not None:                            # generator.py:168-169
            fake_data.update(self.state_model({"state_code": state_code}, mode='vid_decoder'))
        return fake_data

    def _decode_codes_stream(self, ws, code_of, final):
        """`decode_codes` as a generator for a clip whose tokens arrive frame by frame (`QVidModel.decode_stream`): `code_of(lo,
        hi)` -> tokens of frames lo .. hi - 1, `final()` -> (code, state_code) of the whole clip once the token stage is done."""
        opt = self.opt
        dec_in = {k: v for k, v in ws["cropped"].items() if k != "code"}
        fake_data = yield from self.vid_model.decode_stream(dec_in, code_of)
        code, state_code = final()
        fake_data["code"], fake_data["state_code"] = code, state_code
        if opt.p2p:
            fake_data["vid"] = torch.cat([fake_data["vid"], ws["data"]["vid"][:, -1:]], dim=1)
        if opt.state and state_code is not None:                            # generator.py:168-169
            fake_data.update(self.state_model({"state_code": state_code}, mode='vid_decoder'))
        return fake_data

    @torch.no_grad()
    def reconstruct(self, ws):
        """The teacher-forced "rec" decode of the clip's own codes (generator.py:172-189)."""
        opt = self.opt
        encoded_data, cropped, data = ws["encoded"], ws["cropped"], ws["data"]
        rec = {"inter": cropped["inter"]}
        if opt.p2p:
            rec["code"] = encoded_data["code"][:, :-opt.z_chunk].contiguous()
            rec["cond_code"] = cropped["cond_code"]
            rec["cond_inter"] = cropped["cond_inter"]
        else:
            rec["code"] = encoded_data["code"]
        if opt.state or opt.stft:
            rec["state_code"] = encoded_data["state_code"]
        rec_data = self.vid_model(rec, mode='vid_decoder')
        if opt.p2p:
            rec_data["vid"] = torch.cat([rec_data["vid"], data["vid"][:, -1:]], dim=1)
        if opt.state:
            rec_data["state"] = data["state"]
        return rec_data

    @torch.no_grad()
    def generate_vid(self, data, global_iter=0, save=False):
        opt = self.opt
        self._seed_sampler(data["vid"].shape[0], global_iter)
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
        ev[0].record()
        ws = self.condition(data)
        ev[1].record()
        encoded_data, cropped = ws["encoded"], ws["cropped"]

        fake_data, rec_data = None, None
        if not opt.rec_only:
            if opt.step_by_step:
                fake_data = self._step_by_step(data, cropped, ws["crop_prop"], ws["total_len"], ws["cond_len"])
                ev[2].record()
                if opt.p2p:
                    fake_data["vid"] = torch.cat([fake_data["vid"], data["vid"][:, -1:]], dim=1)
            else:
                fake_encoded = self.transformer_model(cropped, mode='inference', total_len=ws["total_len"], show_progress=False)
                ev[2].record()
                state_code = fake_encoded.get("state_code")
                if state_code is not None and 0 in state_code.size():
                    state_code = None
                fake_data = self.decode_codes(ws, fake_encoded["code"], state_code)
        else:
            ev[2].record()
        ev[3].record()

        if not opt.gen_from_img and (getattr(opt, "rec_pass", True) or opt.rec_only):
            rec_data = self.reconstruct(ws)

        self._events = ev
        out = {"real": data["vid"], "fake": fake_data, "rec": rec_data, "enc_code": encoded_data["code"],
               "real_state": data.get("state") if opt.state else None}
        if save:
            self.save_results(out, global_iter)
        return out

    # ------------------------------------------------------------------ several batches in flight
    def _seed_sampler_group(self, batch, iters, net_t):
        """Sampler words of a token stage over the stacked rows of several batches (`iters`: their global iterations): one
        (key, first global clip) pair per row group -- exactly the words each batch has when it runs alone."""
        net_t.noise_key = [self.noise_key(i) for i in iters]
        net_t.row_offset = [self.first_clip(batch)] * len(iters)
        net_t.noise_call = 0

    def _token_group_size(self, batch, lanes):
        """Batches whose token loops run as ONE loop (`lanes`, capped): the stacked rows must fit one decode step (256 rows)
        and be a few row blocks of the weight-stream GEMM at most (64 rows = one pass over the weights; more rows re-read them from
        L2: 128 by default); host-drawn sampling noise is one generator stream per BATCH in the reference's order, so it is not stacked."""
        opt = self.opt
        if getattr(opt, "sample", False) and getattr(self.transformer_model, "sample_noise", "host") != "device":
            return 1
        if getattr(opt, "beam_size", None) is not None:
            return 1
        max_rows = int(os.environ.get("CCVS_PIPELINE_MAX_ROWS", "128"))   # (64 until Kinetics measured 2 x 64 rows x 3 chains: 362 -> 373 frames/s; up to 256)
        return max(1, min(int(lanes), min(max_rows, 256) // max(batch, 1)))

    @staticmethod
    def _lanes_that_fit(lanes, chains, batch, height, width, total_bytes, frac=0.7, bytes_per_clip_pixel=10.0e3):
        """Lanes per token group such that the batches in flight -- up to lanes x (chains + 2): two groups in the token loops, one
        encoded behind them, one being decoded -- fit `frac` of the device memory.  A batch in flight holds its context rings, skip
        features, token caches and the decoder's intermediates: ~10 KB per clip and pixel measured (BAIR: 174 GB at 16 batches of
        16 x 256^2 in flight, weights and captured steps included).  BAIR at batch 16 keeps 4 lanes on 288 GB; batch 32 gets 2."""
        per_batch = float(batch) * height * width * bytes_per_clip_pixel
        while lanes > 1 and lanes * (chains + 2) * per_batch > frac * total_bytes:
            lanes -= 1
        return lanes

    def _token_chain(self, k):
        """Token chain k: (Transformer, stream).  A chain runs one token group at a time; chains run beside each other.  Chain 0
        is the model itself, the others are shallow copies that SHARE its parameters (and their packed forms) but own their
        engine state -- KV caches, captured decode steps, sampler words."""
        import copy
        while len(self._chains) <= k:
            tr = self.transformer_model
            if self._chains:
                net = copy.copy(tr.net_t)                  # same Parameter objects / blocks, separate engine state
                net.drop_engine_state()
                lane = copy.copy(tr)
                lane._modules = dict(tr._modules)
                lane._modules["net_t"] = net
                tr = lane
            dev = torch.device("cuda", torch.cuda.current_device())
            prio = int(os.environ.get("CCVS_PIPELINE_PRIORITIES", "0,-1").split(",")[0])
            self._chains.append((tr, torch.cuda.Stream(device=dev, priority=prio)))
        return self._chains[k]

    @torch.no_grad()
    def run_pipelined(self, batches, first_iter=0, cu_limit=None, finish=None, lanes=None, chains=None, ramp=None):
        """generate_vid over a sequence of batches with SEVERAL batches in flight on one GPU.  The token loop of a batch is a
        latency-bound chain of ~120 small dependent launches per token that streams every GPT weight once per token and cannot
        fill the chip; the flow-guided decoder is a throughput-bound stream of large MFMA / HBM kernels.  So

          * the token loops of `lanes` consecutive batches run as ONE loop over their stacked rows ("token group": one KV cache
            of lanes x B rows, one captured decode step, per-group sampler words -- `ccvs_gpt_decode.groups`): the weights are
            streamed once per token for all of them instead of once per batch;
          * `chains` such loops run beside each other on their own streams, each fed by its own worker thread (beside the
            decoder a dependent launch waits longer for its memory: independent chains hide each other's latency);
          * meanwhile the first decode stream encodes the batches of the groups to come (chains + 1 groups in front of the decoder, so that a
            chain never waits for its encoder) and decodes -- FRAME BY FRAME, as the tokens arrive: the decoder needs the tokens
            of frame t only for frame t (`QVidModel.decode_frames`), so a running token loop hands every finished frame to a
            `_FrameFeed` (a copy out of its buffer + an event, `GPT.progress`) and the decode of a batch is a generator advanced
            one frame at a time, the oldest batch whose next frame is there first.  In the steady state that is the old
            order (the oldest batch has all its tokens); at the start the decoder follows the first token loops one frame
            behind instead of idling through a whole token stage (`CCVS_PIPELINE_STREAM=0`: decode when the stage is done).
            The host stays `CCVS_PIPELINE_DEPTH` pieces ahead of a decode stream, so that "there" is judged late;
          * the decode of batch i runs on decode stream i % D (`CCVS_PIPELINE_DEC_STREAMS`; the encodes on the first): the small
            launches of one batch's coarse pyramid levels run under the large ones of another's fine levels.
        `ramp`: sizes of the first groups (e.g. (1, 2): the decoder gets its first batch after one short token stage instead
        of idling through a full one); then every group has `lanes` batches.

        Per batch the work is exactly generate_vid's -- encode, crop, token synthesis, decode -- and so are the results: a
        row's arithmetic does not depend on the rows it shares a launch with and every batch keeps its own sampler words
        (tests/test_pipeline_gpu.py checks bit-equality with the serial schedule).

        `cu_limit` > 0 caps everything on the decode streams to that many CUs (`ccvs_stream_cu_limit`) while a token loop is in flight
        (default 0).  A token stage is enqueued by a worker thread because a hipGraph launch blocks its caller once the
        stream's queue is a few dozen steps deep, and the decoder's launches must not wait behind that.  Whenever a token
        stage would have to capture its decode step (first use of a chain with a group size, changed weights or sampler),
        the capture is done up front from this thread before the workers start (`warm_up`): graph capture and first-time
        launches never race with launches of another thread, none falls into the steady state.  Every wait on a worker has a time
        limit (`CCVS_PIPELINE_TIMEOUT`, 600 s) and raises with the stage and batches it was waiting for, after dumping every
        thread's stack: the schedule cannot hang silently.

        batches: iterable of data dicts.  finish(i, out) -> anything: called on the decode stream of batch i when its clip is
        decoded, in batch order on every rank (pack / all-gather); its return values are collected.  Returns the list of per-batch results
        ({"fake", "enc_code", "finished", "index"}); the rec pass is not run here."""
        import queue
        import threading
        from collections import deque
        opt = self.opt
        if opt.step_by_step or opt.rec_only:
            raise NotImplementedError("run_pipelined covers the plain synthesis schedule (use generate_vid for step_by_step / rec_only)")
        dev = torch.device("cuda", torch.cuda.current_device())
        if lanes is None:
            lanes = int(os.environ.get("CCVS_PIPELINE_LANES", "4"))
        if chains is None:
            chains = int(os.environ.get("CCVS_PIPELINE_CHAINS", "2"))
        if ramp is None:
            ramp = tuple(int(v) for v in os.environ.get("CCVS_PIPELINE_RAMP", "").split(",") if v)
        chains = max(1, chains)
        # stream D and its twins: the encodes run on the first, the decode of batch i on stream i % n (two decodes side by side:
        # the small launches of one's coarse pyramid levels run under the other's large ones)
        # (default: two for frames of 128^2 and more -- there the decoder is the longer stage; one for the 64^2 configurations, whose
        # token loops are, and lose what a second decode takes: Kinetics 353 against 361 frames/s)
        n_dec = os.environ.get("CCVS_PIPELINE_DEC_STREAMS", "")
        n_dec = max(1, int(n_dec)) if n_dec else (2 if int(getattr(self.opt, "max_dim", 256)) >= 128 else 1)
        if len(getattr(self, "_dec_streams", ())) < n_dec:
            prio = [int(v) for v in os.environ.get("CCVS_PIPELINE_PRIORITIES", "0,-1").split(",")]   # (token streams, decode streams)
            have = list(getattr(self, "_dec_streams", ()))
            self._dec_streams = have + [torch.cuda.Stream(device=dev, priority=prio[1]) for _ in range(n_dec - len(have))]
            self._dec_stream = self._dec_streams[0]
        dec_streams = self._dec_streams[:n_dec]
        s_dec = dec_streams[0]
        # the encodes: on the first decode stream, or (CCVS_PIPELINE_ENC_STREAM=1) on a stream of their own
        s_enc = s_dec
        if os.environ.get("CCVS_PIPELINE_ENC_STREAM", "0") == "1":
            if getattr(self, "_enc_stream", None) is None:
                self._enc_stream = torch.cuda.Stream(device=dev, priority=int(os.environ.get("CCVS_PIPELINE_PRIORITIES", "0,-1").split(",")[1]))
            s_enc = self._enc_stream
        chain_list = [self._token_chain(k) for k in range(chains)]
        if cu_limit is None:   # 0 = no budget
            cu_limit = int(os.environ.get("CCVS_PIPELINE_CU_LIMIT", "0"))
        timeout = float(os.environ.get("CCVS_PIPELINE_TIMEOUT", "600"))
        entry = torch.cuda.current_stream()
        for st in dec_streams + [s_enc]:
            st.wait_stream(entry)
        for _, st in chain_list:
            st.wait_stream(entry)
        results, timings = [], []
        it = iter(batches)
        held = []         # a batch read ahead that did not fit its group (ragged size): first of the next group
        index = first_iter
        n_groups = 0
        debug = os.environ.get("CCVS_PIPELINE_DEBUG", "0") == "1"
        queues = [queue.Queue() for _ in range(chains)]
        self.last_cu_limit, self.last_lanes, self.last_chains, self.last_dec_streams = cu_limit, lanes, chains, len(dec_streams)

        def budget(n):   # CU budget of everything submitted to the decode streams from now on
            for st in set(dec_streams + [s_enc]):
                ops.stream_cu_limit(st, n)

        def capture_key(chain, rows, groups):
            """What a captured decode step depends on (the cache length follows from the options): such a key is warmed once."""
            net = chain_list[chain][0].net_t
            ver = sum(p_._version for p_ in net.parameters())
            return (chain, rows, groups, opt.vid_len, opt.z_len, opt.cond_len, bool(opt.p2p), bool(opt.sample), opt.top_k, float(opt.temperature),
                    chain_list[chain][0].sample_noise, ver)

        abort = threading.Event()   # set on the error path: queued token stages are dropped instead of run
        stream_frames = os.environ.get("CCVS_PIPELINE_STREAM", "1") != "0"   # 0: a batch is decoded when its whole token stage is done
        frame_tokens = int(self.qvid_opt.z_shape[0]) * int(self.qvid_opt.z_shape[1])

        def worker(chain):
            tr, s_tok = chain_list[chain]
            torch.cuda.set_device(dev)
            while True:
                job = queues[chain].get()
                if job is None:
                    return
                if abort.is_set():
                    job["error"] = RuntimeError("run_pipelined: aborted")
                    job["done"].set()
                    job["feed"].release()
                    continue
                feed = job["feed"]
                try:
                    with torch.cuda.stream(s_tok), torch.no_grad():
                        s_tok.wait_event(job["enc_done"])
                        members = job["members"]
                        tok_in = stack_inputs([m["ws"]["cropped"] for m in members])
                        self._seed_sampler_group(job["batch"], [m["i"] for m in members], tr.net_t)
                        # one window of tokens, frame tokens only: the loop reports every finished frame (mingpt `progress`)
                        by_frame = stream_frames and int(job["total_len"]) <= opt.z_len and not (opt.state or opt.stft)
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
                    tr.net_t.noise_key, tr.net_t.row_offset = None, 0
                    job["done"].set()
                    feed.release()

        def wait_job(job, what):
            if not job["done"].wait(timeout):
                import faulthandler
                faulthandler.dump_traceback(file=sys.stderr, all_threads=True)
                raise RuntimeError(f"run_pipelined: {what} of batches {[m['i'] for m in job['members']]} (chain {job['chain']}) was not enqueued within "
                                   f"{timeout:.0f} s (stacks of all threads are on stderr)")
            if job["error"] is not None:
                raise job["error"]

        def submit_group():
            """Encode + crop the next group's batches on D and queue their token stage; None when the input is exhausted."""
            nonlocal index, n_groups
            members = []
            group = None
            while group is None or len(members) < group:
                data = held.pop() if held else next(it, None)
                if data is None:
                    break
                if group is None:
                    want = ramp[n_groups] if n_groups < len(ramp) else lanes
                    group = self._token_group_size(data["vid"].shape[0], min(want, lanes))
                elif data["vid"].shape[0] != members[0]["batch"]:
                    held.append(data)         # a ragged batch starts the next group
                    break
                chain = n_groups % chains
                ev = {k: torch.cuda.Event(enable_timing=True) for k in ("e0", "e1", "d0", "d1")}
                with torch.cuda.stream(s_enc):
                    ev["e0"].record()
                    ws = self.condition(data)
                    ev["e1"].record()
                for t in (ws["cropped"]["code"], ws["cropped"].get("cond_code"), ws["cropped"].get("state_code")):
                    if torch.is_tensor(t) and t.is_cuda:
                        t.record_stream(chain_list[chain][1])
                members.append({"i": index, "ws": ws, "ev": ev, "batch": data["vid"].shape[0]})
                index += 1
            if not members:
                return None
            if any(m["ws"]["total_len"] != members[0]["ws"]["total_len"] for m in members):
                raise RuntimeError("run_pipelined: the batches of a token group must share total_len")
            chain = n_groups % chains
            n_groups += 1
            if is_cold(chain, members[0]["batch"], len(members)):
                # a batch size warm_up() has not seen (ragged input): the capture must not race with a worker's launches --
                # wait until every queued token stage is enqueued (the workers then sleep on their queues), capture here
                for queued in jobs:
                    wait_job(queued, "the token stage (before a capture for a new batch size)")
                warm(chain, members[0]["batch"], len(members), members[0]["ws"])
            with torch.cuda.stream(s_enc):
                feed = _FrameFeed(len(members) * members[0]["batch"], opt.vid_len, frame_tokens, dev)
                feed.codes.record_stream(chain_list[chain][1])
                enc_done = torch.cuda.Event()
                enc_done.record()
            job = {"members": members, "batch": members[0]["batch"], "total_len": members[0]["ws"]["total_len"], "chain": chain,
                   "enc_done": enc_done, "t0": torch.cuda.Event(enable_timing=True), "t1": torch.cuda.Event(enable_timing=True),
                   "codes": None, "error": None, "done": threading.Event(), "feed": feed, "left": len(members)}
            queues[chain].put(job)
            return job

        def stack_inputs(cropped):
            tok_in = {}
            for key in ("code", "cond_code", "state_code", "vid_lbl", "delta_length_cond"):
                if key in cropped[0]:
                    tok_in[key] = cropped[0][key] if len(cropped) == 1 else torch.cat([c[key] for c in cropped], dim=0)
            return tok_in

        def is_cold(c, nb, g):
            # the captured steps live in the engine's cache of that row count: ask the engine, not only the side set (a serial
            # generate_vid with the same rows but another group count or a longer sequence, the eviction at 8 entries or
            # drop_engine_state() rebuild the cache and drop its graphs -- the capture would then happen in a worker thread)
            cache = chain_list[c][0].net_t._caches.get(nb * g)
            live = cache is not None and cache.get("G") == g and len(cache.get("graphs", {})) > 0
            key = capture_key(c, nb * g, g)
            if not live:
                self._warm_keys.discard(key)
            return key not in self._warm_keys

        def warm(c, nb, g, ws):
            """Capture the decode step of chain c for g stacked batches of nb clips (inputs: the working set of one such batch),
            from this thread; nothing is replayed, the sampler words are put back."""
            tr, s_tok = chain_list[c]
            s_tok.wait_stream(s_enc)
            with torch.cuda.stream(s_tok):
                self._seed_sampler_group(nb, list(range(g)), tr.net_t)
                tr.net_t.warm_only = True
                try:
                    tr(stack_inputs([ws["cropped"]] * g), mode='inference', total_len=ws["total_len"])
                finally:
                    tr.net_t.warm_only = False
                    tr.net_t.noise_key, tr.net_t.row_offset = None, 0
            s_tok.synchronize()
            self._warm_keys.add(capture_key(c, nb * g, g))
            s_enc.wait_stream(s_tok)

        def warm_up():
            """Capture the decode step of every (chain, group size) this run can use that is not captured yet, from this thread,
            before any worker exists: first-time launches and graph captures never race with another thread's launches, and
            no capture falls into the steady state.  The first batch is encoded once more for it; results are discarded.  (A
            later batch of another size: `submit_group` waits until no worker is launching and captures then.)"""
            first = next(it, None)
            if first is None:
                return
            held.append(first)
            nb = first["vid"].shape[0]
            nonlocal lanes
            fit = self._lanes_that_fit(lanes, chains, nb, first["vid"].shape[-2], first["vid"].shape[-1], torch.cuda.mem_get_info(dev)[1],
                                       frac=float(os.environ.get("CCVS_PIPELINE_MEM_FRAC", "0.7")))
            if fit < lanes:
                print(f"[pipeline] {lanes} -> {fit} batches per token group: {lanes * (chains + 2)} batches of {nb} clips in flight would not fit "
                      "the device memory (CCVS_PIPELINE_MEM_FRAC)", file=sys.stderr, flush=True)
                lanes = fit
                self.last_lanes = lanes
            sizes = sorted({self._token_group_size(nb, g) for g in range(1, lanes + 1)})
            cold = [(c, g) for c in range(chains) for g in sizes if is_cold(c, nb, g)]
            if not cold:
                return
            with torch.cuda.stream(s_enc):
                ws = self.condition({k: (v.clone() if torch.is_tensor(v) else v) for k, v in first.items()})
            for c, g in cold:
                warm(c, nb, g, ws)

        warm_up()
        threads = [threading.Thread(target=worker, args=(k,), name=f"ccvs-token-chain-{k}", daemon=True) for k in range(chains)]
        for th in threads:
            th.start()
        jobs = deque()        # submitted token groups with undecoded batches, oldest first
        tasks = []            # the decodes of their batches, oldest first
        in_flight = [deque() for _ in dec_streams]   # per decode stream: events behind the pieces of decoder work enqueued last
        finished = {}         # batch index -> (task, clip) decoded ahead of an earlier batch: handed out in index order
        depth = max(1, int(os.environ.get("CCVS_PIPELINE_DEPTH", "2")))
        state = {"exhausted": False, "next_out": first_iter, "budget": 0}

        def top_up():
            """Keep every chain busy and one more group encoded behind them (at most chains + 2 groups in flight)."""
            while (not state["exhausted"] and len(jobs) < chains + 2 and sum(1 for j in jobs if not j["done"].is_set()) < chains + 1):
                job = submit_group()
                if job is None:
                    state["exhausted"] = True
                    break
                jobs.append(job)
                tasks.extend({"job": job, "k": k, "m": m, "gen": None, "need": None, "sid": (m["i"] - first_iter) % len(dec_streams)}
                             for k, m in enumerate(job["members"]))

        def start(task):
            """The decode of one batch as a generator over its frames (`QVidModel.decode_frames`): it reads the tokens of a frame
            from the group's feed and the whole sequence, as the token stage returns it, when it is done."""
            job, k = task["job"], task["k"]
            nb, feed = job["batch"], job["feed"]

            def code_of(lo, hi):
                return feed.codes[k * nb:(k + 1) * nb, lo * frame_tokens:hi * frame_tokens]

            def final():
                wait_job(job, "the token stage")
                codes = job["codes"]
                state_all = codes.get("state_code")
                if state_all is not None and 0 in state_all.size():
                    state_all = None
                for t in (codes["code"], state_all):
                    if torch.is_tensor(t):
                        t.record_stream(st)
                st.wait_event(job["t1"])
                return codes["code"][k * nb:(k + 1) * nb], (state_all[k * nb:(k + 1) * nb] if state_all is not None else None)

            st = dec_streams[task["sid"]]
            m = task["m"]
            m["segs"] = []
            if st is not s_enc:     # encoded on one stream, decoded on this one
                st.wait_event(m["ev"]["e1"])
                for holder in (m["ws"]["cropped"], m["ws"]["encoded"], m["ws"]["data"]):
                    for v in holder.values():
                        for t in (v if isinstance(v, (list, tuple)) else (v,)):
                            if torch.is_tensor(t) and t.is_cuda:
                                t.record_stream(st)
                feed.codes.record_stream(st)
            with torch.cuda.stream(st):
                task["gen"] = self._decode_codes_stream(m["ws"], code_of, final)
                task["need"] = next(task["gen"])

        def is_ready(task):
            if task["gen"] is None:
                start(task)
            feed, f = task["job"]["feed"], task["need"] - 1
            if not feed.flags[f].is_set():
                return False
            ev = feed.events[f]
            return ev is None or ev.query()    # None: released without tokens (error path) -- `advance` raises

        def has_room(task):
            """The host stays `depth` pieces ahead of a decode stream, so that "ready" is judged late."""
            q = in_flight[task["sid"]]
            while q and q[0].query():
                q.popleft()
            return len(q) <= depth

        def pick():
            """The oldest batch whose next frame of tokens is there and whose stream has room; none: wait for one (the worker's
            error, if that is why)."""
            t_end = time.perf_counter() + timeout
            while True:
                blocked = None
                for task in tasks:
                    if is_ready(task):
                        if has_room(task):
                            return task
                        blocked = blocked or task
                if blocked is not None:       # work is there, its stream is `depth` pieces behind: wait for a piece to finish
                    in_flight[blocked["sid"]].popleft().synchronize()
                    continue
                for job in jobs:
                    if job["error"] is not None:
                        raise job["error"]
                if time.perf_counter() > t_end:
                    wait_job(tasks[0]["job"], "the token stage")   # raises with the stacks if it was never enqueued
                    raise RuntimeError(f"run_pipelined: no tokens for frame {tasks[0]['need'] - 1} of batch {tasks[0]['m']['i']} within {timeout:.0f} s")
                time.sleep(2e-4)

        def advance(task):
            """One piece of a batch's decode (its conditioning frames / one new frame) behind the event of the tokens it reads."""
            job, m = task["job"], task["m"]
            ev = job["feed"].events[task["need"] - 1]
            if ev is None:
                if job["error"] is not None:
                    raise job["error"]
                raise RuntimeError(f"run_pipelined: the token stage of batch {m['i']} ended without frame {task['need'] - 1}")
            want = cu_limit if any(not j["done"].is_set() for j in jobs) else 0
            if want != state["budget"]:
                budget(want)
                state["budget"] = want
            clip = None
            st = dec_streams[task["sid"]]
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
            in_flight[task["sid"]].append(e1)
            if clip is not None:
                tasks[:] = [t for t in tasks if t is not task]
                finished[m["i"]] = (task, clip)
                job["left"] -= 1
                if job["left"] == 0:
                    assert jobs[0] is job or any(j is job for j in jobs)
                    keep = [j for j in jobs if j is not job]
                    jobs.clear()
                    jobs.extend(keep)
                if debug:
                    print(f"[pipeline] batch {m['i']} (chain {job['chain']}) decoded in {len(m['segs'])} pieces", file=sys.stderr, flush=True)

        def hand_out():
            while state["next_out"] in finished:
                task, fake = finished.pop(state["next_out"])
                job, m = task["job"], task["m"]
                with torch.cuda.stream(dec_streams[task["sid"]]):
                    done = finish(m["i"], fake) if finish is not None else None
                    for t in (fake["vid"], fake["code"]):   # handed to the caller's stream
                        t.record_stream(entry)
                m["ev"]["d0"], m["ev"]["d1"] = m["segs"][0][0], m["segs"][-1][1]
                results.append({"fake": fake, "enc_code": m["ws"]["encoded"]["code"], "finished": done, "index": m["i"]})
                timings.append(dict(m["ev"], t0=job["t0"], t1=job["t1"], group=len(job["members"]), segs=m["segs"],
                                    hand_overs=len({id(e) for e in job["feed"].events if e is not None})))
                m["ws"] = None
                state["next_out"] += 1

        try:
            top_up()
            while tasks:
                task = pick()
                advance(task)
                hand_out()
                top_up()
        except BaseException:
            abort.set()   # the workers drop what is still queued; the error surfaces now, not after every queued token loop has run
            raise
        finally:
            for q_ in queues:
                q_.put(None)
            for th in threads:
                th.join(30.0 if abort.is_set() else timeout)
            budget(0)
            # (also on the error path: the caller's stream must not run ahead of work still queued on ours)
            for st in dec_streams + [s_enc]:
                entry.wait_stream(st)
            for _, st in chain_list:
                entry.wait_stream(st)
        self._pipeline_events = timings
        return results

    def pipeline_stage_ms(self):
        """Encode / transformer / decode milliseconds of the last run_pipelined, summed over its batches (the stages overlap in
        time).  A token stage serves the `group` batches of its token group at once: each is charged 1 / group of it."""
        torch.cuda.synchronize()
        out = {"encode": 0.0, "transformer": 0.0, "decode": 0.0}
        for ev in self._pipeline_events:
            out["encode"] += ev["e0"].elapsed_time(ev["e1"])
            out["transformer"] += ev["t0"].elapsed_time(ev["t1"]) / ev["group"]
            out["decode"] += sum(a.elapsed_time(b) for a, b in ev["segs"])   # its pieces; other batches' may lie between them
        return out

    def pipeline_timeline(self):
        """Per batch of the last run_pipelined: milliseconds since the first encode started at which its encode, token stage and
        decode began and ended (HIP events on the stages' own streams) -- where the schedule's bubbles are."""
        torch.cuda.synchronize()
        base = self._pipeline_events[0]["e0"]
        return [{k: round(base.elapsed_time(ev[k]), 1) for k in ("e0", "e1", "t0", "t1", "d0", "d1")} | {"group": ev["group"]}
                for ev in self._pipeline_events]

    def pipeline_token_groups(self):
        """[(batches in the group, milliseconds of its token stage)] of the last run_pipelined, one entry per token group."""
        torch.cuda.synchronize()
        seen, out = set(), []
        for ev in self._pipeline_events:
            if id(ev["t0"]) not in seen:
                seen.add(id(ev["t0"]))
                out.append((ev["group"], ev["t0"].elapsed_time(ev["t1"])))
        return out

    def _step_by_step(self, data, cropped, crop_prop, total_len, cond_len):
        """generator.py:132-159: predict one frame of tokens, decode it, re-encode it and overwrite
        the predicted tokens with the re-encoded ones."""
        opt = self.opt
        fake_vid = data["vid"][:, :int(crop_prop * data["vid"].size(1))]
        fake_enc = {"code": cropped["code"]}
        step_enc = {"inter": cropped["inter"]}
        if opt.p2p:
            fake_enc["cond_code"] = cropped["cond_code"]
            fake_enc["delta_length_cond"] = cropped["delta_length_cond"]
            step_enc["cond_inter"] = cropped["cond_inter"]
        all_codes = [cropped["code"]]
        for _ in range((total_len - opt.cond_len - cond_len) // opt.z_chunk):
            if opt.p2p and fake_enc["code"].size(1) > opt.z_len - 2 * opt.z_chunk:
                fake_enc["delta_length_cond"] = fake_enc["delta_length_cond"] - ((fake_enc["code"].size(1) - opt.z_len) // opt.z_chunk + 2)
                fake_enc["code"] = fake_enc["code"][:, -(opt.z_len - 2 * opt.z_chunk):]
            elif fake_enc["code"].size(1) > opt.z_len - opt.z_chunk:
                fake_enc["code"] = fake_enc["code"][:, -(opt.z_len - opt.z_chunk):]
            # like the reference, the returned dict REPLACES fake_enc (generator.py:150)
            fake_enc = self.transformer_model(fake_enc, mode='inference', total_len=fake_enc["code"].size(1) + opt.z_chunk)
            step_enc["code"] = fake_enc["code"][:, -opt.z_chunk:]
            step_dec = self.vid_model(step_enc, mode='vid_step_decoder')
            step_enc["inter"] = step_dec["inter"]
            fake_enc["code"][:, -opt.z_chunk:] = step_dec["code"]
            all_codes.append(step_dec["code"])
            fake_vid = torch.cat([fake_vid, step_dec["vid"]], dim=1)
        return {"vid": fake_vid, "code": torch.cat(all_codes, dim=1)}

    def stage_ms(self):
        """encode / transformer / decode milliseconds of the last `generate_vid` (the reference's
        unused t0..t3, generator.py:68-165), from HIP events."""
        torch.cuda.synchronize()
        e = self._events
        return {"encode": e[0].elapsed_time(e[1]), "transformer": e[1].elapsed_time(e[2]), "decode": e[2].elapsed_time(e[3])}

    # ------------------------------------------------------------------ output stage
    def save_results(self, out, global_iter):
        bs = out["real"].shape[0]
        for name in ("real", "fake", "rec"):
            item = out[name]
            if item is None:
                continue
            vid = item["vid"] if isinstance(item, dict) else item
            save_video_batch(vid, bs, global_iter, os.path.join(self.opt.result_path, name), self.opt.fps, True,
                             self.opt.imagenet_norm, [-1, 1], self.opt.dataset)
            state = out.get("real_state") if name == "real" else (item.get("state") if isinstance(item, dict) else None)
            if self.opt.state and state is not None:                        # generator.py:213-223: clips with the state marker
                save_video_batch(vid, bs, global_iter, os.path.join(self.opt.result_path, name + "_state"), self.opt.fps, True,
                                 self.opt.imagenet_norm, [-1, 1], self.opt.dataset, state=state)

    def run(self):
        with Engine(self.opt) as engine:
            self.engine = engine
            self.valid_data_info = self.get_data_info("valid", "img" if self.opt.gen_from_img else "vid")
            self.build_models(is_main=True)
            results = None
            for global_iter in range(self.opt.n_iter):
                data = self.next_batch(self.valid_data_info)
                if self.opt.gen_from_img:
                    data["vid"] = data.pop("img").unsqueeze(1)
                out = self.generate_vid(data, global_iter)
                if out["fake"] is not None:
                    packed = ops.pack_u8(out["fake"]["vid"].contiguous())
                    results = engine.all_gather_clips(packed)
            print('Generation was successfully finished.')
            return results


def save_video_batch(vid, bs, global_iter, path, fps, normalize, imagenet_norm, span, dataset, state=None, cat=None, idx=None,
                     is_layout=False):
    """helpers/generator.py:285-333.  The clamp / rescale / uint8 / channels-last pack runs on the
    GPU; files are written as mp4 when torchvision is importable, else as .npy uint8 [T,H,W,3]."""
    if is_layout:
        raise NotImplementedError("layout colour maps are not on the MI355X path (SURVEY 8f)")
    if normalize and imagenet_norm:
        u8 = ops.pack_u8_norm(vid, (0.229, 0.224, 0.225), (0.485, 0.456, 0.406))   # generator.py:303-305, same op order
    elif normalize:
        u8 = ops.pack_u8(vid.contiguous(), float(span[0]), float(span[1]))
    else:
        u8 = (vid.permute(0, 1, 3, 4, 2) * 255).to(torch.uint8)
    u8 = u8.cpu()
    if state is not None:  # generator.py:311-323: mark the (x, y) state on every frame
        res = {"bair": 64, "bairhd": 256}.get(dataset)
        if res is not None:
            st = state.detach().cpu()
            for i in range(u8.size(0)):
                for j in range(u8.size(1)):
                    x, y = st[i, j]
                    u8[i, j] = draw_cross(u8[i, j], min(int(res * x), res - 1), min(int(res * y), res - 1))
    os.makedirs(path, exist_ok=True)
    try:
        from torchvision.io import write_video
    except ImportError:
        write_video = None
    for i in range(u8.size(0)):
        suffix = '' if cat is None else f'_{cat[i]}'
        suffix += '' if idx is None else f'_{idx[i]}'
        stem = os.path.join(path, f"vid_{bs * global_iter + i:05d}{suffix}")
        if write_video is not None:
            write_video(stem + ".mp4", u8[i], fps)
        else:
            import numpy as np
            np.save(stem + ".npy", u8[i].numpy())
    return u8


def draw_cross(img, x, y):
    """helpers/generator.py:336-359: 3x3 marker, white plus on black corners, clipped at the border."""
    height, width = img.shape[:2]
    for dy in (-1, 0, 1):
        for dx in (-1, 0, 1):
            yy, xx = y + dy, x + dx
            if 0 <= yy < height and 0 <= xx < width:
                img[yy, xx] = 255 if (dx == 0 or dy == 0) else 0
    return img


def square_trajectory(init_state, vid_len):
    """helpers/generator.py:362-379: walk a square inside [0.2, 0.8)^2 with steps of 10/64."""
    state = init_state.repeat(1, vid_len, 1)
    step = 10 / 64
    moves = [(0, -step), (step, 0), (0, step), (-step, 0)]
    for i in range(state.size(0)):
        x, y = state[i, 0].clone()
        t = 0
        for j in range(1, vid_len):
            while not (0.2 <= x + moves[t][0] < 0.8 and 0.2 <= y + moves[t][1] < 0.8):
                t = (t + 1) % 4
            x, y = x + moves[t][0], y + moves[t][1]
            state[i, j, 0], state[i, j, 1] = x, y
    return {"state": state}


def blur(data, blur_sigma=10):
    """helpers/generator.py:381-390 (deblurring mode input)."""
    raise NotImplementedError("deblurring mode is not on the MI355X path yet (SURVEY 8f)")


if __name__ == "__main__":
    options = Options().parse(load_qvid_generator=True, load_transformer=True, load_state_estimator=True, load_stft_ae=True, save=False)
    Generator(options).run()
