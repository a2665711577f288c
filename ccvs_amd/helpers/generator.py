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
from ccvs_amd.helpers.pipeline import FrameFeed as _FrameFeed, PipelinedRun   # noqa: F401  (`_FrameFeed`: the name round 4's tests patch)


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
    def condition(self, data, draw=True):
        """First half of generate_vid (generator.py:57-130): encode every frame, tokenise the ancillary streams, crop to the
        conditioning window.  Returns the working set of one batch as a dict: `encoded`, `cropped`, `total_len`,
        `cond_len`, `crop_prop`.  `draw=False` (a warm-up pass whose results are discarded): nothing is drawn from the process
        generator -- missing class labels are zeros instead of the reference's randint (generator.py:124)."""
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
                n_lbl = encoded_data["code"].size(0)
                data["vid_lbl"] = torch.randint(low=0, high=len(opt.categories), size=[n_lbl]) if draw else torch.zeros(n_lbl, dtype=torch.int64)
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
        fake_data = yield from self.vid_model.decode_stream(dec_in, code_of, frames=opt.vid_len - (1 if opt.p2p else 0))
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
    def generate_vid(self, data, global_iter=0, save=False, schedule=None):
        """One batch through the synthesis path (reference helpers/generator.py:57-230).  `schedule`:
          "serial"  the reference's order -- encode, the whole token loop, then the decode;
          "stream"  the same work with the DECODER FOLLOWING THE TOKEN LOOP frame by frame (it needs the tokens of frame t only to decode
                    frame t, quantized_video_model.py:868-903): the one-batch case of `run_pipelined` (its token stream + one decode stream,
                    the call's host noise pre-drawn by the noise threads instead of inline) -- same clips, bit for bit, the call returns when
                    everything is enqueued on the caller's stream as before;
        None = `CCVS_GENERATE_VID_SCHEDULE` (default "stream": BAIR batch 16, one call, 2.02 s against 2.23 s = 119 against 107 frames/s,
        profiles/r06_single_call_probe.txt).  step_by_step / rec_only calls are serial whatever is asked."""
        opt = self.opt
        if schedule is None:
            schedule = os.environ.get("CCVS_GENERATE_VID_SCHEDULE", "stream")
        if schedule == "stream" and not (opt.step_by_step or opt.rec_only):
            from ccvs_amd.helpers.pipeline import PipelinedRun
            # (the records of the last run_pipelined -- its events, lanes, chains -- are the caller's to read afterwards: a single call
            #  in between must not replace them)
            keep = {k: getattr(self, k, None) for k in ("_pipeline_events", "last_cu_limit", "last_lanes", "last_chains", "last_dec_streams")}
            try:
                run = PipelinedRun(self, [data], first_iter=global_iter, lanes=1, chains=1,
                                   rec_pass=not opt.gen_from_img and getattr(opt, "rec_pass", True))
                out = run.run()[0]
                ev = self._pipeline_events[-1]
            finally:
                for k, v in keep.items():
                    if v is not None:
                        setattr(self, k, v)
            self._events_stream = ev
            out = {"real": out["real"], "fake": out["fake"], "rec": out["rec"], "enc_code": out["enc_code"], "real_state": out["real_state"]}
            if save:
                self.save_results(out, global_iter)
            return out
        self._events_stream = None
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

    def _host_noise_streams_ok(self, batch=None):
        """Host-drawn sampling noise (`--x_sample_noise host`: the reference's seeded torch.multinomial stream) of a batch can be
        pre-drawn as ONE stream of [B, V] blocks, one per new token, when the batch's tokens are one graph-replayed `fill_code`
        call over frame tokens only: no ancillary (state / STFT) picks with their own widths, no sliding token window (several
        calls), no beam search (n proposals per pick).  Then the token loops of several batches stack into one loop, each batch
        reading its own stream (`ccvs_gpt_decode.noise_stream`), and several loops run beside each other."""
        opt = self.opt
        size = int(self.qvid_opt.z_shape[0]) * int(self.qvid_opt.z_shape[1])
        if not (not (opt.state or opt.stft) and getattr(opt, "beam_size", None) is None and getattr(opt, "use_graph", True)
                and opt.vid_len * size <= opt.z_len):
            return False
        if batch is not None:
            # ... and when a batch's stream is of a size worth holding (pinned host memory + device memory per batch in flight):
            # BAIR 960 x 16 x 1024 floats = 63 MB; Kinetics 704 x 64 x 16384 = 3 GB (CCVS_NOISE_STREAM_MAX_MB, default 1024) -- beyond
            # it every pick draws its own block, one batch per loop, one loop at a time (the reference's order needs nothing else)
            steps = opt.vid_len * size - int(opt.cond_len) - (size if opt.p2p else 0)
            mb = 4.0 * steps * batch * int(getattr(opt, "z_num", 1024)) / 2 ** 20
            if mb > float(os.environ.get("CCVS_NOISE_STREAM_MAX_MB", "1024")):
                return False
        return True

    def _token_group_size(self, batch, lanes):
        """Batches whose token loops run as ONE loop (`lanes`, capped): the stacked rows must fit one decode step (256 rows)
        and be a few row blocks of the weight-stream GEMM at most (64 rows = one pass over the weights; more rows re-read them from
        L2: 128 by default).  Host-drawn sampling noise is one generator stream per BATCH in the reference's order: stacked when
        that stream can be pre-drawn (`_host_noise_streams_ok`), else one batch per loop."""
        opt = self.opt
        if (getattr(opt, "sample", False) and getattr(self.transformer_model, "sample_noise", "host") != "device"
                and not self._host_noise_streams_ok(batch)):
            return 1
        if getattr(opt, "beam_size", None) is not None:
            return 1
        max_rows = int(os.environ.get("CCVS_PIPELINE_MAX_ROWS", "128"))   # (64 until Kinetics measured 2 x 64 rows x 3 chains: 362 -> 373 frames/s; up to 256)
        return max(1, min(int(lanes), min(max_rows, 256) // max(batch, 1)))

    @staticmethod
    def _lanes_that_fit(lanes, chains, batch, height, width, total_bytes, frac=0.7, bytes_per_clip_pixel=10.0e3, taken=0, dec_streams=2,
                        noise_bytes_per_batch=0.0):
        """Lanes per token group such that the batches in flight -- up to lanes x (chains + 2): two groups in the token loops, one
        encoded behind them, one being decoded -- fit `frac` of the device memory minus what OTHER processes hold on the device
        (`taken`).  A batch in flight holds its context rings, skip features, token caches and the decoder's intermediates: ~10 KB
        per clip and pixel measured with two decode streams (BAIR: 174 GB at 16 batches of 16 x 256^2 in flight, weights and
        captured steps included); every decode stream beyond two keeps one more batch's decoder intermediates alive (~1/4 of a
        batch each).  BAIR at batch 16 keeps 4 lanes on 288 GB; batch 32 gets 2.
        `noise_bytes_per_batch`: the pre-drawn host noise stream of a batch (steps x B x V x 4 bytes, up to CCVS_NOISE_STREAM_MAX_MB),
        resident on the device (and pinned on the host) for every batch in flight -- 63 MB for BAIR, but 1 GB per batch just under the
        cap: 16 GB unaccounted at 16 batches in flight before round 6."""
        per_batch = float(batch) * height * width * bytes_per_clip_pixel + float(noise_bytes_per_batch)
        extra = 0.25 * max(0, dec_streams - 2) * per_batch
        while lanes > 1 and lanes * (chains + 2) * per_batch + extra > frac * total_bytes - taken:
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
    def run_pipelined(self, batches, first_iter=0, cu_limit=None, finish=None, lanes=None, chains=None, ramp=None, rec_pass=None, consume=None,
                      n_batches=None):
        """generate_vid over a sequence of batches with SEVERAL batches in flight on one GPU (`helpers/pipeline.py` says how: token
        groups x chains beside the encode stream and the decode streams, the decoder following the token loops frame by frame,
        host-drawn sampling noise pre-drawn in the reference's order).  Per batch the work is exactly generate_vid's -- encode,
        crop, token synthesis, decode (and, with `rec_pass`, the teacher-forced "rec" decode, generator.py:172-189) -- and so are
        the results: a row's arithmetic does not depend on the rows it shares a launch with and every batch keeps its own sampler
        state (tests/test_pipeline_gpu.py checks bit-equality with the serial schedule).

        batches: iterable of data dicts.  finish(i, out) -> anything: called on the decode stream of batch i when its clip is
        decoded, in batch order on every rank (pack / all-gather / save); its return values are collected.  `cu_limit` > 0 caps
        the decode streams to that many CUs while a token loop is in flight; `lanes` / `chains` / `ramp`: batches per token group,
        concurrent token loops, sizes of the first groups.  `rec_pass` None: off (the benchmark's metric counts synthesized
        frames only); `run()` passes the option.  Returns the list of per-batch results ({"fake", "rec", "real", "enc_code",
        "real_state", "finished", "index"}); with `consume(out)` every such dict is handed to the caller instead, on the batch's
        decode stream right behind `finish`, and only {"index", "finished"} stays in the list (a run of a thousand batches).
        `n_batches`: the length of `batches` when it is an iterator and the caller knows it (the warm-up then prepares only the
        token-group sizes that will occur)."""
        return PipelinedRun(self, batches, first_iter=first_iter, cu_limit=cu_limit, finish=finish, lanes=lanes, chains=chains, ramp=ramp,
                            rec_pass=bool(rec_pass), consume=consume, n_batches=n_batches).run()

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
        ev = getattr(self, "_events_stream", None)
        if ev is not None:      # schedule "stream": the stages overlap in time (HIP events on the stages' own streams)
            return {"encode": ev["e0"].elapsed_time(ev["e1"]), "transformer": ev["t0"].elapsed_time(ev["t1"]),
                    "decode": sum(a.elapsed_time(b) for a, b in ev["segs"])}
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

    def run(self, save=True):
        """The reference's entry point (helpers/generator.py:248-282): build the models, then `n_iter` batches through
        generate_vid, each saved as real / fake / rec clips (generator.py:191-223).  Here the batches go through the
        several-batches-in-flight schedule (`run_pipelined`; `CCVS_RUN_SCHEDULE=serial`, `--step_by_step`, `--rec_only` and a
        single batch: one generate_vid after the other) -- the same clips bit for bit (tests/test_run_gpu.py) -- the files are
        written by a writer thread behind the schedule, and the packed uint8 clips of all ranks are all-gathered (RCCL, side
        stream).  Returns the gathered clips of the last batch."""
        opt = self.opt
        with Engine(opt) as engine:
            self.engine = engine
            self.valid_data_info = self.get_data_info("valid", "img" if opt.gen_from_img else "vid")
            self.build_models(is_main=True)
            rec = not opt.gen_from_img and (getattr(opt, "rec_pass", True) or opt.rec_only)
            writer = _ResultWriter(self) if save else None

            def batches():
                for _ in range(opt.n_iter):
                    data = self.next_batch(self.valid_data_info)
                    if opt.gen_from_img:
                        data["vid"] = data.pop("img").unsqueeze(1)
                    yield data

            def gather(_, fake):
                return engine.all_gather_clips_async(ops.pack_u8(fake["vid"].contiguous()))

            pipelined = (opt.n_iter > 1 and not opt.step_by_step and not opt.rec_only
                         and os.environ.get("CCVS_RUN_SCHEDULE", "pipelined") != "serial")
            last = None
            try:
                if pipelined:
                    res = self.run_pipelined(batches(), finish=gather, rec_pass=rec, consume=writer.put if writer is not None else (lambda out: None),
                                             n_batches=opt.n_iter)
                    last = res[-1]["finished"] if res else None
                else:
                    for global_iter, data in enumerate(batches()):
                        out = self.generate_vid(data, global_iter)
                        out["index"] = global_iter
                        if out["fake"] is not None:
                            last = gather(global_iter, out["fake"])
                        if writer is not None:
                            writer.put(out)
            finally:
                if writer is not None:
                    writer.close()
            print('Generation was successfully finished.')
            return last.wait() if last is not None else None


class _ResultWriter:
    """The files of `Generator.run` (generator.py:191-223 via `save_results`), written by one thread behind the schedule: `put` is
    called on the stream that produced the batch and records an event there; the thread waits for it on the host, packs and
    copies the clips out and writes them, at most `depth` batches behind (a slower disk throttles the run, not the memory)."""

    def __init__(self, gen, depth=4):
        import queue
        import threading
        self.gen, self.error = gen, None
        self.dev = torch.cuda.current_device()
        self.queue = queue.Queue(maxsize=depth)
        self.thread = threading.Thread(target=self._serve, name="ccvs-result-writer", daemon=True)
        self.thread.start()

    def put(self, out):
        if self.error is not None:
            raise self.error
        ready = torch.cuda.Event()
        ready.record()
        self.queue.put((out, ready))

    def close(self):
        self.queue.put(None)
        self.thread.join()
        if self.error is not None:
            raise self.error

    def _serve(self):
        torch.cuda.set_device(self.dev)
        while True:
            item = self.queue.get()
            if item is None:
                return
            if self.error is not None:
                continue
            out, ready = item
            try:
                ready.synchronize()
                with torch.no_grad():
                    self.gen.save_results(out, out["index"])
            except BaseException as exc:
                self.error = exc


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
