"""GPT over latent tokens on the MI355X kernel library.

Host-side mirror of the reference's `models/skip_vid_generator/models/mingpt.py` (class `GPT`
and its `Block` / `CausalSelfAttention` parameter tree, lines 33-305): same constructor
signature, same sub-module registration order (so `self.apply(_init_weights)` consumes the RNG
identically) and the same `state_dict()` keys, minus the 1024x1024 `attn.mask` buffers.

What differs is the execution model.  The reference re-runs all layers over the whole prefix
for every sampled token (transformer_model.py:343-350).  Here the network is an incremental
engine over a KV cache: `begin` -> `prefill` -> `step` ... ; `forward` (teacher-forced logits
for a whole sequence, mingpt.py:232-305) is the same engine run as one prefill.  Every layer is
a libccvs_hip.so call: fused QKV GEMM, cache append, cached causal attention, projection GEMM
with the residual add in its epilogue, LayerNorm, MLP GEMMs with GELU / residual epilogues.
"""
import torch
import torch.nn as nn

from ccvs_amd import ops


class GPTConfig:
    """mingpt.py:9-18."""
    embd_pdrop = 0.1
    resid_pdrop = 0.1
    attn_pdrop = 0.1

    def __init__(self, block_size, **kwargs):
        self.block_size = block_size
        for k, v in kwargs.items():
            setattr(self, k, v)


class CausalSelfAttention(nn.Module):
    """Parameter holder (mingpt.py:33-61).  The causal mask is implied by the cache positions."""

    def __init__(self, config):
        super().__init__()
        assert config.n_embd % config.n_head == 0
        self.key = nn.Linear(config.n_embd, config.n_embd)
        self.query = nn.Linear(config.n_embd, config.n_embd)
        self.value = nn.Linear(config.n_embd, config.n_embd)
        self.attn_drop = nn.Dropout(config.attn_pdrop)
        self.resid_drop = nn.Dropout(config.resid_pdrop)
        self.proj = nn.Linear(config.n_embd, config.n_embd)
        self.n_head = config.n_head
        self._qkv = None

    def qkv_packed(self):
        """[query; key; value] stacked into one [3C, C] weight + [3C] bias for a single GEMM."""
        ws = (self.query.weight, self.key.weight, self.value.weight)
        key = tuple((w.data_ptr(), w._version) for w in ws) + (ws[0].device,)
        if self._qkv is None or self._qkv[0] != key:
            w = torch.cat([w.detach() for w in ws], dim=0).contiguous()
            b = torch.cat([self.query.bias.detach(), self.key.bias.detach(), self.value.bias.detach()]).contiguous()
            self._qkv = (key, w, b)
        return self._qkv[1], self._qkv[2]


class NoiseInjection(nn.Module):
    """mingpt.py:84-97 with use_noise=False (identity); kept so `mlp.{0,3}` keep their indices."""

    def __init__(self, use_noise):
        super().__init__()
        if use_noise:
            raise NotImplementedError("resid_noise is a training-time option")

    def forward(self, x):
        return x


class Block(nn.Module):
    """mingpt.py:99-117."""

    def __init__(self, config):
        super().__init__()
        self.ln1 = nn.LayerNorm(config.n_embd)
        self.ln2 = nn.LayerNorm(config.n_embd)
        self.attn = CausalSelfAttention(config)
        self.mlp = nn.Sequential(
            nn.Linear(config.n_embd, 4 * config.n_embd),
            NoiseInjection(config.resid_noise),
            nn.GELU(),
            nn.Linear(4 * config.n_embd, config.n_embd),
            nn.Dropout(config.resid_pdrop),
        )
        self._folded = None

    def folded(self):
        """(ln1 folded into the stacked QKV weight, ln2 folded into mlp[0]) -- `ops.pack_ln_linear`."""
        src = (self.ln1.weight, self.ln1.bias, self.attn.query.weight, self.attn.key.weight, self.attn.value.weight,
               self.attn.query.bias, self.attn.key.bias, self.attn.value.bias, self.ln2.weight, self.ln2.bias,
               self.mlp[0].weight, self.mlp[0].bias)
        key = tuple((t.data_ptr(), t._version) for t in src) + (src[0].device,)
        if self._folded is None or self._folded[0] != key:
            wqkv, bqkv = self.attn.qkv_packed()
            qkv = ops.pack_ln_linear(wqkv, bqkv, self.ln1.weight, self.ln1.bias)
            fc = ops.pack_ln_linear(self.mlp[0].weight, self.mlp[0].bias, self.ln2.weight, self.ln2.bias)
            self._folded = (key, qkv, fc)
        return self._folded[1], self._folded[2]


class GPT(nn.Module):
    """mingpt.py:120-305."""

    def __init__(self, vocab_size, block_size, num_blocks, n_layer=12, n_head=8, n_embd=256, embd_pdrop=0., resid_pdrop=0.,
                 attn_pdrop=0., n_unmasked=0, resid_noise=False, emb_mode=None, shape=None, state_vocab_size=0, state_size=0,
                 use_start_token=False, num_lbl=0, use_lbl=False, state_front=False):
        super().__init__()
        if n_unmasked:
            raise NotImplementedError("unmasked prefixes are not on the MI355X path (SURVEY 8f)")
        config = GPTConfig(block_size=block_size, vocab_size=vocab_size, embd_pdrop=embd_pdrop, resid_pdrop=resid_pdrop,
                           attn_pdrop=attn_pdrop, n_layer=n_layer, n_head=n_head, n_embd=n_embd, n_unmasked=n_unmasked,
                           resid_noise=resid_noise, shape=shape, emb_mode=emb_mode, state_vocab_size=state_vocab_size,
                           state_size=state_size, use_start_token=use_start_token, num_blocks=num_blocks, num_lbl=num_lbl,
                           use_lbl=use_lbl, state_front=state_front)
        self.tok_emb = nn.Embedding(config.vocab_size, config.n_embd)
        if config.state_vocab_size > 0:  # ancillary (state / STFT) token stream, mingpt.py:134-135
            self.state_tok_emb = nn.Embedding(config.state_vocab_size, config.n_embd)
        if config.use_start_token:   # mingpt.py:136-137 (consumes the RNG stream right here)
            self.start_tok_emb = nn.Parameter(torch.randn(1, config.n_embd))
        if config.use_lbl:           # mingpt.py:140-141
            self.lbl_emb = nn.Embedding(config.num_lbl, config.n_embd)
        height, width = config.shape
        if config.emb_mode is not None:
            if config.emb_mode == "spatio-temporal":
                self.h_emb = nn.Parameter(torch.zeros(1, height, config.n_embd))
                self.w_emb = nn.Parameter(torch.zeros(1, width, config.n_embd))
                self.t_emb = nn.Parameter(torch.zeros(1, config.num_blocks, config.n_embd))
            elif config.emb_mode == "temporal":
                self.s_emb = nn.Parameter(torch.zeros(1, height * width, config.n_embd))
                self.t_emb = nn.Parameter(torch.zeros(1, config.num_blocks, config.n_embd))
            else:
                raise ValueError
        else:
            self.pos_emb = nn.Parameter(torch.zeros(1, config.num_blocks * height * width, config.n_embd))
        if config.state_size > 0:  # mingpt.py:158-162
            if config.emb_mode is not None:
                self.state_s_emb = nn.Parameter(torch.zeros(1, config.state_size, config.n_embd))
            else:
                self.state_pos_emb = nn.Parameter(torch.zeros(1, config.num_blocks * config.state_size, config.n_embd))
        self.drop = nn.Dropout(config.embd_pdrop)
        self.blocks = nn.Sequential(*[Block(config) for _ in range(config.n_layer)])
        self.ln_f = nn.LayerNorm(config.n_embd)
        self.head = nn.Linear(config.n_embd, max(config.vocab_size, config.state_vocab_size), bias=False)
        self.block_size = config.block_size + (1 if config.use_start_token else 0) + (1 if config.use_lbl else 0)   # mingpt.py:171
        self.apply(self._init_weights)
        self.config = config
        self._cache = None      # the engine state in use: one of `_caches`
        self._caches = {}       # per batch size (a group of several generation batches is one more batch size)
        # in-kernel sampling noise (Philox): key words, global index of batch row 0 and a per-call counter.  `noise_key`
        # None draws a fresh key from torch's generator per call (honours torch.manual_seed); the Generator sets a key
        # derived from (seed, iteration) plus the rank's first clip, so sampled tokens do not depend on the world size.
        # ROW GROUPS: when the batch is the rows of several generation batches stacked (`Generator.run_pipelined`: one
        # token loop, i.e. one pass over the weights per token, for all of them), `noise_key` and `row_offset` are lists with
        # one entry per group of `batch / len(list)` consecutive rows: every batch keeps the words it has when run alone.
        self.noise_key = None
        self.row_offset = 0
        self.noise_call = 0
        # warm_only: the decode steps of a call are CAPTURED (KV cache allocated, descriptor built, hipGraphs recorded) but not
        # replayed -- `Generator.run_pipelined` prepares every (token chain, group size) this way before its worker threads start
        self.warm_only = False
        # progress(n, codes): called on the stream of a graph-replayed generate() whenever columns [0, n) of `codes` (the cache's
        # token buffer) are final -- `Generator.run_pipelined` hands the finished frames to the decoder while the loop goes on
        self.progress = None
        # host-drawn sampling noise of the NEXT generate() call, pre-drawn: one device tensor [add_len, rows of the group, V] per row
        # group, block i = the [rows, V] Exp(1) draw of pick i in the order of the reference's generator (`Generator.run_pipelined`
        # draws them on a noise thread, batch after batch).  None: generate() draws its own through `host_noise` (one group only).
        self.noise_streams = None
        # persistent_step: the decode step as ONE launch of resident workgroups with in-launch grid barriers (gpt.hip:
        # gpt_step_kernel; `ccvs_gpt_decode.persistent`) instead of 5 n_layer + 3 dependent launches -- bit-identical tokens
        # (tests/test_persistent_step_gpu.py).  A step then holds its CU slots for its whole duration: for schedules with ONE
        # token loop in flight (`Generator.run_pipelined` with chains = 1).  CCVS_DECODE_PERSISTENT=1 turns it on.
        import os
        self.persistent_step = os.environ.get("CCVS_DECODE_PERSISTENT", "0") == "1"

    @property
    def _graphs(self):
        return self._cache["graphs"]

    def n_groups(self):
        """Row groups of the next call (1 unless the Generator stacked several batches)."""
        return len(self.noise_key) if isinstance(self.noise_key, list) else 1

    def drop_engine_state(self):
        """Free every KV cache / captured decode step (they are rebuilt on the next call)."""
        self._cache, self._caches = None, {}

    def _philox_words(self):
        """[(key0, key1, row0, call)] of this generate() call, one tuple per row group."""
        if self.noise_key is None:
            k = torch.randint(0, 2**32, (2,), dtype=torch.int64).tolist()
            return [(int(k[0]), int(k[1]), int(self.row_offset), 0)]
        call = self.noise_call
        self.noise_call += 1
        if isinstance(self.noise_key, list):
            assert isinstance(self.row_offset, list) and len(self.row_offset) == len(self.noise_key)
            return [(int(k[0]) & 0xffffffff, int(k[1]) & 0xffffffff, int(r), call) for k, r in zip(self.noise_key, self.row_offset)]
        return [(int(self.noise_key[0]) & 0xffffffff, int(self.noise_key[1]) & 0xffffffff, int(self.row_offset), call)]

    def _set_decode_state(self, words):
        """Device-resident `state` of ccvs_gpt_decode ([groups][8]): counters zeroed, Philox words in place (one upload)."""
        c = self._cache
        st = torch.zeros(c["G"], 8, dtype=torch.int64)
        if words is not None:
            assert len(words) == c["G"], (len(words), c["G"])
            for g, (k0, k1, row0, call) in enumerate(words):
                st[g, 3], st[g, 4], st[g, 5], st[g, 6] = call, k0, k1, row0
        st = torch.where(st >= 2**31, st - 2**32, st).to(torch.int32)
        c["state"].copy_(st, non_blocking=True)

    def _pick(self, logits, sampler, noise, out, words, step):
        """One eager pick for all rows: `ops.sample_topk` per row group when the noise is drawn in the kernel (each group has
        its own Philox words `words[g]`; `step` = (step word, call-word mask))."""
        if words is None:
            return ops.sample_topk(logits, sampler["top_k"], sampler["temperature"], noise=noise, out=out)
        b = logits.shape[0]
        n = len(words)
        assert b % n == 0
        per = b // n
        if out is None:
            out = torch.empty(b, dtype=torch.int64, device=logits.device)
        for g, (k0, k1, row0, call) in enumerate(words):
            ops.sample_topk(logits[g * per:(g + 1) * per], sampler["top_k"], sampler["temperature"], out=out[g * per:(g + 1) * per],
                            philox=(k0, k1, row0, step[0], call | step[1]))
        return out

    def get_block_size(self):
        return self.block_size

    def _init_weights(self, module):
        """mingpt.py:177-184."""
        if isinstance(module, (nn.Linear, nn.Embedding)):
            module.weight.data.normal_(mean=0.0, std=0.02)
            if isinstance(module, nn.Linear) and module.bias is not None:
                module.bias.data.zero_()
        elif isinstance(module, nn.LayerNorm):
            module.bias.data.zero_()
            module.weight.data.fill_(1.0)

    # ------------------------------------------------------------------ positional tables
    @torch.no_grad()
    def get_pos_emb(self, t, delta_length=None):
        """[n, t, C] positional embeddings (mingpt.py:186-217); n = len(delta_length) or 1."""
        if t == 0:
            return 0
        cfg = self.config
        height, width = cfg.shape
        size = height * width
        dev = self.tok_emb.weight.device
        if delta_length is None or 0 in delta_length.size():
            delta_length = torch.zeros(1, dtype=torch.long)
        deltas = [int(d) for d in delta_length.view(-1).tolist()]
        n = len(deltas)
        if cfg.emb_mode is None:
            return torch.stack([self.pos_emb[0, d * size: d * size + t] for d in deltas]).to(dev)
        length = t // size + (1 if t % size != 0 else 0)
        t_emb = torch.stack([self.t_emb[0, d: d + length] for d in deltas])  # [n, length, C]
        if cfg.emb_mode == "temporal":
            pos = self.s_emb.view(1, 1, size, -1) + t_emb.view(n, length, 1, -1)
        else:
            pos = self.h_emb.view(1, 1, height, 1, -1) + self.w_emb.view(1, 1, 1, width, -1)
            pos = pos + t_emb.view(n, length, 1, 1, -1)
        return pos.reshape(n, length * size, -1)[:, :t]

    @torch.no_grad()
    def get_state_pos_emb(self, t):
        """[1, t, C] positional embeddings of the ancillary stream (mingpt.py:219-230)."""
        cfg = self.config
        size = cfg.state_size
        if cfg.emb_mode is not None:
            length = t // size + (1 if t % size != 0 else 0)
            pos = self.state_s_emb.view(1, 1, size, -1) + self.t_emb[:, :length].view(1, length, 1, -1)
            return pos.reshape(1, length * size, -1)[:, :t]
        return self.state_pos_emb[:, :t]

    # ------------------------------------------------------------------ ancillary stream layout
    def stream_kinds(self, n_code, n_state):
        """The merged sequence of `GPT.forward` (mingpt.py:246-282) for n_code frame tokens and n_state ancillary tokens:
        a list of (kind, index) with kind 0 = frame token `index`, 1 = ancillary token `index`.  Per frame the
        `state_size` ancillary tokens precede the frame's h*w tokens (`state_front`: all ancillary tokens first)."""
        cfg = self.config
        size, ss = cfg.shape[0] * cfg.shape[1], cfg.state_size
        n_state = min(n_state, cfg.num_blocks * ss)  # mingpt.py:249
        if n_state == 0:
            return [(0, i) for i in range(n_code)]
        if cfg.state_front:
            return [(1, i) for i in range(n_state)] + [(0, i) for i in range(n_code)]
        length = n_code // size
        if length == 0:
            # reference quirk (mingpt.py:281-282): with less than one whole frame the frame tokens are DROPPED
            return [(1, i) for i in range(min(ss, n_state))]
        assert n_state >= length * ss, "ancillary stream shorter than the frames it annotates (the reference's view() fails too)"
        seq = []
        for f in range(length):
            seq += [(1, f * ss + i) for i in range(ss)] + [(0, f * size + i) for i in range(size)]
        seq += [(1, i) for i in range(length * ss, min((length + 1) * ss, n_state))]
        seq += [(0, i) for i in range(length * size, n_code)]
        return seq

    def _token_table(self):
        """Embedding rows addressed by the engine: [tok_emb ; state_tok_emb] (an ancillary token v is row vocab_size + v)."""
        if self.config.state_vocab_size <= 0:
            return self.tok_emb.weight
        src = (self.tok_emb.weight, self.state_tok_emb.weight)
        key = tuple((t.data_ptr(), t._version) for t in src) + (src[0].device,)
        tt = getattr(self, "_tok_table", None)
        if tt is None or tt[0] != key:
            tt = self._tok_table = (key, torch.cat([t.detach() for t in src], dim=0).contiguous())
        return tt[1]

    def _stream_rows(self, code, state_code):
        """[B, T] embedding-table rows of the merged sequence of (code, state_code)."""
        kinds = self.stream_kinds(code.shape[1], 0 if state_code is None else state_code.shape[1])
        dev = code.device
        kind = torch.tensor([k for k, _ in kinds], dtype=torch.bool, device=dev)
        idx = torch.tensor([i for _, i in kinds], dtype=torch.long, device=dev)
        frame_rows = code[:, idx.clamp(max=max(code.shape[1] - 1, 0))] if code.shape[1] else torch.zeros(code.shape[0], len(kinds), dtype=torch.long, device=dev)
        if state_code is None or state_code.shape[1] == 0:
            return frame_rows.contiguous()
        state_rows = state_code[:, idx.clamp(max=state_code.shape[1] - 1)] + self.config.vocab_size
        return torch.where(kind.view(1, -1), state_rows, frame_rows).contiguous()

    def _stream_pos_table(self, n_rows, n_state_front=0):
        """[n_rows, C] positional rows in merged-sequence order."""
        cfg = self.config
        size, ss = cfg.shape[0] * cfg.shape[1], cfg.state_size
        if cfg.state_front:
            parts = [self.get_state_pos_emb(n_state_front)[0]] if n_state_front else []
            if n_rows - n_state_front > 0:
                parts.append(self.get_pos_emb(n_rows - n_state_front)[0])
            return torch.cat(parts, dim=0)
        tot = size + ss
        nf = (n_rows + tot - 1) // tot
        assert nf <= cfg.num_blocks, "Cannot forward, model block size is exhausted."
        fp = self.get_pos_emb(nf * size)[0].view(nf, size, -1)
        sp = self.get_state_pos_emb(nf * ss)[0].view(nf, ss, -1)
        return torch.cat([sp, fp], dim=1).reshape(nf * tot, -1)[:n_rows]

    # ------------------------------------------------------------------ incremental engine
    @torch.no_grad()
    def begin(self, batch, max_len, stream=False, n_state_front=0):
        """Allocate (or reuse) the KV cache and the device-resident decode state for `batch` sequences of
        at most `max_len` positions.  `stream`: positions are those of the merged frame / ancillary sequence."""
        cfg = self.config
        if not stream:
            assert max_len <= self.block_size, "Cannot forward, model block size is exhausted."  # mingpt.py:299
        dev = self.tok_emb.weight.device
        d = cfg.n_embd // cfg.n_head
        groups = self.n_groups()
        assert batch % groups == 0, f"{batch} rows do not split into {groups} row groups"
        c = self._caches.get(batch)
        if c is None or c["T"] < max_len or c["dev"] != dev or c["G"] != groups:
            # (captured graphs live in the cache dict: they go with the buffers they hold)
            if c is None and len(self._caches) >= 8:
                self._caches.pop(next(iter(self._caches)))   # a handful of batch sizes at most: drop the oldest
            C, V = cfg.n_embd, self.head.weight.shape[0]
            f32 = dict(dtype=torch.float32, device=dev)
            kc = [torch.empty(batch, cfg.n_head, max_len, d, **f32) for _ in range(cfg.n_layer)]
            vc = [torch.empty(batch, cfg.n_head, max_len, d, **f32) for _ in range(cfg.n_layer)]
            c = self._caches[batch] = {
                "B": batch, "G": groups, "T": max_len, "dev": dev, "k": kc, "v": vc,
                "len_dev": torch.zeros(groups, dtype=torch.int32, device=dev),  # cache positions filled, per row group (they advance together)
                "widx": torch.zeros(groups, dtype=torch.int32, device=dev),     # column of `codes` the next token goes to
                "tok": torch.zeros(batch, 1, dtype=torch.int64, device=dev),   # last sampled token
                "codes": torch.zeros(batch, max_len, dtype=torch.int64, device=dev),
                # scratch of ccvs_gpt_decode_step
                "x": torch.empty(batch, C, **f32), "q": torch.empty(batch, C, **f32), "att": torch.empty(batch, C, **f32),
                "h": torch.empty(batch, 4 * C, **f32), "logits": torch.empty(batch, V, **f32), "noise": torch.empty(batch, V, **f32),
                "state": torch.zeros(groups, 8, dtype=torch.int32, device=dev),    # per group: [0] steps done, [4..5] Philox key (ccvs_hip.h)
                "noise_ptrs": torch.zeros(groups, dtype=torch.int64, device=dev),  # per group: its pre-drawn noise stream (ccvs_gpt_decode.noise_stream)
                "noise_keep": None,
                "desc": None, "graphs": {},
            }
        self._cache = c
        c["len"] = 0
        c["stream"] = bool(stream)
        if stream:
            tot = cfg.shape[0] * cfg.shape[1] + cfg.state_size
            table = self._stream_pos_table(c["T"] if cfg.state_front else min(c["T"], cfg.num_blocks * tot), n_state_front)
        else:
            table = self.get_pos_emb(min(c["T"], cfg.block_size))[0]   # frame-token positions (label / start tokens carry none)
        if "pos_table" in c and c["pos_table"].shape == table.shape:
            c["pos_table"].copy_(table)   # same storage: captured graphs keep pointing at it
        else:
            c["pos_table"] = table.contiguous().clone()
            c["desc"], c["graphs"] = None, {}
        c["frame_pos0"] = 0
        return c

    def _layers(self, x, b, tq, pos_dev=None):
        """x [b*tq, C] -> x after all blocks; appends tq positions to the cache (at the host-side
        length, or at the device-resident `pos_dev` when given)."""
        c = self._cache
        pos0 = 0 if pos_dev is not None else c["len"]
        C = self.config.n_embd
        # whole-sequence calls always run the row-blocked GEMM form, single-position calls the weight-stream form: a row's
        # bits must not depend on how many rows share the launch (include/ccvs_hip.h, CCVS_GEMM_SEQ)
        seq = ops.GEMM_SEQ if tq > 1 else 0
        for i, blk in enumerate(self.blocks):
            # 5 launches per layer: [ln1 + QKV + cache scatter] [attention] [proj + residual]
            #                       [ln2 + fc + GELU] [fc2 + residual]
            qkv_w, fc_w = blk.folded()
            q = ops.gemm_ln_qkv(x, *qkv_w, c["k"][i], c["v"][i], b, tq, pos0, pos_dev, eps=blk.ln1.eps).view(b, tq, C)
            att = ops.attention(q, c["k"][i], c["v"][i], pos0, pos_dev)
            ops.gemm_nt(att.view(b * tq, C), blk.attn.proj.weight, blk.attn.proj.bias, ops.EPI_RESIDUAL | seq, residual=x, out=x)
            h = ops.gemm_ln(x, *fc_w, eps=blk.ln2.eps, epilogue=ops.EPI_GELU | seq)
            ops.gemm_nt(h, blk.mlp[3].weight, blk.mlp[3].bias, ops.EPI_RESIDUAL | seq, residual=x, out=x)
        if pos_dev is None:
            c["len"] = pos0 + tq
        return x

    def _head_packed(self):
        """ln_f folded into the (bias-free) head projection: (key, (W*gamma, W beta, rowsum))."""
        src = (self.ln_f.weight, self.ln_f.bias, self.head.weight)
        key = tuple((t.data_ptr(), t._version) for t in src) + (src[0].device,)
        hf = getattr(self, "_head_folded", None)
        if hf is None or hf[0] != key:
            hf = self._head_folded = (key, ops.pack_ln_linear(self.head.weight, None, self.ln_f.weight, self.ln_f.bias))
        return hf

    def _head(self, x):
        return ops.gemm_ln(x, *self._head_packed()[1], eps=self.ln_f.eps)

    def n_prefix(self):
        """Positions the label / start tokens occupy in front of every sequence (mingpt.py:289-297)."""
        return (1 if self.config.use_lbl else 0) + (1 if self.config.use_start_token else 0)

    def _prefix_emb(self, b, lbl_idx):
        """[B, n_prefix, C]: [label embedding | start embedding] -- no positional part (mingpt.py:289-297)."""
        rows = []
        if self.config.use_lbl:
            assert lbl_idx is not None and 0 not in lbl_idx.size(), "label tokens (--x_cat) need data['vid_lbl']"
            rows.append(self.lbl_emb.weight.detach()[lbl_idx.to(self.lbl_emb.weight.device).view(-1)].view(b, 1, -1))
        if self.config.use_start_token:
            rows.append(self.start_tok_emb.detach().view(1, 1, -1).expand(b, 1, -1))
        return torch.cat(rows, dim=1) if rows else None

    @torch.no_grad()
    def prefill(self, idx, cond_idx=None, delta_length_cond=None, all_logits=False, lbl_idx=None):
        """Run [label | start | cond prefix | idx] through the network, filling the cache from position 0.  `idx` holds rows of
        `_token_table()` (= frame token ids; merged-sequence rows for a stream cache, see `_stream_rows`).
        Returns logits of the last position [B,V], or -- like the reference's `head(x)[:, t_cond:]` -- of every position
        from t_cond on [B, n_prefix + T, V]."""
        c = self._cache
        b, t = idx.shape
        C = self.config.n_embd
        use_cond = cond_idx is not None and 0 not in cond_idx.size()
        parts = []
        prefix = self._prefix_emb(b, lbl_idx)
        n_pre = 0 if prefix is None else prefix.shape[1]
        if prefix is not None:
            parts.append(prefix.float())
        if use_cond:
            t_cond = cond_idx.shape[1]
            cond_tab = self.get_pos_emb(t_cond, delta_length_cond)               # [n, t_cond, C]
            if cond_tab.shape[0] == 1 and b > 1:
                cond_tab = cond_tab.expand(b, -1, -1)
            table = torch.cat([c["pos_table"], cond_tab.reshape(-1, C)], dim=0).contiguous()
            off = (c["pos_table"].shape[0] + torch.arange(b, dtype=torch.int32) * t_cond).to(idx.device)
            parts.append(ops.gpt_embed(cond_idx.contiguous(), self._token_table(), table, 0, off).view(b, t_cond, C))
        else:
            t_cond = 0
        if t > 0:   # (t == 0: unconditional generation, scripts/bairhd/save_videos_unc.sh -- the start token alone opens the sequence)
            parts.append(ops.gpt_embed(idx.contiguous(), self._token_table(), c["pos_table"], 0).view(b, t, C))
        assert parts, "nothing to run: no frame token, no conditioning prefix, no start / label token"
        x = parts[0] if len(parts) == 1 else torch.cat(parts, dim=1)
        tq = n_pre + t_cond + t
        assert c["len"] == 0 and tq <= c["T"], "Cannot forward, model block size is exhausted."
        x = self._layers(x.reshape(b * tq, C).contiguous(), b, tq)
        c["frame_pos0"] = n_pre + t_cond  # cache position of frame token 0
        if all_logits:
            xs = x.view(b, tq, C)[:, t_cond:].reshape(b * (tq - t_cond), C)
            return self._head(xs).view(b, tq - t_cond, -1)
        return self._head(x.view(b, tq, C)[:, -1].contiguous())

    @torch.no_grad()
    def step(self, tok):
        """tok int64 [B,1] (any row stride): the token at the next frame position. Returns [B,V]."""
        c = self._cache
        b = tok.shape[0]
        frame_pos = c["len"] - c["frame_pos0"]
        assert c["len"] < c["T"], "Cannot forward, model block size is exhausted."
        x = ops.gpt_embed(tok, self._token_table(), c["pos_table"], frame_pos)
        x = self._layers(x, b, 1)
        return self._head(x)

    @torch.no_grad()
    def reorder_cache(self, rows):
        """Cache row r continues the sequence of old row rows[r] (beam-search pruning): gather of the filled part of every
        layer's keys and values."""
        c = self._cache
        n = c["len"]
        for cache in c["k"] + c["v"]:
            cache[:, :, :n] = cache[rows, :, :n]

    @torch.no_grad()
    def extend(self, rows):
        """Append tq >= 1 more positions (rows of `_token_table()`, [B,tq]) to the cache; logits of the last one [B,V]."""
        c = self._cache
        b, tq = rows.shape
        C = self.config.n_embd
        assert c["len"] + tq <= c["T"], "Cannot forward, model block size is exhausted."
        x = ops.gpt_embed(rows.contiguous(), self._token_table(), c["pos_table"], c["len"] - c["frame_pos0"])
        x = self._layers(x.view(b * tq, C), b, tq)
        return self._head(x.view(b, tq, C)[:, -1].contiguous())

    # -- sampled generation: prefill + (add_len - 1) decode steps, all state on the device ----------
    def _decode_desc(self, sampler):
        """The `ccvs_gpt_decode` descriptor of the current cache (rebuilt when weights or sampler change)."""
        c = self._cache
        cfg = self.config
        folded = [blk.folded() for blk in self.blocks]
        head_key, (hw, hb, hs) = self._head_packed()
        device_rng = sampler["sample"] and sampler["noise"] == "device"
        host_stream = bool(sampler["sample"] and not device_rng and sampler.get("stream"))   # host noise read from a pre-drawn stream
        key = (tuple(blk._folded[0] for blk in self.blocks), head_key, sampler["sample"], sampler["top_k"],
               sampler["temperature"], device_rng, host_stream, c["frame_pos0"], bool(self.persistent_step))
        if c["desc"] is None or c["desc"][0] != key:
            layers = []
            for i, blk in enumerate(self.blocks):
                (qw, qb, qs), (fw, fb, fs) = folded[i]
                layers.append(dict(qkv_w=qw, qkv_b=qb, qkv_s=qs, proj_w=blk.attn.proj.weight, proj_b=blk.attn.proj.bias,
                                   fc_w=fw, fc_b=fb, fc_s=fs, fc2_w=blk.mlp[3].weight, fc2_b=blk.mlp[3].bias,
                                   kcache=c["k"][i], vcache=c["v"][i]))
            desc = ops.GptDecodeStep(
                layers, B=c["B"], groups=c["G"], C=cfg.n_embd, H=cfg.n_head, Tmax=c["T"], ln_eps=self.ln_f.eps,
                tok_emb=self._token_table(), pos_table=c["pos_table"], pos_off=-c["frame_pos0"], head=(hw, hb, hs),
                tok=c["tok"], codes=c["codes"], widx=c["widx"], length=c["len_dev"],
                x=c["x"], q=c["q"], att=c["att"], h=c["h"], logits=c["logits"],
                noise=c["noise"] if (sampler["sample"] and not device_rng and not host_stream) else None, rng=device_rng,
                noise_stream=c["noise_ptrs"] if host_stream else None,
                top_k=sampler["top_k"], temperature=sampler["temperature"], state=c["state"], persistent=self.persistent_step)
            c["desc"] = (key, desc)
            c["graphs"] = {}   # captured graphs replay the OLD descriptor's pointers (packed weights are freed with it)
        return c["desc"][1]

    def check_steps(self):
        """Persistent decode steps: wait for the current stream and raise if a grid barrier of a step enqueued so far gave up (its
        tokens would be garbage).  No-op for the launch chain.  generate() calls it in front of every sequence -- i.e. behind the
        previous one on this stream --, callers that read the last sequence's tokens call it themselves (tests, bench.py)."""
        for c in self._caches.values():
            if c.get("desc") is not None and c["desc"][1].persistent:
                c["desc"][1].status()
                return      # one workspace per stream: one check covers every cache of this engine

    def _emit(self, logits, sampler, noise, col, words=None, step=None):
        """Pick the next token from `logits` into c['tok'] and store it in column `col` of c['codes'] (`words`, `step`: in-kernel
        noise, see `_pick`)."""
        c = self._cache
        self._pick(logits, sampler, noise, c["tok"].view(-1), words, step)
        c["codes"][:, col] = c["tok"][:, 0]
        c["widx"].fill_(col + 1)

    def _decode_body(self, sampler, noise=None, trace=None):
        """One decode step driven entirely by device-resident state, hence hipGraph-capturable: embed
        c['tok'] at frame position len - frame_pos0, run the layers against the cache, pick and store the
        next token, advance the counters -- one `ccvs_gpt_decode_step` call (include/ccvs_hip.h)."""
        c = self._cache
        desc = self._decode_desc(sampler)
        if sampler["sample"] and sampler["noise"] != "device" and not sampler.get("stream"):
            c["noise"].copy_(noise, non_blocking=True)   # host-drawn Exp(1) noise (reference-reproducible stream), one block per eager step
        desc.launch()
        if trace is not None:
            trace.append(c["logits"].clone())

    GRAPH_STEPS = 8   # decode steps per replay of the long form of the captured graph

    def _replay_steps(self, sampler, key, n, known=None):
        """n decode steps: replays of a hipGraph of GRAPH_STEPS steps (all per-step state is device-resident, so a graph of
        several steps is just the step captured several times; it costs the host 1/GRAPH_STEPS of the launches -- the token
        stages of several batches are enqueued by concurrent host threads) and of the one-step graph for the remainder.
        `known`: columns of c['codes'] that are final before the first step; with a `progress` callback set, it is told after
        every replay how many are final then (a consumer may read them behind an event it records there)."""
        k = self.GRAPH_STEPS
        if self.warm_only:
            self._decode_graph(sampler, key + (k,), steps=k)
            self._decode_graph(sampler, key)
            return
        report = self.progress if known is not None else None
        codes = self._cache["codes"]
        if n >= 2 * k:
            long_graph = self._decode_graph(sampler, key + (k,), steps=k)
            for _ in range(n // k):
                long_graph.replay()
                if report is not None:
                    known += k
                    report(known, codes)
            n -= (n // k) * k
        if n:
            graph = self._decode_graph(sampler, key)
            for _ in range(n):
                graph.replay()
                if report is not None:
                    known += 1
                    report(known, codes)

    def _decode_graph(self, sampler, key, steps=1):
        """`steps` decode steps of the current cache captured in a hipGraph (once per key).  Captured on live state: a warm-up
        step runs first (one-time attribute calls, descriptor), then the device-resident state is put back."""
        c = self._cache
        self._decode_desc(sampler)   # a changed weight version / sampler rebuilds the descriptor and drops the graphs built on the old one
        graph = self._graphs.get(key)
        if graph is None:
            state = {k: c[k].clone() for k in ("len_dev", "widx", "tok", "codes", "state")}
            warm = torch.cuda.Stream()
            warm.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(warm):
                self._decode_body(sampler)
            torch.cuda.current_stream().wait_stream(warm)
            for k, v in state.items():
                c[k].copy_(v)
            graph = torch.cuda.CUDAGraph()
            # thread_local: with torch.distributed initialised the RCCL watchdog thread issues HIP calls of its own, which the
            # default 'global' capture mode would treat as capture violations
            with torch.cuda.graph(graph, capture_error_mode="thread_local"):
                for _ in range(steps):
                    self._decode_body(sampler)
            for k, v in state.items():   # capture does not execute; kept for clarity
                c[k].copy_(v)
            self._graphs[key] = graph
        return graph

    @torch.no_grad()
    def _generate_stream(self, code, state_code, add_len, cond_idx, delta_length_cond, sampler, state_sampler, host_noise, trace,
                         use_graph):
        """`Transformer.fill_code` (transformer_model.py:343-392) for a frame + ancillary token stream on the KV cache.
        The merged sequence only ever grows at its end, so each new token (plus, at a frame boundary, the next frame's
        given ancillary tokens) is appended to the cache: runs of frame tokens replay the captured decode step, the
        rest goes through `extend`."""
        cfg = self.config
        if cfg.state_front:
            return self._generate_state_front(code, state_code, add_len, cond_idx, delta_length_cond, sampler, state_sampler, host_noise, trace)
        size, ss = cfg.shape[0] * cfg.shape[1], cfg.state_size
        tot = size + ss
        b, t0 = code.shape
        ns0 = state_code.shape[1]
        cap = cfg.num_blocks * ss
        if t0 < size:
            raise NotImplementedError("less than one whole frame of tokens with an ancillary stream (the reference drops them)")
        n_cond = cond_idx.shape[1] if cond_idx is not None else 0
        slen = lambda nc, nst: len(self.stream_kinds(nc, nst))
        # which stream each of the add_len picks goes to (transformer_model.py:352), and whether the step before pick i
        # appends exactly the frame token picked at i-1 (then it is a replay of the captured decode step)
        plan, single = [], []
        nc, nst = t0, ns0
        for i in range(add_len):
            fed = slen(nc, nst)
            is_state = fed % tot < ss
            plan.append(1 if is_state else 0)
            single.append(i > 0 and plan[i - 1] == 0 and not is_state and fed - prev_fed == 1)
            prev_fed = fed
            if is_state:
                nst += 1
            else:
                nc += 1
        assert n_cond + nc <= self.block_size, "Cannot forward, model block size is exhausted."  # mingpt.py:299 (frame tokens)
        max_len = n_cond + slen(nc, nst)
        c = self.begin(b, max_len, stream=True)
        dev = code.device
        frame_codes = c["codes"]          # the captured step stores its picks here
        frame_codes[:, :t0] = code
        state_buf = torch.zeros(b, max(nst, ns0), dtype=torch.int64, device=dev)
        state_buf[:, :ns0] = state_code
        n_code, n_state = t0, ns0
        device_noise = sampler["sample"] and sampler["noise"] == "device"
        eager = trace is not None or (sampler["sample"] and not device_noise) or not use_graph
        words = self._philox_words() if device_noise else None
        self._set_decode_state(words)

        def rows_of(lo, hi):  # merged-sequence rows [lo, hi) from the current buffers
            kinds = self.stream_kinds(n_code, n_state)[lo:hi]
            cols = [frame_codes[:, i:i + 1] if k == 0 else state_buf[:, i:i + 1] + cfg.vocab_size for k, i in kinds]
            return torch.cat(cols, dim=1)

        def draw(kind, width, pick):
            """(noise, words, step) of eager pick number `pick`: in-kernel Philox with the call word's top bit set (the captured
            steps use the plain call word and their own step counter), or host noise."""
            smp = state_sampler if kind else sampler
            if not smp["sample"]:
                return None, None, None
            if device_noise:
                return None, words, (pick, 0x80000000)
            return host_noise(b, width).to(dev, non_blocking=True), None, None

        logits = self.prefill(self._stream_rows(code, state_code), cond_idx, delta_length_cond)
        fed = slen(n_code, n_state)
        i = 0
        while i < add_len:
            if single[i] and not eager:
                run = 1
                while i + run < add_len and single[i + run]:
                    run += 1
                c["len_dev"].fill_(c["len"])
                c["widx"].fill_(n_code)
                self._replay_steps(sampler, (sampler["sample"], sampler["top_k"], sampler["temperature"], n_cond, b, "stream"), run)
                c["len"] += run          # each replay appended the previous pick, then picked the next
                n_code += run
                fed += run
                i += run
                logits = None
                continue
            if logits is None:           # after a graph run: the last pick (and what follows it) is not in the cache yet
                new_fed = slen(n_code, n_state)
                logits = self.extend(rows_of(fed, new_fed))
                fed = new_fed
            if plan[i]:
                lg = logits[:, :state_sampler["vocab"]].contiguous()
                if trace is not None:
                    trace.append(lg.clone())
                nz, wd, stp = draw(1, lg.shape[1], i)
                tok = self._pick(lg, state_sampler, nz, None, wd, stp)
                state_buf[:, n_state] = tok
                n_state += 1
            else:
                if trace is not None:
                    trace.append(logits.clone())
                nz, wd, stp = draw(0, logits.shape[1], i)
                self._pick(logits, sampler, nz, c["tok"].view(-1), wd, stp)
                frame_codes[:, n_code] = c["tok"][:, 0]
                n_code += 1
            i += 1
            logits = None
            if i < add_len and not (single[i] and not eager):
                new_fed = slen(n_code, n_state)
                logits = self.extend(rows_of(fed, new_fed))
                fed = new_fed
        return frame_codes[:, :n_code].clone(), state_buf[:, :n_state].clone()

    @torch.no_grad()
    def _generate_state_front(self, code, state_code, add_len, cond_idx, delta_length_cond, sampler, state_sampler, host_noise, trace):
        """`Transformer.fill_code` (transformer_model.py:343-357) with `--x_state_front` (mingpt.py:261-263): the merged sequence
        is [every ancillary token][every frame token], while the picks still alternate between the streams by the interleaved
        count (transformer_model.py:352).  A new ancillary token therefore lands in the MIDDLE of the sequence, every frame
        token behind it moves one position and attends to it: nothing of a KV cache survives such a pick, and the prediction
        is read at the last FRAME position either way.  Every pick is a full forward over the merged prefix -- what the
        reference does for every token of every configuration."""
        cfg = self.config
        size, ss = cfg.shape[0] * cfg.shape[1], cfg.state_size
        tot, cap = size + ss, cfg.num_blocks * ss
        b = code.shape[0]
        dev = code.device
        n_cond = cond_idx.shape[1] if cond_idx is not None else 0
        device_noise = sampler["noise"] == "device"
        words = self._philox_words() if device_noise and (sampler["sample"] or state_sampler["sample"]) else None
        state_code = state_code.to(dev)
        for i in range(add_len):
            st = state_code[:, :cap]
            assert n_cond + code.shape[1] <= self.block_size, "Cannot forward, model block size is exhausted."  # mingpt.py:299
            rows = self._stream_rows(code, st)
            self.begin(b, n_cond + rows.shape[1], stream=True, n_state_front=st.shape[1])
            logits = self.prefill(rows, cond_idx, delta_length_cond)
            is_state = rows.shape[1] % tot < ss
            smp = state_sampler if is_state else sampler
            lg = logits[:, :state_sampler["vocab"]].contiguous() if is_state else logits
            if trace is not None:
                trace.append(lg.clone())
            nz, wd, stp = None, None, None
            if smp["sample"]:
                if device_noise:
                    wd, stp = words, (i, 0x80000000)
                else:
                    nz = host_noise(b, lg.shape[1]).to(dev, non_blocking=True)
            tok = self._pick(lg, smp, nz, None, wd, stp).view(b, 1).to(torch.int64)
            if is_state:
                state_code = torch.cat((state_code, tok), dim=1)
            else:
                code = torch.cat((code, tok), dim=1)
        return code, state_code

    @torch.no_grad()
    def generate(self, code, add_len, cond_idx=None, delta_length_cond=None, sample=False, top_k=None, temperature=1.0,
                 noise="device", host_noise=None, trace=None, use_graph=True, state_code=None, state_sampler=None, lbl_idx=None):
        """code [B,t0] -> [B, t0+add_len]: one prefill, then KV-cached decode steps.
        With an ancillary stream (`state_code` [B,ns], `state_sampler` = dict(sample, top_k, temperature, vocab)) the
        add_len new tokens are split between the two streams as the reference does and (code, state_code) is returned.

        Without a trace the decode step is captured ONCE in a hipGraph and replayed, whatever the sampler: greedy, in-kernel
        Philox noise, or host-supplied noise (`host_noise(b, v) -> [B,V]` Exp(1) blocks, the reference-reproducible stream) -- the
        call's whole noise stream is then drawn up front, in the reference's order, and the captured step reads block after block
        of it (`ccvs_gpt_decode.noise_stream`); `noise_streams` (set by the caller) supplies pre-drawn streams instead, one per row
        group.  With a trace, `use_graph=False` or fewer than three new tokens the same step runs eagerly."""
        b, t0 = code.shape
        use_cond = cond_idx is not None and 0 not in cond_idx.size()
        n_cond = cond_idx.shape[1] if use_cond else 0
        sampler = {"sample": bool(sample), "top_k": top_k, "temperature": float(temperature), "noise": noise}
        n_pre = self.n_prefix()
        if state_code is not None and 0 not in state_code.size():
            if n_pre:
                raise NotImplementedError("label / start tokens together with an ancillary token stream")
            return self._generate_stream(code, state_code, add_len, cond_idx if use_cond else None,
                                         delta_length_cond if use_cond else None, sampler, state_sampler, host_noise, trace, use_graph)
        eager = trace is not None or not use_graph or add_len < 3
        host = bool(sample) and noise != "device"
        if host and eager and self.noise_streams is not None:   # pre-drawn streams, eager steps: block i of every group for pick i
            streams, self.noise_streams = self.noise_streams, None
            blocks = iter(range(add_len))

            def host_noise(nb, nv):
                i = next(blocks)
                return streams[0][i] if len(streams) == 1 else torch.cat([s_[i] for s_ in streams], dim=0)
        # host-drawn noise inside the captured step: the call's whole noise stream is on the device before the first replay -- unless
        # the caller did not pre-draw it and it would be too large to hold (Kinetics: 704 x 64 x 16384 floats = 3 GB; CCVS_NOISE_STREAM_MAX_MB):
        # then every step draws and uploads its own block, eagerly, as in rounds 1-4
        if host and not eager and self.noise_streams is None:
            import os
            if 4.0 * add_len * b * self.head.weight.shape[0] / 2 ** 20 > float(os.environ.get("CCVS_NOISE_STREAM_MAX_MB", "1024")):
                eager = True
        sampler["stream"] = host and not eager
        max_len = n_pre + n_cond + t0 + add_len
        if self.persistent_step and not self.warm_only:
            self.check_steps()   # the previous sequence's steps on this stream all passed their barriers
        c = self.begin(b, max_len)

        device_rng = sample and noise == "device"
        words = self._philox_words() if device_rng else None
        logits = self.prefill(code, cond_idx if use_cond else None, delta_length_cond if use_cond else None, lbl_idx=lbl_idx)
        if trace is not None:
            trace.append(logits.clone())
        c["codes"][:, :t0] = code
        c["len_dev"].fill_(n_pre + n_cond + t0)
        self._set_decode_state(words)     # the step counter (a Philox counter word / the block index of a noise stream) restarts with the call
        if device_rng:                    # first pick: step word 0xffffffff (the decode steps count 0, 1, ...)
            self._emit(logits, sampler, None, t0, words=words, step=(0xffffffff, 0))
        elif sampler["stream"]:           # block 0 of every group's stream; the decode steps read blocks 1, 2, ... (pointer table)
            self._emit(logits, sampler, self._stage_noise_streams(b, add_len, logits.shape[1], host_noise), t0)
        else:
            self._emit(logits, sampler, host_noise(b, logits.shape[1]).to(logits.device, non_blocking=True) if sample else None, t0)

        if not eager:
            if self.progress is not None:
                self.progress(t0 + 1, c["codes"])
            self._replay_steps(sampler, (bool(sample), top_k, float(temperature), n_pre + n_cond, b), add_len - 1, known=t0 + 1)
        else:
            for _ in range(add_len - 1):
                nz = None
                if host:
                    nz = host_noise(b, self.head.weight.shape[0]).to(code.device, non_blocking=True)
                self._decode_body(sampler, noise=nz, trace=trace)
        c["len"] = n_pre + n_cond + t0 + add_len - 1
        return c["codes"][:, :t0 + add_len].clone()

    def _stage_noise_streams(self, b, add_len, v, host_noise):
        """The host-drawn noise of a graph-replayed generate() call: the stream of every row group ([add_len, rows, V] Exp(1)
        blocks in the reference generator's order -- `noise_streams` set by the caller, else drawn here, one group only) is made
        resident, the cache's pointer table is set to block 1 of each (the captured steps read block `steps completed`), and the
        [B, V] block of the first pick is returned.  A warm-only call (graph capture) consumes nothing of the generator."""
        c = self._cache
        dev = c["noise_ptrs"].device
        groups = c["G"]
        rows = b // groups
        streams, self.noise_streams = self.noise_streams, None
        if self.warm_only:
            streams = [torch.ones(2, rows, v, dtype=torch.float32, device=dev)] * groups
        elif streams is None:
            assert groups == 1, "host-drawn sampling noise of stacked batches is pre-drawn per batch (GPT.noise_streams)"
            buf = torch.empty(add_len, b, v, dtype=torch.float32, pin_memory=True)
            for i in range(add_len):      # one [B, V] block per pick, exactly the draws torch.multinomial makes
                buf[i].copy_(host_noise(b, v))
            streams = [buf.to(dev, non_blocking=True)]
        assert len(streams) == groups, (len(streams), groups)
        cur = torch.cuda.current_stream()
        for s_ in streams:
            assert s_.is_cuda and s_.dtype == torch.float32 and s_.is_contiguous() and tuple(s_.shape[1:]) == (rows, v) and s_.shape[0] >= min(add_len, 2), s_.shape
            s_.record_stream(cur)
        ptrs = torch.tensor([s_.data_ptr() + 4 * rows * v for s_ in streams], dtype=torch.int64)
        c["noise_ptrs"].copy_(ptrs, non_blocking=True)
        c["noise_keep"] = streams         # alive until the next call replaces them (stream-ordered behind this call's replays)
        return streams[0][0] if groups == 1 else torch.cat([s_[0] for s_ in streams], dim=0)

    # ------------------------------------------------------------------ reference-shaped forward
    @torch.no_grad()
    def forward(self, idx, cond_idx=torch.tensor([]), state_idx=torch.tensor([]), lbl_idx=torch.tensor([]), delta_length_cond=None):
        """Teacher-forced logits [B, T, V] for positions after the conditioning prefix (mingpt.py:232-305); with an
        ancillary stream T counts the merged sequence."""
        t_cond = cond_idx.shape[1] if 0 not in cond_idx.size() else 0
        n_pre = self.n_prefix()
        assert t_cond + n_pre + idx.shape[1] <= self.block_size, "Cannot forward, model block size is exhausted."  # mingpt.py:299
        if 0 not in state_idx.size():
            if n_pre:   # same combination, same answer as generate(): not on the path, never a misleading assert
                raise NotImplementedError("label / start tokens together with an ancillary token stream")
            state_idx = state_idx[:, :self.config.num_blocks * self.config.state_size].to(idx.device)
            rows = self._stream_rows(idx, state_idx)
            self.begin(idx.shape[0], t_cond + rows.shape[1], stream=True, n_state_front=state_idx.shape[1])
            return self.prefill(rows, cond_idx if t_cond else None, delta_length_cond, all_logits=True)
        self.begin(idx.shape[0], n_pre + t_cond + idx.shape[1])
        return self.prefill(idx, cond_idx if t_cond else None, delta_length_cond, all_logits=True, lbl_idx=lbl_idx)
