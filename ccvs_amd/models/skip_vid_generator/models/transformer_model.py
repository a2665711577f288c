"""Transformer wrapper: the autoregressive sampling loop over latent tokens.

Host-side mirror of the reference's `models/skip_vid_generator/models/transformer_model.py`
(inference only): `Transformer(opt, is_train, is_main, logger)`, `forward(data, mode='inference',
total_len=...) -> {"code", "state_code"}`, `generate_fake`, `fill_code`, `get_icode`,
`top_k_logits` keep their names, argument meaning and error behaviour.

The loop itself is re-designed: one prefill of the known tokens, then one KV-cached decode step
per new token (the reference re-runs the full prefix each time, transformer_model.py:343-350);
temperature / top-k / softmax / pick run fused on the GPU and write straight into the code
buffer.  Sampling reproduces `torch.multinomial(probs, 1)`: that op is `argmax(probs / q)` with
`q ~ Exp(1)` drawn from the generator, so with `sample_noise='host'` the noise is drawn from the
same CPU generator stream the reference's CPU path would consume and uploaded; with
`sample_noise='device'` it is drawn on the GPU (throughput mode).
"""
import torch

from .mingpt import GPT
from ccvs_amd.tools.utils import to_cuda
from ccvs_amd.models import load_network, print_network
from ccvs_amd import ops


class Transformer(torch.nn.Module):
    def __init__(self, opt, is_train=False, is_main=True, logger=None):
        super().__init__()
        if is_train:
            raise NotImplementedError("training is outside the MI355X hot path")
        self.opt = opt
        self.is_main = is_main
        self.net_t = self.initialize_networks(is_train)
        self.logger = logger if self.is_main else None
        height, width = self.opt.z_shape
        self.size = height * width
        self.state_size = self.opt.state_size
        self.tot_size = self.size + self.state_size
        self.sample_noise = getattr(opt, "sample_noise", "host")  # 'host' (reference-reproducible) | 'device'
        self.generator = None                                      # optional torch.Generator for the noise
        self.trace = None                                          # optional list collecting per-step logits (tests)

    def forward(self, data, prefix='', mode='', total_len=None, log=False, global_iter=None, show_progress=False):
        code, state_code, cond_code, delta_length_cond, vid_lbl = self.preprocess_input(data)
        if mode == 'inference':
            return self.generate_fake(code, state_code, cond_code, delta_length_cond, vid_lbl, total_len, show_progress)
        if mode in ('transformer', 'eval_transformer'):
            raise NotImplementedError(f"mode '{mode}' (training loss) is outside the MI355X hot path")
        raise ValueError(f"mode '{mode}' is invalid")

    def preprocess_input(self, data):
        """transformer_model.py:48-54."""
        data["code"] = to_cuda(data, "code", flatten_empty=False)
        data["state_code"] = to_cuda(data, "state_code", flatten_empty=False)
        data["cond_code"] = to_cuda(data, "cond_code")
        data["vid_lbl"] = to_cuda(data, "vid_lbl")
        data["delta_length_cond"] = to_cuda(data, "delta_length_cond")
        return data["code"], data["state_code"], data["cond_code"], data["delta_length_cond"], data["vid_lbl"]

    def initialize_networks(self, is_train):
        """transformer_model.py:57-73."""
        opt = self.opt
        if getattr(opt, "is_continuous", False):
            raise NotImplementedError("continuous-token GPT (CGPT) is outside the hot path")
        num_lbl = len(opt.categories) if getattr(opt, "categories", None) is not None else None
        net_t = GPT(vocab_size=opt.z_num, block_size=opt.z_len, n_layer=opt.n_layer, n_head=opt.n_head, n_embd=opt.n_embd,
                    emb_mode=opt.emb_mode, shape=opt.z_shape, state_vocab_size=opt.state_num, num_blocks=opt.num_blocks,
                    state_size=opt.state_size, use_start_token=opt.use_start_token, use_lbl=opt.cat, num_lbl=num_lbl,
                    state_front=opt.state_front).cuda()
        if self.is_main:
            net_t = load_network(net_t, "transformer_t", opt, head_to_n=getattr(opt, "head_to_n", 0))
        return net_t

    def top_k_logits(self, logits, k):
        """transformer_model.py:256-260 (torch ops; the fused kernel applies the same rule)."""
        v, _ = torch.topk(logits, k)
        out = logits.clone()
        out[out < v[..., [-1]]] = -float('Inf')
        return out

    def token_windows(self, total_len):
        """The sliding token windows behind the first one (transformer_model.py:301-326): window w starts w frames into the
        clip and produces one more chunk of `z_chunk` tokens -- fewer for the last.  Yields (frames to drop in front, tokens to
        add or None for a full chunk)."""
        produced, w = self.opt.z_len, 1
        while produced < total_len:
            step = min(self.opt.z_chunk, total_len - produced)
            yield w, (step if step < self.opt.z_chunk else None)
            produced += step
            w += 1

    @torch.no_grad()
    def generate_fake(self, code, state_code, cond_code, delta_length_cond, vid_lbl, total_len, show_progress=False):
        """transformer_model.py:263-328.  Up to `z_len` tokens are one `fill_code`; beyond that the token window slides by one
        frame per chunk (`token_windows`): every window restarts its positions at 0, so the kept part is re-prefilled, and the
        conditioning prefix moves one frame closer (`delta_length_cond - 1` per window).  The ancillary (state / STFT) stream
        slides with its frames."""
        opt = self.opt
        has_state = 0 not in state_code.size()
        n_cond = cond_code.size(1) if 0 not in cond_code.size() else 0
        n_state_known = min(state_code.size(1), opt.state_size * opt.num_blocks) if has_state else 0
        if total_len is None or int(total_len) <= opt.z_len:
            add_len = None if total_len is None else int(total_len) - code.size(1) - n_cond - n_state_known
            code, state_code = self.fill_code(code, state_code, cond_code, delta_length_cond, vid_lbl, add_len=add_len)
            return {"code": code, "state_code": state_code}
        code, state_code = self.fill_code(code, state_code, cond_code, delta_length_cond, vid_lbl)   # the first window, filled
        for w, add_len in self.token_windows(int(total_len)):
            if n_cond:
                delta_length_cond = delta_length_cond - 1
            win_code = code[:, w * self.size:]
            win_state = state_code[:, w * self.state_size:] if has_state else state_code
            out_code, out_state = self.fill_code(win_code, win_state, cond_code, delta_length_cond, vid_lbl, add_len=add_len)
            grown = out_code.size(1) - win_code.size(1)
            code = torch.cat([code, out_code[:, -grown:]], dim=1)   # (sic) grown == 0 would append the whole window: `[-0:]`, as upstream (:311-312)
            if has_state and out_state.size(1) > win_state.size(1):
                state_code = torch.cat([state_code, out_state[:, win_state.size(1):]], dim=1)
        return {"code": code, "state_code": state_code}

    def _noise(self, b, v, device):
        if self.sample_noise == "device":
            return torch.empty(b, v, dtype=torch.float32, device=device).exponential_(1)
        q = torch.empty(b, v, dtype=torch.float32).exponential_(1, generator=self.generator)
        return q.to(device, non_blocking=True)

    @torch.no_grad()
    def fill_code(self, code, state_code, cond_code, delta_length_cond, vid_lbl, add_len=None, show_progress=False):
        """transformer_model.py:331-392 on the KV-cached engine."""
        opt = self.opt
        if getattr(opt, "beam_size", None) is not None:
            return self._beam_fill(code, state_code, cond_code, add_len)
        b, t0 = code.shape
        n_cond = cond_code.size(1) if 0 not in cond_code.size() else 0
        use_state = 0 not in state_code.size()
        if add_len is None:  # transformer_model.py:336-339
            add_len = opt.z_len - t0 - n_cond
            add_len -= min(state_code.size(1), opt.state_size * opt.num_blocks) if use_state else 0
        if add_len <= 0:
            return code, state_code
        def host_noise(nb, nv):  # the stream torch.multinomial would consume (module docstring)
            return torch.empty(nb, nv, dtype=torch.float32).exponential_(1, generator=self.generator)

        state_sampler = None
        if use_state:  # ancillary tokens: first state_num logits, their own sampling options (transformer_model.py:353-356)
            state_sampler = {"sample": bool(opt.sample_state), "top_k": opt.top_k_state, "temperature": float(opt.temperature_state),
                             "vocab": opt.state_num}
        out = self.net_t.generate(code, add_len, cond_code if n_cond else None, delta_length_cond if n_cond else None,
                                  sample=opt.sample, top_k=opt.top_k, temperature=opt.temperature, noise=self.sample_noise,
                                  host_noise=host_noise, trace=self.trace, use_graph=getattr(opt, "use_graph", True),
                                  state_code=state_code if use_state else None, state_sampler=state_sampler,
                                  lbl_idx=vid_lbl if getattr(opt, "cat", False) else None)
        if use_state:
            return out
        return out, state_code

    @torch.no_grad()
    def _beam_fill(self, code, state_code, cond_code, add_len):
        """Beam search of `fill_code` (transformer_model.py:358-391) on the KV cache.

        The reference keeps `beam_size` hypotheses per clip as extra batch rows.  The first pick takes `beam_size` proposals
        from the prefix's distribution (:362-367); afterwards each hypothesis either continues with ONE pick whose
        log-probability is added to its score (default, :369-372), or -- with `--x_no_sample` -- proposes `beam_size`
        continuations, the clip's beam^2 candidates are pruned to the best `beam_size` by accumulated score and the
        hypotheses are re-ordered (:374-386); the best hypothesis is returned (:387-391).  Here the hypotheses are rows of
        the cache from the start (the prefix is prefilled once per row) and a re-ordering is a row gather of the cached
        keys / values.  (sic) In the pruning branch the reference adds score[b, j] -- not score[b, i] -- to candidate j of
        hypothesis i (`log_p.unsqueeze(1).repeat(...)`, :377-378); kept, the golden vectors pin it.
        The reference's expansion does not replicate `cond_code` / `state_code`, so it fails with them: so does this."""
        opt = self.opt
        beam = int(opt.beam_size)
        if 0 not in cond_code.size() or 0 not in state_code.size():
            raise NotImplementedError("beam search with a conditioning prefix / ancillary stream: the reference's expansion "
                                      "(transformer_model.py:361) does not replicate them and fails too")
        bs, t0 = code.shape
        if add_len is None:
            add_len = opt.z_len - t0
        if add_len <= 0:
            return code, state_code
        net = self.net_t
        if net.config.use_lbl:
            raise NotImplementedError("beam search with a class-label token: the reference's expansion (transformer_model.py:361) "
                                      "does not replicate vid_lbl and fails too")
        rows = code.repeat_interleave(beam, dim=0).contiguous()            # row b * beam + i = hypothesis i of clip b
        net.begin(bs * beam, net.n_prefix() + t0 + add_len)                 # (a start token occupies one more cache position)
        logits = net.prefill(rows)                                           # identical for the beam rows of a clip
        icode, log_p = self.get_icode(logits[::beam].unsqueeze(1), opt.temperature, opt.top_k, opt.sample, n=beam)   # [bs, beam]
        rows = torch.cat((rows, icode.reshape(-1, 1)), dim=1)
        for _ in range(add_len - 1):
            logits = net.step(rows[:, -1:].contiguous())
            if not getattr(opt, "no_sample", False):
                icode, ilog_p = self.get_icode(logits.unsqueeze(1), opt.temperature, opt.top_k, opt.sample, n=1)
                log_p = log_p + ilog_p.view(bs, beam)
                rows = torch.cat((rows, icode.view(-1, 1)), dim=1)
            else:
                icode, ilog_p = self.get_icode(logits.unsqueeze(1), opt.temperature, opt.top_k, opt.sample, n=beam)   # [bs*beam, beam]
                cand = log_p.unsqueeze(1).repeat(1, beam, 1) + ilog_p.view(bs, beam, beam)
                log_p, keep = torch.topk(cand.view(bs, beam * beam), dim=1, k=beam)
                icode = torch.gather(icode.view(bs, beam * beam), dim=1, index=keep).view(-1, 1)
                parent = (torch.arange(bs, device=keep.device).view(bs, 1) * beam + keep // beam).view(-1)   # cache row each survivor extends
                net.reorder_cache(parent)
                rows = torch.cat((rows[parent], icode), dim=1)
        best = torch.topk(log_p, dim=1, k=1)[1]                            # [bs, 1]
        rows = rows.view(bs, beam, -1)
        out = torch.gather(rows, dim=1, index=best.unsqueeze(-1).repeat(1, 1, rows.size(-1))).view(bs, rows.size(-1))
        return out, state_code

    @torch.no_grad()
    def get_icode(self, logits, temperature, top_k, sample, n=1):
        """transformer_model.py:395-409: logits [B,T,V] -> (icode [B,n], log p [B,n]) in one kernel (`ccvs_sample_topn`):
        temperature, top-k mask, softmax, the n best of the probabilities -- or of probabilities / Exp(1) noise, which is what
        `torch.multinomial(probs, n)` without replacement computes, the noise drawn like the single pick's -- and their log p."""
        last = logits[:, -1].contiguous()
        noise = self._noise(last.shape[0], last.shape[1], last.device) if sample else None
        return ops.sample_topn(last, top_k, temperature, n, noise=noise)
