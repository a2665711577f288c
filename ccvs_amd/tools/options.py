"""Command-line flags of the synthesis hot path, with the reference's names, prefixes,
defaults and Namespace split (tools/options.py:17-633).

The reference declares ~330 flags; `helpers/generator.py` and the inference halves of the
models read about 90 of them.  Those are declared here with identical
names / defaults; the reference's launch scripts also pass training-only flags (`--q_lr`,
`--q_gan_loss`, ...), which are accepted and ignored so the same command lines parse.
`Options().parse(...)` returns the same dict of Namespaces: base flags are copied into every
model Namespace and the prefixes `q_` / `x_` / `s_` / `a_` are stripped.
"""
import argparse
import datetime
import os
from argparse import Namespace


def str2bool(v):
    if isinstance(v, bool):
        return v
    if v.lower() in ("yes", "true", "t", "y", "1"):
        return True
    if v.lower() in ("no", "false", "f", "n", "0"):
        return False
    raise argparse.ArgumentTypeError("Boolean value expected.")


def _flag(parser, name, default=False):
    parser.add_argument(name, type=str2bool, nargs="?", const=True, default=default)


# dataset presets applied after a first parse (tools/options.py:396-450)
_PRESETS = {
    "bairhd": dict(dataroot="datasets/bairhd", true_ratio=1, aspect_ratio=1, true_dim=256, categories=None, fps=4),
    "kinetics600": dict(dataroot="datasets/kinetics", true_ratio=1, aspect_ratio=1, true_dim=256, imagenet_norm=True),
    "drums": dict(dataroot="datasets/drums", true_ratio=1, aspect_ratio=1, true_dim=96, categories=None, fps=30),
    "ucf101": dict(dataroot="datasets/ucf101", true_ratio=1, aspect_ratio=1, true_dim=256, categories=None, fps=4),
}


class Options:
    def initialize(self, parser):
        p = parser
        # ---- base (tools/options.py:36-153)
        p.add_argument("--name", type=str, default="ccvs_amd")
        p.add_argument("--gpu_ids", type=str, default="0")
        p.add_argument("--local_rank", type=int, default=int(os.environ.get("LOCAL_RANK", 0)))
        p.add_argument("--dataset", type=str, default="youtube_faces")
        p.add_argument("--dataroot", type=str, default=None)
        p.add_argument("--max_dim", type=int, default=512)
        p.add_argument("--dim", type=int, default=-1)
        p.add_argument("--true_dim", type=int, default=1024)
        p.add_argument("--true_ratio", type=float, default=1.0)
        p.add_argument("--aspect_ratio", type=float, default=2.0)
        _flag(p, "--imagenet_norm")
        p.add_argument("--vid_len", type=int, default=16)
        p.add_argument("--p2p_len", type=int, default=None)
        p.add_argument("--batch_size_vid", type=int, default=1)
        p.add_argument("--batch_size_img", type=int, default=1)
        p.add_argument("--batch_size_valid_mult", type=int, default=1)
        p.add_argument("--categories", type=str, nargs="+", default=None)
        p.add_argument("--num_workers", type=int, default=8)
        p.add_argument("--fps", type=int, default=10)
        p.add_argument("--save_path", type=str, default="./")
        p.add_argument("--n_iter", type=int, default=1000)
        p.add_argument("--iter_function", type=str, default="iter")
        p.add_argument("--seed", type=int, default=0, help="(ccvs_amd) base seed of the synthetic input / sampling noise")
        for f in ("rec_only", "step_by_step", "gen_from_img", "keep_state", "custom_state", "layout", "include_id"):
            _flag(p, "--" + f)
        p.add_argument("--down_size", type=int, nargs="+", default=None)
        # (ccvs_amd) the teacher-forced "rec" decode of the real codes, which the reference always runs unless gen_from_img
        # (helpers/generator.py:172-189); on by default like the reference, switchable because it is not part of the
        # synthesized-frames metric
        _flag(p, "--rec_pass", default=True)
        # (ccvs_amd) the reference encodes EVERY frame of the input clip (helpers/generator.py:69) although synthesis reads the
        # conditioning frames only -- the other codes feed its rec pass and `enc_code`.  `--encode_all false`: with the rec pass
        # off and no state / audio / point-to-point conditioning, encode just the frames the conditioning crop keeps (the same
        # bits for `fake`: frames are encoded independently); `enc_code` then holds those frames' codes only.
        _flag(p, "--encode_all", default=True)
        # ---- q_: quantised video model (tools/options.py:159-264)
        p.add_argument("--q_enc_model", type=str, default="taming")
        p.add_argument("--q_dec_model", type=str, default="stylegan2")
        for f in ("use_ema", "is_continuous", "use_enc", "use_dec", "use_q_anyway", "not_strict", "use_inter", "use_masked_flow",
                  "use_deformed_conv", "use_tradeoff", "no_corr", "no_proj", "normalize_out", "keep_first", "skip_rgb", "skip_tanh",
                  "use_layout", "same_decoder_layout"):
            _flag(p, "--q_" + f)
        p.add_argument("--q_necf", type=int, default=128)
        p.add_argument("--q_necf_mult", type=int, nargs="+", default=[1, 1, 2, 2, 4])
        p.add_argument("--q_ndcf", type=int, default=128)        # parsed and ignored by the decoder, like the reference
        p.add_argument("--q_ndcf_mult", type=int, nargs="+", default=[1, 1, 2, 2, 4])
        p.add_argument("--q_z_size", type=int, default=256)
        p.add_argument("--q_z_num", type=int, default=256)
        p.add_argument("--q_z_mult", type=int, default=1)
        p.add_argument("--q_z_shape", type=int, nargs="+", default=[16, 16])
        p.add_argument("--q_load_path", type=str, default=None)
        p.add_argument("--q_which_iter", type=str, default=0)
        p.add_argument("--q_block_delta", type=int, default=None)
        p.add_argument("--q_inter_p", type=float, default=0.5)
        p.add_argument("--q_skip_mode", type=str, default="enc")
        p.add_argument("--q_skip_context", type=int, nargs="+", default=[1])
        p.add_argument("--q_n_first", type=int, default=1)
        p.add_argument("--q_skip_memory", type=int, default=1)
        # ---- x_: transformer (tools/options.py:270-345)
        p.add_argument("--x_z_num", type=int, default=256)
        p.add_argument("--x_z_len", type=int, default=256)
        p.add_argument("--x_num_blocks", type=int, default=16)
        p.add_argument("--x_cond_len", type=int, default=256)
        p.add_argument("--x_z_chunk", type=int, default=256)
        p.add_argument("--x_n_layer", type=int, default=24)
        p.add_argument("--x_n_head", type=int, default=16)
        p.add_argument("--x_n_embd", type=int, default=1024)
        for f in ("is_continuous", "not_strict", "sample", "no_sample", "p2p", "state", "state_front", "sample_state",
                  "use_start_token", "cat", "stft", "deblurring"):
            _flag(p, "--x_" + f)
        p.add_argument("--x_load_path", type=str, default=None)
        p.add_argument("--x_which_iter", type=str, default=0)
        p.add_argument("--x_head_to_n", type=int, default=0)
        p.add_argument("--x_temperature", type=float, default=1.0)
        p.add_argument("--x_top_k", type=int, default=None)
        p.add_argument("--x_beam_size", type=int, default=None)
        p.add_argument("--x_emb_mode", type=str, default=None)
        p.add_argument("--x_z_shape", type=int, nargs="+", default=None)
        p.add_argument("--x_state_num", type=int, default=None)
        p.add_argument("--x_state_size", type=int, default=None)
        p.add_argument("--x_temperature_state", type=float, default=1.0)
        p.add_argument("--x_top_k_state", type=int, default=None)
        p.add_argument("--x_blur_sigma", type=int, default=10)
        p.add_argument("--x_sample_noise", type=str, default="host", help="(ccvs_amd) 'host': reference-reproducible noise, 'device'")
        # ---- s_ / a_: ancillary token streams (declared for parity of the Namespace split)
        p.add_argument("--s_state_size", type=int, default=0)
        p.add_argument("--s_state_num", type=int, default=0)
        p.add_argument("--s_z_shape", type=int, nargs="+", default=None)
        p.add_argument("--s_z_size", type=int, default=None)
        p.add_argument("--s_state_hsize", type=int, default=128)
        _flag(p, "--s_quantize_only")
        p.add_argument("--s_load_path", type=str, default=None)
        p.add_argument("--s_which_iter", type=str, default=0)
        _flag(p, "--s_not_strict")
        p.add_argument("--a_stft_size", type=int, default=None)
        p.add_argument("--a_stft_shape", type=int, nargs="+", default=None)
        p.add_argument("--a_stft_num", type=int, default=None)
        p.add_argument("--a_stft_hsize", type=int, default=128)
        p.add_argument("--a_load_path", type=str, default=None)
        p.add_argument("--a_which_iter", type=str, default=0)
        p.add_argument("--a_not_strict", action="store_true")
        return parser

    def update_defaults(self, opt, parser):
        """tools/options.py:396-450."""
        if opt.x_z_shape is None:
            parser.set_defaults(x_z_shape=opt.q_z_shape)
        if opt.x_state_num is None:
            parser.set_defaults(x_state_num=opt.s_state_num)
        if opt.x_state_size is None:
            parser.set_defaults(x_state_size=opt.s_state_size)
        if opt.s_z_shape is None:
            parser.set_defaults(s_z_shape=opt.q_z_shape)
        if opt.s_z_size is None:
            parser.set_defaults(s_z_size=opt.q_z_size)
        if opt.dim == -1:
            parser.set_defaults(dim=opt.max_dim)
        if opt.dataset in _PRESETS:
            parser.set_defaults(**_PRESETS[opt.dataset])
        return parser

    def gather_options(self, argv=None):
        parser = argparse.ArgumentParser(formatter_class=argparse.ArgumentDefaultsHelpFormatter, allow_abbrev=False)
        parser = self.initialize(parser)
        opt, _ = parser.parse_known_args(argv)
        parser = self.update_defaults(opt, parser)
        opt, unknown = parser.parse_known_args(argv)
        self.ignored = unknown  # training-only flags of the reference's scripts
        self.parser = parser
        return opt

    def split_options(self, opt):
        """tools/options.py:524-544."""
        groups = {"": Namespace(), "q_": Namespace(), "x_": Namespace(), "s_": Namespace(), "a_": Namespace()}
        for k, v in sorted(vars(opt).items()):
            for pre in ("q_", "x_", "s_", "a_"):
                if k.startswith(pre):
                    setattr(groups[pre], k[len(pre):], v)
                    break
            else:
                setattr(groups[""], k, v)
        return groups[""], groups["q_"], groups["x_"], groups["s_"], groups["a_"]

    def process_base(self, base_opt, signature):
        """tools/options.py:552-588 (the parts inference reads)."""
        base_opt.gpu_ids = [int(s) for s in str(base_opt.gpu_ids).split(",") if int(s) >= 0]
        base_opt.checkpoint_path = os.path.join(base_opt.save_path, "checkpoints", signature)
        base_opt.log_path = os.path.join(base_opt.save_path, "logs", signature)
        base_opt.result_path = os.path.join(base_opt.save_path, "results", signature)
        assert (base_opt.max_dim & (base_opt.max_dim - 1)) == 0, f"Max dim {base_opt.max_dim} must be power of two."
        base_opt.width_size = int(base_opt.dim * base_opt.aspect_ratio)
        base_opt.height_size = int(base_opt.width_size / base_opt.aspect_ratio)
        base_opt.signature = signature

    def parse(self, load_qvid_generator=False, load_transformer=False, load_extra_base=False, load_state_estimator=False,
              load_stft_ae=False, save=False, argv=None):
        opt = self.gather_options(argv)
        signature = datetime.datetime.now().strftime("%Y-%m-%d-%H:%M:%S") + "-" + opt.name
        base, q, x, s, a = self.split_options(opt)
        self.process_base(base, signature)
        for target in (q, x, s, a):
            for k, v in vars(base).items():
                setattr(target, k, v)
        self.opt = {"base": base, "extra_base": None,
                    "qvid_generator": q if load_qvid_generator else None,
                    "transformer": x if load_transformer else None,
                    "state_estimator": s if load_state_estimator else None,
                    "stft_ae": a if load_stft_ae else None}
        return self.opt


# canonical flag sets of the BASELINE.json configs (scripts/bairhd/*.sh, scripts/kinetics/save_videos.sh)
BAIR_ARGV = ["--name", "bair", "--dataset", "bairhd", "--max_dim", "256", "--vid_len", "16", "--x_cond_len", "64",
             "--x_sample", "--x_top_k", "100", "--x_temperature", "1.0",
             "--q_skip_context"] + [str(i) for i in range(1, 16)] + ["--q_skip_memory", "15",
             "--q_z_num", "1024", "--q_z_size", "512", "--q_z_shape", "8", "8", "--q_use_enc", "--q_use_dec",
             "--q_necf", "128", "--q_necf_mult", "1", "1", "2", "2", "4", "4", "--q_enc_model", "skipgan", "--q_dec_model", "skipgan",
             "--q_use_inter", "--q_inter_p", "0.75", "--x_z_num", "1024", "--x_z_len", "1024", "--x_n_layer", "24", "--x_n_head", "16",
             "--x_n_embd", "1024", "--x_z_chunk", "64", "--x_emb_mode", "temporal"]

KINETICS_ARGV = ["--name", "kinetics", "--dataset", "kinetics600", "--max_dim", "64", "--vid_len", "16", "--x_cond_len", "320",
                 "--x_sample", "--x_top_k", "100", "--x_temperature", "1.0",
                 "--q_skip_context"] + [str(i) for i in range(1, 9)] + ["--q_skip_memory", "8",
                 "--q_z_num", "16384", "--q_z_size", "512", "--q_z_shape", "8", "8", "--q_use_enc", "--q_use_dec",
                 "--q_necf", "256", "--q_necf_mult", "1", "1", "2", "2", "--q_enc_model", "skipgan", "--q_dec_model", "skipgan",
                 "--q_use_inter", "--q_inter_p", "0.75", "--x_z_num", "16384", "--x_z_len", "1024", "--x_n_layer", "24",
                 "--x_n_head", "16", "--x_n_embd", "1024", "--x_z_chunk", "64", "--x_emb_mode", "temporal"]

# BASELINE.json configs[3]: the reference's point-to-point launch line (scripts/bairhd/save_videos_p2p.sh:7-22; training-only
# flags dropped): start frame + end frame given, 14 frames interpolated.
BAIR_P2P_ARGV = list(BAIR_ARGV) + ["--x_p2p", "--p2p_len", "16"]

# BASELINE.json configs[4]: the reference's audio-conditioned launch line (scripts/drums/save_videos_audio_on.sh:9-23; training-only
# flags dropped): 128x128 / 5 levels, 45 frames from 15 conditioning frames, the 16 STFT tokens of every frame given
# (--keep_state), a 1280-token window of 16 x (16 + 64) that slides by one frame, the first 8 ring slots pinned.
# (sic) the script passes --dataset "drum", which matches no preset of tools/options.py:431-439 ("drums": aspect_ratio 1, fps 30);
# the preset's name is used here -- 8 x 8 tokens per frame need the square 128 x 128 frames.
DRUMS_ARGV = ["--name", "drums", "--dataset", "drums", "--max_dim", "128", "--vid_len", "45", "--x_cond_len", "960",
              "--x_sample", "--x_top_k", "100", "--x_temperature", "1.0", "--keep_state",
              "--q_skip_context"] + [str(i) for i in range(1, 16)] + ["--q_skip_memory", "15", "--q_keep_first", "--q_n_first", "8",
              "--q_z_num", "1024", "--q_z_size", "512", "--q_z_shape", "8", "8", "--q_use_enc", "--q_use_dec",
              "--q_necf", "128", "--q_necf_mult", "1", "1", "2", "2", "4", "--q_enc_model", "skipgan", "--q_dec_model", "skipgan",
              "--q_use_inter", "--q_inter_p", "0.75",
              "--a_stft_num", "1024", "--a_stft_size", "512", "--a_stft_hsize", "512", "--a_stft_shape", "8", "2",
              "--x_z_num", "1024", "--x_z_len", "1280", "--x_n_layer", "24", "--x_n_head", "16", "--x_n_embd", "1024", "--x_z_chunk", "80",
              "--x_num_blocks", "16", "--x_state_num", "1024", "--x_state_size", "16", "--x_stft", "--x_emb_mode", "temporal"]

