"""STFT auto-encoder, inference side (reference: models/skip_vid_generator/models/stft_model.py).

Only what the synthesis path uses: `encode` turns the spectrogram frames of a clip into the ancillary token stream
(`state_code`) that conditions the transformer in the audio-conditioned configuration (SURVEY 8f row f2,
scripts/drums/save_videos_audio_on.sh).  The decoder half (`StftDecoder`, tokens -> spectrogram) and the training
losses are outside the path and raise.
"""
import torch

from ..models.skip_autoencoder import StftEncoder
from ..modules.quantize import VectorQuantizer
from ccvs_amd.tools.utils import to_cuda
from ccvs_amd.models import load_network


class StftModel(torch.nn.Module):
    def __init__(self, opt, is_train=False, is_main=True, logger=None):
        super().__init__()
        if is_train:
            raise NotImplementedError("training is outside the MI355X hot path")
        self.opt = opt
        self.is_main = is_main
        self.initialize_networks(is_train)
        self.logger = logger if self.is_main else None

    def forward(self, data, mode='', log=False, global_iter=None):
        stft, stft_code = self.preprocess_input(data)
        if mode in ('img_encoder', 'vid_encoder'):
            return self.encode(stft)
        if mode in ('stft_reconstruction', 'eval_stft_reconstruction', 'img_decoder', 'vid_decoder'):
            raise NotImplementedError(f"mode '{mode}' (STFT decoder / losses) is outside the MI355X hot path")
        raise ValueError(f"mode '{mode}' is invalid")

    def preprocess_input(self, data):
        """stft_model.py:50-53."""
        data["stft"] = to_cuda(data, "stft")
        data["state_code"] = to_cuda(data, "state_code")
        return data["stft"], data["state_code"]

    def initialize_networks(self, is_train):
        """stft_model.py:55-66 (encoder and quantiser only)."""
        self.net_e = StftEncoder(self.opt).cuda()
        self.net_q = VectorQuantizer(self.opt.stft_num, self.opt.stft_size, beta=0.25).cuda()
        if self.is_main:
            self.net_e = load_network(self.net_e, "stft_e", self.opt)
            self.net_q = load_network(self.net_q, "stft_q", self.opt)

    @torch.no_grad()
    def encode(self, stft):
        """stft_model.py:121-125: [B,T,1,H,W] -> state_code [B, T*h*w] (bit-exact VQ indices)."""
        z = self.net_e(stft)
        _, _, info = self.net_q(z)
        return {"state_code": info[2].view(stft.shape[0], -1)}
