"""StateModel: the ancillary *state* token stream of the state-conditioned configurations
(scripts/bairhd/save_videos_state_{on,off}.sh): a StateEstimator regresses a low-dimensional state (the
robot-arm position) from the quantised latent map of every frame, and a scalar VectorQuantizer
(e_dim = 1) turns each state coordinate into a token.

Host-side mirror of the inference half of the reference's
`models/skip_vid_generator/models/state_model.py` (forward mode dispatch :22-46, preprocess_input :48-52,
initialize_networks :54-62, encode :109-117, decode :119-124): same constructor, modes, dict keys,
`ValueError` on an unknown mode; the training modes raise.
"""
import torch

from ..models.skip_autoencoder import StateEstimator
from ..modules.quantize import VectorQuantizer
from ccvs_amd.tools.utils import to_cuda
from ccvs_amd.models import load_network


class StateModel(torch.nn.Module):
    def __init__(self, opt, is_train=False, is_main=True, logger=None):
        super().__init__()
        if is_train:
            raise NotImplementedError("training is outside the MI355X hot path")
        self.opt = opt
        self.is_main = is_main
        self.initialize_networks(is_train)
        self.logger = logger if self.is_main else None

    def forward(self, data, mode='', log=False, global_iter=None):
        if mode in ('state_estimator', 'eval_state_estimator'):
            raise NotImplementedError(f"mode '{mode}' (training loss) is outside the MI355X hot path")
        if mode not in ('img_encoder', 'vid_encoder', 'img_decoder', 'vid_decoder'):
            raise ValueError(f"mode '{mode}' is invalid")
        z, state, state_code = self.preprocess_input(data)
        if mode in ('img_encoder', 'vid_encoder'):
            return self.encode(state, z)
        return self.decode(state_code, "img" if mode == 'img_decoder' else "vid")

    def preprocess_input(self, data):
        """state_model.py:48-52."""
        data["z"] = to_cuda(data, "z")
        data["state"] = to_cuda(data, "state")
        data["state_code"] = to_cuda(data, "state_code")
        return data["z"], data["state"], data["state_code"]

    def initialize_networks(self, is_train):
        """state_model.py:54-62."""
        self.net_s = StateEstimator(self.opt).cuda() if not getattr(self.opt, "quantize_only", False) else None
        self.net_q = VectorQuantizer(self.opt.state_num, 1, beta=0.25).cuda()
        if self.is_main:
            self.net_s = load_network(self.net_s, "state_s", self.opt) if self.net_s is not None else None
            self.net_q = load_network(self.net_q, "state_q", self.opt)

    @torch.no_grad()
    def encode(self, state, z):
        """state_model.py:109-117: estimate the state from z unless one is given, quantise each coordinate."""
        if 0 in state.size():
            if self.net_s is None:
                raise ValueError("quantize_only StateModel needs data['state']")
            state = self.net_s(z)
        _, _, info = self.net_q(state.contiguous())
        return {"state_code": info[2].view(state.shape[0], -1)}

    @torch.no_grad()
    def decode(self, state_code, dtype):
        """state_model.py:119-124."""
        shape = [self.opt.state_size] if dtype == "img" else [-1, self.opt.state_size]
        state = self.net_q.embed_code(state_code)
        return {"state": state.view(state_code.size(0), *shape)}
