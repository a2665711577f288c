"""Import the upstream CCVS reference on CPU (this container only).

TEST INFRASTRUCTURE.  Never imported by the product (`ccvs_amd/`), by `bench.py`
or by any `-m gpu` test: `/root/reference` does not exist on the GPU box.  It is
used by `tests/golden/make_golden.py` to produce the committed `.npz` fixtures and
by the `not gpu` test `tests/test_oracle_vs_reference.py` (skipped when the
reference tree is absent).

The reference hard-codes `.cuda()`, JIT-builds two CUDA extensions at import and
uses a cupy kernel for the cost volume.  The recipe here:

* `sys.modules` stubs for cupy / apex / torchvision / tensorboardX,
* `torch.utils.cpp_extension.load` -> dummy object (the CPU branch of
  `upfirdn2d` never touches it),
* `Tensor.cuda` / `Module.cuda` -> identity,
* `skip_autoencoder.FunctionCorrelation` -> a brute-force CPU loop restatement of
  the CUDA index math (`modules/correlation.py:44-96`); the reference has no CPU
  implementation (`correlation.py:333-334`), so this op is "parity unpinned",
* the overlapping in-place shift of `quantized_video_model.py:898,900,946`
  (rejected by torch >= 1.8) is patched at run time by wrapping
  `Tensor.__setitem__` so that an overlapping RHS view is cloned first.
"""
import os
import sys
import types
import contextlib

import torch

REF_ROOT = os.environ.get("CCVS_REFERENCE_ROOT", "/root/reference")


def reference_available():
    return os.path.isdir(os.path.join(REF_ROOT, "models", "skip_vid_generator"))


def _stub(name, **attrs):
    m = types.ModuleType(name)
    for k, v in attrs.items():
        setattr(m, k, v)
    sys.modules[name] = m
    return m


def correlation_bruteforce(first, second, stride):
    """out[b, 7*(dy+3)+(dx+3), y, x] = mean_c A[b,c,y*s,x*s] * B[b,c,y*s+dy*s,x*s+dx*s]
    with zero padding; restates `kernel_Correlation_updateOutput`
    (reference modules/correlation.py:44-96) with explicit Python loops over the
    49 displacements (vectorised over pixels only)."""
    b, c, h, w = first.shape
    s = int(stride)
    ho, wo = -(-h // s), -(-w // s)
    pad = 3 * s
    p0 = torch.zeros(b, c, h + 2 * pad, w + 2 * pad, dtype=first.dtype)
    p1 = torch.zeros_like(p0)
    p0[:, :, pad:pad + h, pad:pad + w] = first
    p1[:, :, pad:pad + h, pad:pad + w] = second
    out = torch.zeros(b, 49, ho, wo, dtype=first.dtype)
    ys = torch.arange(ho) * s + pad
    xs = torch.arange(wo) * s + pad
    a = p0[:, :, ys][:, :, :, xs]
    for ch in range(49):
        dx = (ch % 7 - 3) * s
        dy = (ch // 7 - 3) * s
        bb = p1[:, :, ys + dy][:, :, :, xs + dx]
        out[:, ch] = (a * bb).sum(dim=1) / c
    return out


_LOADED = {}


def load_reference():
    """Returns a namespace with the reference classes, importing them once."""
    if _LOADED:
        return _LOADED["ns"]
    assert reference_available(), f"reference tree not found at {REF_ROOT}"

    class _Any:
        def __init__(self, *a, **k):
            pass

        def __call__(self, *a, **k):
            return self

        def __getattr__(self, name):
            return _Any()

    cupy = _stub("cupy")
    cupy.memoize = lambda **kw: (lambda f: f)
    cupy.cuda = _Any()
    _stub("apex")
    _stub("apex.parallel", DistributedDataParallel=_Any)
    tv = _stub("torchvision")
    for sub in ["ops", "transforms", "io", "models", "utils", "datasets", "datasets.video_utils",
                "datasets.utils", "datasets.folder", "transforms.functional"]:
        m = _stub("torchvision." + sub)
        m.__getattr__ = lambda name: _Any  # type: ignore
    tv.ops = sys.modules["torchvision.ops"]
    tv.transforms = sys.modules["torchvision.transforms"]
    tv.io = sys.modules["torchvision.io"]
    tv.models = sys.modules["torchvision.models"]
    tv.utils = sys.modules["torchvision.utils"]
    tv.datasets = sys.modules["torchvision.datasets"]
    tbx = _stub("tensorboardX", SummaryWriter=_Any)

    import torch.utils.cpp_extension as cpp_ext
    cpp_ext.load = lambda *a, **k: _Any()
    torch.Tensor.cuda = lambda self, *a, **k: self
    torch.nn.Module.cuda = lambda self, *a, **k: self

    if REF_ROOT not in sys.path:
        sys.path.insert(0, REF_ROOT)
    # the reference's `data` package drags in datasets + torchvision; stub it
    data_pkg = _stub("data", create_dataset=_Any, custom_collate_fn=_Any)
    data_pkg.__path__ = [os.path.join(REF_ROOT, "data")]  # lets `data.cat` (a pure list) import

    from models.skip_vid_generator.models import skip_autoencoder as sae
    from models.skip_vid_generator.models import quantized_video_model as qvm
    from models.skip_vid_generator.models import transformer_model as tm
    from models.skip_vid_generator.models import mingpt
    from models.skip_vid_generator.modules import quantize
    try:  # ancillary-stream models; perceptual.py only needs torchvision.models to exist
        from models.skip_vid_generator.models import stft_model
    except Exception as exc:  # pragma: no cover
        stft_model = None
        print("[ref_harness] stft_model not importable:", exc)
    try:
        from models.skip_vid_generator.models import state_model
    except Exception as exc:  # pragma: no cover
        state_model = None
        print("[ref_harness] state_model not importable:", exc)
    upf = sys.modules["models.skip_vid_generator.modules.upfirdn2d"]  # the package re-exports the function under the same name
    from tools import options as ref_options

    sae.FunctionCorrelation = correlation_bruteforce

    ns = types.SimpleNamespace(sae=sae, qvm=qvm, tm=tm, mingpt=mingpt, quantize=quantize,
                               upfirdn2d=upf, options=ref_options, stft_model=stft_model, state_model=state_model)
    _LOADED["ns"] = ns
    return ns


def parse_reference_options(argv):
    """Run the reference's own argparse (tools/options.py:590-633) on `argv`."""
    ns = load_reference()
    old = sys.argv
    sys.argv = ["generator.py"] + list(argv)
    try:
        with open(os.devnull, "w") as devnull, contextlib.redirect_stdout(devnull):
            opt = ns.options.Options().parse(load_qvid_generator=True, load_transformer=True,
                                             load_state_estimator=True, load_stft_ae=True, save=False)
    finally:
        sys.argv = old
    return opt


@contextlib.contextmanager
def patched_overlapping_shift():
    """`inter[i][:, :-1] = inter[i][:, 1:]` (quantized_video_model.py:900) is an
    overlapping in-place copy that torch >= 1.8 rejects; clone the RHS."""
    orig = torch.Tensor.__setitem__

    def safe_setitem(self, idx, value):
        if isinstance(value, torch.Tensor) and value.untyped_storage().data_ptr() == self.untyped_storage().data_ptr():
            value = value.clone()
        return orig(self, idx, value)

    torch.Tensor.__setitem__ = safe_setitem
    try:
        yield
    finally:
        torch.Tensor.__setitem__ = orig


TINY_ARGV = [
    "--name", "tiny", "--dataset", "bairhd", "--max_dim", "32", "--vid_len", "4",
    "--q_z_num", "32", "--q_z_size", "16", "--q_z_shape", "8", "8",
    "--q_use_enc", "--q_use_dec", "--q_necf", "8", "--q_necf_mult", "1", "2", "2",
    "--q_enc_model", "skipgan", "--q_dec_model", "skipgan", "--q_use_inter", "--q_inter_p", "0.75",
    "--q_skip_context", "1", "2", "3", "--q_skip_memory", "3",
    "--x_z_num", "32", "--x_z_len", "256", "--x_n_layer", "2", "--x_n_head", "2", "--x_n_embd", "32",
    "--x_z_chunk", "64", "--x_cond_len", "64", "--x_emb_mode", "temporal", "--x_num_blocks", "4",
    "--batch_size_vid", "2",
]

# tiny configuration with an ancillary (STFT) token stream: 2 STFT tokens + 64 frame tokens per frame, window of 4 frames
TINY_STATE_ARGV = [a for a in TINY_ARGV] + [
    "--x_stft", "--x_state_num", "24", "--x_state_size", "2", "--x_z_len", "256", "--x_z_chunk", "66",
    "--x_top_k_state", "5", "--x_temperature_state", "0.8",
    "--a_stft_num", "24", "--a_stft_size", "16", "--a_stft_hsize", "8", "--a_stft_shape", "2", "1",
]

# tiny configuration of the state-conditioned scripts (scripts/bairhd/save_videos_state_*.sh): 2 state tokens per frame from a
# StateEstimator + scalar quantiser with 24 levels
TINY_STATEMODEL_ARGV = [a for a in TINY_ARGV] + [
    "--x_state", "--x_z_len", "256", "--x_z_chunk", "66", "--x_top_k_state", "5",
    "--s_state_size", "2", "--s_state_num", "24", "--s_state_hsize", "8",
]
