"""ctypes binding of libccvs_hip.so (the C ABI declared in include/ccvs_hip.h).

The product has no CPU fallback: if the shared library is missing or a call fails this
module raises.  Build the library with `python -c "import __graft_entry__ as g; g.build()"`
or `make -C ccvs_amd/csrc`.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# (CCVS_LIB: another build of the same library, for A/B measurements of kernel changes on one GPU box)
LIB_PATH = os.environ.get("CCVS_LIB") or os.path.join(_HERE, "csrc", "libccvs_hip.so")

# every symbol include/ccvs_hip.h declares
EXPORTS = [
    "ccvs_last_error", "ccvs_abi_version", "ccvs_conv2d", "ccvs_conv2d_bf16x3", "ccvs_conv_fetch_bytes_per_lane", "ccvs_conv_persistent_tiles", "ccvs_upfirdn2d", "ccvs_dwconvT4x4s2",
    "ccvs_correlation7x7", "ccvs_backwarp", "ccvs_backwarp_ctx", "ccvs_backwarp_p8_ctx", "ccvs_backwarp_proj_ctx", "ccvs_warp_fuse_blend", "ccvs_warp_fuse_blend_ctx", "ccvs_tap_shift_add", "ccvs_vq_argmin", "ccvs_embed_gather", "ccvs_l2_normalize_channels",
    "ccvs_gpt_embed", "ccvs_layernorm", "ccvs_gemm_workspace_bytes", "ccvs_gemm_nt", "ccvs_gemm_ln", "ccvs_gemm_ln_qkv", "ccvs_attention", "ccvs_kv_append", "ccvs_sample_topk", "ccvs_sample_topk_philox", "ccvs_sample_topn",
    "ccvs_gpt_decode_step", "ccvs_gpt_decode_status", "ccvs_gpt_program_bytes", "ccvs_gpt_decode_prepare", "ccvs_pack_u8", "ccvs_pack_u8_norm", "ccvs_stream_cu_limit", "ccvs_psnr", "ccvs_ssim_workspace_bytes", "ccvs_ssim", "ccvs_resize_bilinear",
]


class ConvDesc(C.Structure):
    """Mirror of `ccvs_conv_desc`."""
    _fields_ = [
        ("N", C.c_int32), ("Cin", C.c_int32), ("Hin", C.c_int32), ("Win", C.c_int32),
        ("in_sN", C.c_int64), ("in_sC", C.c_int64),
        ("Cout", C.c_int32), ("CoutPad", C.c_int32), ("Hout", C.c_int32), ("Wout", C.c_int32),
        ("out_sN", C.c_int64), ("out_sC", C.c_int64), ("res_sN", C.c_int64), ("res_sC", C.c_int64),
        ("kh", C.c_int32), ("kw", C.c_int32), ("stride", C.c_int32), ("pad", C.c_int32), ("transposed", C.c_int32),
        ("act", C.c_int32), ("accumulate", C.c_int32), ("out_scale", C.c_float),
        ("pre", C.c_void_p), ("pre_sN", C.c_int64), ("pre_sC", C.c_int64), ("pre_div", C.c_int32),
        ("in_p8", C.c_int32), ("out_p8", C.c_int32), ("cu_limit", C.c_int32),
        ("w_ktail", C.c_void_p),
    ]


MAX_CTX = 16


class CtxList(C.Structure):
    """Mirror of `ccvs_ctx_list`."""
    _fields_ = [("k", C.c_int32), ("p", C.c_void_p * MAX_CTX), ("sN", C.c_int64 * MAX_CTX)]


class GptLayer(C.Structure):
    """Mirror of `ccvs_gpt_layer`."""
    _fields_ = [(n, C.c_void_p) for n in ("qkv_w", "qkv_b", "qkv_s", "proj_w", "proj_b", "fc_w", "fc_b", "fc_s", "fc2_w", "fc2_b",
                                          "kcache", "vcache")]


class GptDecode(C.Structure):
    """Mirror of `ccvs_gpt_decode`."""
    _fields_ = [
        ("B", C.c_int32), ("C", C.c_int32), ("H", C.c_int32), ("F", C.c_int32), ("n_layer", C.c_int32), ("Tmax", C.c_int32),
        ("vocab", C.c_int32), ("V", C.c_int32), ("ln_eps", C.c_float),
        ("layers", C.POINTER(GptLayer)),
        ("tok_emb", C.c_void_p), ("pos_table", C.c_void_p), ("pos_off", C.c_int32),
        ("head_w", C.c_void_p), ("head_b", C.c_void_p), ("head_s", C.c_void_p),
        ("tok", C.c_void_p), ("codes", C.c_void_p), ("codes_sB", C.c_int64),
        ("widx", C.c_void_p), ("len", C.c_void_p),
        ("x", C.c_void_p), ("q", C.c_void_p), ("att", C.c_void_p), ("h", C.c_void_p), ("logits", C.c_void_p),
        ("noise", C.c_void_p), ("rng", C.c_int32), ("top_k", C.c_int32), ("temperature", C.c_float),
        ("workspace", C.c_void_p), ("state", C.c_void_p), ("groups", C.c_int32),
        ("noise_stream", C.c_void_p),
        ("persistent", C.c_int32), ("program", C.c_void_p),
    ]


class CcvsError(RuntimeError):
    pass


_lib = None


def load():
    """Load the shared library once; fail loudly if it is not built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise CcvsError(f"{LIB_PATH} not found: the HIP kernel library is not built "
                        f"(run `make -C {os.path.dirname(LIB_PATH)}`); there is no CPU fallback")
    lib = C.CDLL(LIB_PATH)
    vp, i32, i64, f32 = C.c_void_p, C.c_int32, C.c_int64, C.c_float
    lib.ccvs_last_error.restype = C.c_char_p
    lib.ccvs_last_error.argtypes = []
    lib.ccvs_abi_version.restype = C.c_int
    lib.ccvs_gemm_workspace_bytes.restype = C.c_int64
    lib.ccvs_gemm_workspace_bytes.argtypes = []
    lib.ccvs_ssim_workspace_bytes.restype = C.c_int64
    lib.ccvs_ssim_workspace_bytes.argtypes = [C.c_int64, C.c_int32, C.c_int32]
    lib.ccvs_gpt_program_bytes.restype = C.c_int64
    lib.ccvs_gpt_program_bytes.argtypes = [C.c_int32]
    lib.ccvs_conv_fetch_bytes_per_lane.restype = C.c_int
    lib.ccvs_conv_fetch_bytes_per_lane.argtypes = [C.c_char_p]
    lib.ccvs_conv_persistent_tiles.restype = C.c_int          # (returns the previous mode, not a status)
    lib.ccvs_conv_persistent_tiles.argtypes = [C.c_int32]
    sigs = {
        "ccvs_conv2d": [vp, vp, vp, vp, vp, C.POINTER(ConvDesc), vp],
        "ccvs_conv2d_bf16x3": [vp, vp, vp, vp, vp, C.POINTER(ConvDesc), vp],
        "ccvs_upfirdn2d": [vp, vp, vp, i64, i32, i32, i32, i32, i32, i32, f32, i32, f32, vp],
        "ccvs_dwconvT4x4s2": [vp, i64, vp, vp, i64, i32, i32, i32, i32, vp],
        "ccvs_correlation7x7": [vp, vp, vp, i32, i32, i32, i32, i32, i32, i32, vp],
        "ccvs_backwarp": [vp, i64, i64, vp, i64, f32, vp, i64, i64, i32, i32, i32, i32, vp],
        "ccvs_warp_fuse_blend": [vp, i64, i64, vp, vp, i64, vp, i64, f32, i32, i32, i32, i32, i32, vp],
        "ccvs_backwarp_ctx": [C.POINTER(CtxList), i64, vp, i64, f32, vp, i64, i64, i32, i32, i32, i32, vp],
        "ccvs_backwarp_p8_ctx": [C.POINTER(CtxList), i64, vp, i64, f32, vp, i32, i32, i32, i32, vp],
        "ccvs_backwarp_proj_ctx": [C.POINTER(CtxList), i64, vp, i64, f32, vp, vp, vp, i32, i32, i32, i32, i32, i32, i32, vp],
        "ccvs_warp_fuse_blend_ctx": [vp, i64, i64, C.POINTER(CtxList), vp, i64, vp, i64, f32, i32, i32, i32, i32, vp],
        "ccvs_tap_shift_add": [vp, vp, vp, i64, i32, i32, i32, i32, i32, i32, vp],
        "ccvs_vq_argmin": [vp, vp, vp, vp, i32, i32, i32, i32, vp],
        "ccvs_embed_gather": [vp, vp, vp, i32, i32, i32, i32, vp],
        "ccvs_l2_normalize_channels": [vp, i32, i32, i64, vp],
        "ccvs_gpt_embed": [vp, i64, vp, i32, vp, i32, vp, vp, vp, i32, i32, i32, vp],
        "ccvs_layernorm": [vp, vp, vp, vp, i32, i32, vp],
        "ccvs_gemm_nt": [vp, i64, vp, vp, vp, vp, i64, i32, i32, i32, i32, vp, vp],
        "ccvs_gemm_ln": [vp, i64, vp, vp, vp, f32, vp, i64, i32, i32, i32, i32, vp],
        "ccvs_gemm_ln_qkv": [vp, i64, vp, vp, vp, f32, vp, vp, vp, i32, i32, i32, i32, i32, vp, i32, vp],
        "ccvs_attention": [vp, i64, i64, vp, vp, vp, i32, i32, i32, i32, vp, i32, i32, vp],
        "ccvs_kv_append": [vp, vp, i64, i64, vp, vp, i32, i32, i32, i32, vp, i32, i32, vp],
        "ccvs_sample_topk": [vp, i64, vp, vp, i64, i32, i32, i32, f32, vp],
        "ccvs_sample_topk_philox": [vp, i64, vp, i64, i32, i32, i32, f32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, vp],
        "ccvs_sample_topn": [vp, i64, vp, vp, vp, i32, i32, i32, f32, i32, vp],
        "ccvs_gpt_decode_step": [C.POINTER(GptDecode), vp],
        "ccvs_gpt_decode_status": [vp, vp],
        "ccvs_gpt_decode_prepare": [C.POINTER(GptDecode), vp],
        "ccvs_pack_u8": [vp, vp, i64, i32, i32, f32, f32, vp],
        "ccvs_pack_u8_norm": [vp, vp, i64, i32, i32, C.POINTER(C.c_float), C.POINTER(C.c_float), vp],
        "ccvs_stream_cu_limit": [vp, i32],
        "ccvs_psnr": [vp, vp, vp, i64, i64, f32, vp],
        "ccvs_ssim": [vp, vp, vp, vp, i64, i32, i32, C.c_double, vp],
        "ccvs_resize_bilinear": [vp, vp, i64, i32, i32, i32, i32, vp],
    }
    for name, argtypes in sigs.items():
        fn = getattr(lib, name)
        fn.restype = C.c_int
        fn.argtypes = argtypes
    _lib = lib
    return lib


def check(rc, name):
    if rc != 0:
        raise CcvsError(f"{name} failed ({rc}): {load().ccvs_last_error().decode()}")
