#!/usr/bin/env python3
"""gfx950's FETCH_SIZE correction, per kernel of this library -- ONE place for every HBM-counter script under tools/.

MI355X_MICROARCH.md, "HBM": on gfx950 FETCH_SIZE (= TCC_EA0_RDREQ x 64 B) reports exactly HALF the bytes of a wide coalesced
streaming read (16 bytes per lane: global_load_dwordx4 and buffer_load ... lds alike); WRITE_SIZE is exact for 16-byte-per-lane
stores; "other access widths are uncalibrated".  So a raw FETCH_SIZE must be doubled for the kernels that read 16 bytes per lane,
left alone for dword streams; the gather kernels' pattern was calibrated on a known byte count in round 6 (x 1.20, below).

Round 5's tools/pmc_conv_traffic.sh kept its own list of "wide" instantiations (NTY in {1, 3, -8}) and missed the packed 3 x 3 form
NTY = -83 -- 42 % of all raw convolution fetch -- so the bench line printed 0.612 GB per launch where the corrected counters say
0.70 ("What's weak" 4).  Now:
  * convolution kernels are classified by the LIBRARY (ccvs_conv_fetch_bytes_per_lane, include/ccvs_hip.h; the kernel header's
    own staging-mode predicates, conv_common.h: an instantiation with an unclassified staging mode does not compile);
  * the decoder's other kernels by the table below (read off their load instructions, flow.hip / resample.hip);
  * a kernel name nobody classifies raises -- a new kernel cannot silently fall into the x1 bucket.
`tests/test_pmc_widths.py` runs the classifier over every kernel name in profiles/*kernel_stats.csv (no GPU)."""
import ctypes
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_LIB = None

# access width of the dominant READ stream of the non-convolution kernels (bytes per lane; None = the gather pattern: x 1.20, calibrated)
STENCIL_READ_BYTES = {
    "blur4x4_tile_kernel": 16,        # resample.hip: F32Quad rows into the LDS tile
    "down2_tile_kernel": 16, "down2_kernel": 16,
    "dwconvT4x4s2x4_kernel": 16,      # aligned 16-byte row loads
    "tap_shift_add4_kernel": 16,
    "upsample2x2_kernel": 4, "upsample2_kernel": 4, "upfirdn2d_generic_kernel": 4, "dwconvT4x4s2x2_kernel": 4, "dwconvT4x4s2_kernel": 4,
    "tap_shift_add_kernel": 4, "backwarp_kernel": 4, "warp_proj_kernel": 4, "warp_fuse_blend_kernel": 4,
    "correlation7x7x2_kernel": 4, "correlation7x7_kernel": 4,   # halo elements and `first` values dword by dword
    "backwarp4_kernel": None,         # flow rows 16 B, the bilinear taps as 8-byte pairs from gathered addresses
    "backwarp_p8_kernel": None, "warp_proj4_kernel": None, "warp_fuse_blend4_kernel": None,
}


def _lib():
    global _LIB
    if _LIB is None:
        path = os.environ.get("CCVS_LIB") or os.path.join(ROOT, "ccvs_amd", "csrc", "libccvs_hip.so")
        _LIB = ctypes.CDLL(path)
        _LIB.ccvs_conv_fetch_bytes_per_lane.restype = ctypes.c_int
        _LIB.ccvs_conv_fetch_bytes_per_lane.argtypes = [ctypes.c_char_p]
    return _LIB


def base_name(kernel_name):
    """`void conv2d_bf16x3_pc_kernel<32, 2, -83, 4, 1>(ConvK, ...)` -> `conv2d_bf16x3_pc_kernel<32, 2, -83, 4, 1>`."""
    name = kernel_name.strip().strip('"')
    name = re.sub(r"^void\s+", "", name)
    depth, out = 0, []
    for ch in name:           # cut at the argument list: the first '(' outside the template brackets
        if ch == "<":
            depth += 1
        elif ch == ">":
            depth -= 1
        elif ch == "(" and depth == 0:
            break
        out.append(ch)
    return "".join(out).strip()


def is_conv(kernel_name):
    return "conv2d_" in kernel_name


def read_bytes_per_lane(kernel_name):
    """16, 4, or None (mixed widths: uncalibrated) for a kernel of this library; raises KeyError for a name nobody classifies."""
    name = base_name(kernel_name)
    if is_conv(name):
        b = _lib().ccvs_conv_fetch_bytes_per_lane(name.encode())
        if b not in (4, 16):
            raise KeyError(f"convolution kernel `{name}` is not classified by ccvs_conv_fetch_bytes_per_lane (conv_common.h: conv_nty_*)")
        return b
    stem = re.sub(r"<.*$", "", name)
    if stem not in STENCIL_READ_BYTES:
        raise KeyError(f"kernel `{name}` has no entry in tools/pmc_widths.py:STENCIL_READ_BYTES")
    return STENCIL_READ_BYTES[stem]


# Round 6 calibrated the gather pattern itself (tools/r06/gather_calibrate.py, profiles/r06_pmc_gather_calibrate.txt): backwarp4 on 48 x 96
# planes of 256 x 256 with sub-pixel flows needs 1233.1 MB of reads per launch (every source byte once + the flows) and FETCH_SIZE says
# 1026.6 MB -- 0.833 of the bytes (the 8-byte tap pairs are tallied in full, the 16-byte flow rows and whatever the L2 turns into wide
# requests at half); WRITE_SIZE is exact (1208.0 = 1208.0 MB).  The kernels of that pattern take x 1.20 instead of a [x1, x2] bracket.
GATHER_FETCH_SCALE = 1233.1 / 1026.6


def fetch_scale(kernel_name):
    """(low, high) multipliers of the raw FETCH_SIZE of this kernel: (2, 2) for 16-byte-per-lane streams, (1, 1) for dword streams,
    (1.2, 1.2) for the gather kernels (8-byte tap pairs from gathered addresses: calibrated on backwarp4, see above)."""
    b = read_bytes_per_lane(kernel_name)
    return (2.0, 2.0) if b == 16 else ((1.0, 1.0) if b == 4 else (GATHER_FETCH_SCALE, GATHER_FETCH_SCALE))


if __name__ == "__main__":
    import sys
    for n in sys.argv[1:]:
        print(n, "->", read_bytes_per_lane(n), fetch_scale(n))
