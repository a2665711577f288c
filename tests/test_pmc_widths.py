"""not gpu: gfx950's FETCH_SIZE correction is applied per kernel by tools/pmc_widths.py, which asks the LIBRARY how its convolution
instantiations read (ccvs_conv_fetch_bytes_per_lane) and keeps a table for the decoder's other kernels.  Round 5's own list missed
the packed 3 x 3 instantiations (NTY = -83) and the bench line printed 0.612 GB of traffic per launch for 0.702: every kernel name
the committed rocprofv3 summaries contain must classify, an unknown name must raise, and the committed conv_traffic.json must be
the corrected reduction of the committed raw counters."""
import csv
import glob
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))


@pytest.fixture(scope="module")
def widths():
    from ccvs_amd import lib
    if not os.path.exists(lib.LIB_PATH):
        subprocess.run(["make", "-C", os.path.dirname(lib.LIB_PATH), "-j4"], check=True)
    import pmc_widths
    return pmc_widths


def profile_kernel_names():
    """Kernel names of the committed rocprofv3 summaries of the last two rounds (older rounds ran kernels with other template
    signatures -- `pc_kernel<16, 1>` of round 1 has no staging mode -- which today's library rightly does not know)."""
    names = set()
    stats = sorted(glob.glob(os.path.join(ROOT, "profiles", "r0*kernel_stats.csv")))
    last = sorted({os.path.basename(f)[:3] for f in stats})[-2:]
    for f in [f for f in stats if os.path.basename(f)[:3] in last]:
        for r in csv.DictReader(open(f)):
            names.add(r["Name"])
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "r0*conv_traffic_raw.json")))[-2:]:
        raw = json.load(open(f))
        for c in raw.values():
            names.update(c.get("per_kernel", c.get("per_kernel_KiB", {})).keys())
    return sorted(names)


def test_every_profiled_convolution_kernel_is_classified_by_the_library(widths):
    conv = [n for n in profile_kernel_names() if widths.is_conv(n) and "pack" not in n]
    assert len(conv) >= 25
    seen = set()
    for n in conv:
        b = widths.read_bytes_per_lane(n)
        assert b in (4, 16), n
        seen.add((widths.base_name(n).split("<")[0], b))
    # the instantiations round 5's list missed
    assert widths.read_bytes_per_lane("void conv2d_bf16x3_pc_kernel<32, 2, -83, 4, 1>(ConvK, uint4 const*, int, int, int)") == 16
    assert widths.fetch_scale("conv2d_bf16x3_pc_kernel<32, 1, -83, 2, 1>") == (2.0, 2.0)
    assert widths.fetch_scale("conv2d_bf16x3_pc_kernel<32, 1, -8, 2, 1>") == (2.0, 2.0)
    assert widths.fetch_scale("conv2d_bf16x3_pc_kernel<32, 4, 3, 2, 1>") == (2.0, 2.0)
    assert widths.fetch_scale("conv2d_bf16x3_pc_kernel<8, 1, 0, 2, 1>") == (1.0, 1.0)
    assert widths.fetch_scale("conv2d_bf16x3_pc_kernel<8, 1, -2, 2, 1>") == (1.0, 1.0)
    assert widths.fetch_scale("conv2d_bf16x3_kernel<32, 4, 8>") == (1.0, 1.0)
    assert ("conv2d_bf16x3_pc_kernel", 16) in seen and ("conv2d_bf16x3_kernel", 4) in seen


def test_unknown_kernels_raise(widths):
    with pytest.raises(KeyError):
        widths.read_bytes_per_lane("conv2d_bf16x3_pc_kernel<32, 2, -84, 4, 1>")     # a staging mode nobody classified
    with pytest.raises(KeyError):
        widths.read_bytes_per_lane("conv2d_next_round_kernel<32>")
    with pytest.raises(KeyError):
        widths.fetch_scale("some_new_stencil_kernel<4>")


def test_decoder_stencils_of_the_profiles_are_in_the_table(widths):
    stems = ("correlation7x7", "backwarp", "warp_fuse_blend", "warp_proj", "blur4x4_tile", "down2", "dwconvT4x4s2", "tap_shift_add", "upsample2")
    found = [n for n in profile_kernel_names() if any(s in n for s in stems)]
    assert found
    for n in found:
        assert widths.read_bytes_per_lane(n) in (4, 16, None), n
    lo, hi = widths.fetch_scale("backwarp4_kernel")                      # 8-byte gathers: calibrated on a known byte count in round 6
    assert lo == hi and 1.15 < lo < 1.25
    assert widths.fetch_scale("void blur4x4_tile_kernel(FirK, int, GridWalk)") == (2.0, 2.0)


def test_committed_traffic_is_the_corrected_reduction(widths, tmp_path):
    """profiles/conv_traffic.json (what bench.py prints as roofline.traffic) against a fresh reduction of the newest raw counters."""
    rec = json.load(open(os.path.join(ROOT, "profiles", "conv_traffic.json")))["bair-b16-bf16x3"]
    raws = sorted(glob.glob(os.path.join(ROOT, "profiles", "r0*_conv_traffic_raw.json")))
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "pmc_conv_traffic_reduce.py"), "--raw", raws[-1], "--out-dir", str(tmp_path)],
                         capture_output=True, text=True, check=True)
    fresh = json.load(open(tmp_path / "conv_traffic.json"))["bair-b16-bf16x3"]
    assert abs(fresh["bytes_per_launch"] / rec["bytes_per_launch"] - 1.0) < 1e-6, out.stdout[-400:]
    assert rec["fetch_bytes_per_launch"] > 1.85 * rec["fetch_raw_bytes_per_launch"]     # nearly every fetched byte is a 16-byte-per-lane read
    assert 0.65e9 < rec["bytes_per_launch"] < 0.76e9
