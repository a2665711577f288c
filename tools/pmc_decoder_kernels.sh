#!/bin/bash
# Counters of the decoder's non-convolution kernels over one BAIR decode (tools/decode_only.py): per kernel name the launches,
# average duration, HBM-side bytes (FETCH_SIZE / WRITE_SIZE, KiB; FETCH_SIZE corrected per kernel by tools/pmc_widths.py: x2 for the
# kernels that read 16 bytes per lane -- gfx950 tallies such a stream at half its bytes, MI355X_MICROARCH.md "HBM" --, x1 for dword
# streams, x1.20 for the gather kernels (8-byte tap pairs: calibrated on backwarp4 with a known byte count, profiles/r06_pmc_gather_calibrate.txt)),
# LDS cycles and bank conflicts, busy cycles.  Separate --pmc passes, kernel trace only (no other tracing domains).
# usage (on the GPU box): bash tools/pmc_decoder_kernels.sh [batch]
B=${1:-16}
cd /tmp && export TMPDIR=/tmp
for c in "GRBM_GUI_ACTIVE" "FETCH_SIZE" "WRITE_SIZE" "SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES"; do
  tag=$(echo $c | tr ' ' '_')
  rm -rf /tmp/pmcd_$tag
  timeout 600 rocprofv3 --kernel-trace --pmc $c -d /tmp/pmcd_$tag -o p --output-format csv -- python3 $GRAFT_REPO_ROOT/tools/decode_only.py $B > /tmp/pmcd_$tag.log 2>&1
done
python3 - <<'PY'
import csv, glob, collections, re, os, sys
sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "tools"))
import pmc_widths
names = ("correlation7x7", "backwarp4", "warp_fuse_blend4", "warp_proj4", "blur4x4_tile", "down2", "dwconvT4x4s2", "tap_shift_add", "upsample2x2")
def short(n):
    for k in names:
        if k in n: return k
    return None
cnt = collections.defaultdict(lambda: collections.defaultdict(float))
dur = collections.defaultdict(list)
for d in glob.glob("/tmp/pmcd_*/"):
    for f in glob.glob(d + "**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = short(r["Kernel_Name"])
            if k: cnt[k][r["Counter_Name"]] += float(r["Counter_Value"])
    if "GRBM" in d:
        for f in glob.glob(d + "**/*kernel_trace.csv", recursive=True):
            for r in csv.DictReader(open(f)):
                k = short(r["Kernel_Name"])
                if k: dur[k].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
full = {}
for d in glob.glob("/tmp/pmcd_FETCH_SIZE/"):
    for f in glob.glob(d + "**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = short(r["Kernel_Name"])
            if k: full.setdefault(k, r["Kernel_Name"])
print(f"{'kernel':20s} {'launches':>8s} {'total ms':>9s} {'avg us':>8s} {'B/lane':>6s} {'fetch raw GB':>12s} {'fetch GB':>15s} {'write GB':>9s} {'(F+W)/t TB/s':>15s} {'LDS act/busy':>13s} {'conflict/LDS':>13s}")
for k in names:
    if not dur[k]: continue
    t = sum(dur[k]) / 1e9
    c = cnt[k]
    f, w = c["FETCH_SIZE"] * 1024 / 1e9, c["WRITE_SIZE"] * 1024 / 1e9
    lo, hi = pmc_widths.fetch_scale(full.get(k, k + "_kernel"))      # raises on a kernel the table does not know
    bl = pmc_widths.read_bytes_per_lane(full.get(k, k + "_kernel"))
    fs = f"{f*hi:.2f}" if lo == hi else f"{f*lo:.2f}-{f*hi:.2f}"
    bl = bl if bl else "8 g"
    rs = f"{(f*hi+w)/t/1e3:.2f}" if lo == hi else f"{(f*lo+w)/t/1e3:.2f}-{(f*hi+w)/t/1e3:.2f}"
    lds = c["SQ_LDS_IDX_ACTIVE"]; busy = c["SQ_BUSY_CYCLES"]
    print(f"{k:20s} {len(dur[k]):8d} {t*1e3:9.2f} {t/len(dur[k])*1e6:8.1f} {str(bl):>6s} {f:12.2f} {fs:>15s} {w:9.2f} {rs:>15s} {lds/busy if busy else 0:13.3f} {c['SQ_LDS_BANK_CONFLICT']/lds if lds else 0:13.3f}")
PY
