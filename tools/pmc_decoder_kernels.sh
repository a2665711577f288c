#!/bin/bash
# Counters of the decoder's non-convolution kernels over one BAIR decode (tools/decode_only.py): per kernel name the launches,
# average duration, HBM-side bytes (FETCH_SIZE / WRITE_SIZE, KiB; fetch not corrected for the access width -- ratios only),
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
import csv, glob, collections, re
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
print(f"{'kernel':20s} {'launches':>8s} {'total ms':>9s} {'avg us':>8s} {'fetch GB':>9s} {'write GB':>9s} {'(F+W)/t TB/s':>13s} {'LDS act/busy':>13s} {'conflict/LDS':>13s}")
for k in names:
    if not dur[k]: continue
    t = sum(dur[k]) / 1e9
    c = cnt[k]
    f, w = c["FETCH_SIZE"] * 1024 / 1e9, c["WRITE_SIZE"] * 1024 / 1e9
    lds = c["SQ_LDS_IDX_ACTIVE"]; busy = c["SQ_BUSY_CYCLES"]
    print(f"{k:20s} {len(dur[k]):8d} {t*1e3:9.2f} {t/len(dur[k])*1e6:8.1f} {f:9.2f} {w:9.2f} {(f+w)/t/1e3:13.2f} {lds/busy if busy else 0:13.3f} {c['SQ_LDS_BANK_CONFLICT']/lds if lds else 0:13.3f}")
PY
