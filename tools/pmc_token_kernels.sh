#!/bin/bash
# HBM-side bytes of the kernels of a decode step (ccvs_gpt_decode_step, launched eagerly here: the counter pass segfaults inside
# a hipGraph replay): tools/token_step_probe.py over
# the stacked rows of ROWS / 16 batches (default 48 = the bench's token group of 3), TOKENS steps from cache length 64.
# FETCH_SIZE x2 (every load of these kernels is 16 bytes per lane: the gfx950 correction of MI355X_MICROARCH.md) and WRITE_SIZE,
# KiB; separate --pmc passes, kernel trace only.  Prints per kernel: launches, average duration, GB per launch, TB/s against 8.
# usage (on the GPU box): bash tools/pmc_token_kernels.sh [rows] [tokens]
ROWS=${1:-48}; TOK=${2:-480}
cd /tmp && export TMPDIR=/tmp
export CCVS_PROBE_EAGER=1
for c in "GRBM_GUI_ACTIVE" "FETCH_SIZE" "WRITE_SIZE"; do
  rm -rf /tmp/pmct_$c
  timeout 600 rocprofv3 --kernel-trace --pmc $c -d /tmp/pmct_$c -o p --output-format csv -- python3 $GRAFT_REPO_ROOT/tools/token_step_probe.py $TOK $ROWS > /tmp/pmct_$c.log 2>&1
  grep "rows" /tmp/pmct_$c.log | tail -1
done
python3 - <<'PY'
import csv, glob, collections
names = ("gemm16_kernel", "attention_decode_kernel", "sample_topk_kernel", "gpt_embed_kernel", "gemm_seq_kernel", "attention_prefill_kernel")
def short(n):
    for k in names:
        if k in n: return k
    return None
cnt = collections.defaultdict(lambda: collections.defaultdict(float))
dur = collections.defaultdict(list)
for d in glob.glob("/tmp/pmct_*/"):
    for f in glob.glob(d + "**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = short(r["Kernel_Name"])
            if k: cnt[k][r["Counter_Name"]] += float(r["Counter_Value"])
    if "GRBM" in d:
        for f in glob.glob(d + "**/*kernel_trace.csv", recursive=True):
            for r in csv.DictReader(open(f)):
                k = short(r["Kernel_Name"])
                if k: dur[k].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
print(f"{'kernel':26s} {'launches':>8s} {'total ms':>9s} {'avg us':>8s} {'fetch x2 GB':>12s} {'write GB':>9s} {'MB/launch':>10s} {'TB/s':>6s} {'of 8 TB/s':>9s}")
tot_t = tot_b = 0.0
for k in names:
    if not dur[k]: continue
    t = sum(dur[k]) / 1e9
    c = cnt[k]
    f, w = 2 * c["FETCH_SIZE"] * 1024 / 1e9, c["WRITE_SIZE"] * 1024 / 1e9
    tot_t += t; tot_b += f + w
    print(f"{k:26s} {len(dur[k]):8d} {t*1e3:9.2f} {t/len(dur[k])*1e6:8.1f} {f:12.2f} {w:9.2f} {(f+w)/len(dur[k])*1e3:10.1f} {(f+w)/t/1e3:6.2f} {(f+w)/t/8e3:9.3f}")
print(f"all of them: {tot_b:.1f} GB in {tot_t*1e3:.1f} ms of kernel time = {tot_b/tot_t/1e3:.2f} TB/s (the counter run serialises dispatches: launch gaps are not in this time)")
PY
