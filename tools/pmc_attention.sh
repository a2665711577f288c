#!/bin/bash
# MFMA utilisation of the prefill attention kernel: SQ_VALU_MFMA_BUSY_CYCLES (cycles the matrix pipe of a SIMD is busy, summed
# over SIMDs) against GRBM_GUI_ACTIVE (summed over the 8 XCDs) x 1024 SIMDs / 8 -- separate --pmc passes, kernel trace only.
# usage (GPU box): bash tools/pmc_attention.sh
cd /tmp && export TMPDIR=/tmp
for c in "GRBM_GUI_ACTIVE" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU"; do
  rm -rf /tmp/pmca; rocprofv3 --kernel-trace --pmc $c -d /tmp/pmca -o p --output-format csv -- python3 $GRAFT_REPO_ROOT/tools/attn_prefill_one.py 16 1024 5 > /tmp/pmca.log 2>&1
  python3 - <<PY
import csv,glob,collections
f=glob.glob("/tmp/pmca/**/*counter_collection.csv",recursive=True)
if not f: print("no counter file"); print(open("/tmp/pmca.log").read()[-800:]); raise SystemExit
agg=collections.defaultdict(list)
for r in csv.DictReader(open(f[0])):
    if "attention_prefill" in r["Kernel_Name"]: agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k,v in agg.items(): print(k, "launches=%d avg=%.6g"%(len(v), sum(v)/len(v)))
t=glob.glob("/tmp/pmca/**/*kernel_trace.csv",recursive=True)
if t:
    d=[int(r["End_Timestamp"])-int(r["Start_Timestamp"]) for r in csv.DictReader(open(t[0])) if "attention_prefill" in r["Kernel_Name"]]
    print("kernel ms avg %.4f over %d launches"%(sum(d)/len(d)/1e6, len(d)))
PY
done
tail -1 /tmp/pmca.log
