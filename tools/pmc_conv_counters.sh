#!/bin/bash
# SQ / GRBM counters of the convolution kernel on one shape (separate --pmc passes; no other tracing domains).
# usage (on the GPU box): [SHAPE="Cin Cout k H N"] bash tools/pmc_conv_counters.sh "<counters pass 1>" "<counters pass 2>" ...
SHAPE=${SHAPE:-195 128 3 256 240}
echo "shape (Cin Cout k H N): $SHAPE"
cd /tmp && export TMPDIR=/tmp
[ $# -eq 0 ] && set -- "GRBM_GUI_ACTIVE GRBM_COUNT" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_BUSY_CYCLES" "SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES"
for c in "$@"; do
  rm -rf /tmp/pmc; rocprofv3 --kernel-trace --pmc $c -d /tmp/pmc -o p --output-format csv -- python3 $GRAFT_REPO_ROOT/tools/conv_one.py $SHAPE > /tmp/pmc.log 2>&1
  python3 - <<PY
import csv,glob,collections
f=glob.glob("/tmp/pmc/**/*counter_collection.csv",recursive=True)
if not f: print("no counter file"); print(open("/tmp/pmc.log").read()[-800:]); raise SystemExit
rows=list(csv.DictReader(open(f[0])))
agg=collections.defaultdict(list)
for r in rows:
    if "pc_kernel" in r["Kernel_Name"] or "pt_kernel" in r["Kernel_Name"]: agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k,v in agg.items(): print(k, "n=%d avg=%.4g"%(len(v), sum(v)/len(v)))
t=glob.glob("/tmp/pmc/**/*kernel_trace.csv",recursive=True)
if t:
    d=[int(r["End_Timestamp"])-int(r["Start_Timestamp"]) for r in csv.DictReader(open(t[0])) if "pc_kernel" in r["Kernel_Name"] or "pt_kernel" in r["Kernel_Name"]]
    print("kernel ms avg %.3f"%(sum(d)/len(d)/1e6))
PY
done
