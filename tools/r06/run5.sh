#!/bin/bash
# Round 6, GPU call 5: is the image-fastest tile order (zi) of launches with a shared pre-activation image still effective?
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06e
O=gpurun_out/r06e
cd /tmp && export TMPDIR=/tmp
for ORD in 1 0; do
  for SHAPE in "99 128 3 256 240 pre=15" "99 128 3 256 30 pre=15"; do
    rm -rf /tmp/pmc; CCVS_CONV_PRE_ORDER=$ORD timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d /tmp/pmc -o p --output-format csv -- python3 $GRAFT_REPO_ROOT/tools/conv_one.py $SHAPE > /tmp/pmc.log 2>&1
    echo "CCVS_CONV_PRE_ORDER=$ORD shape $SHAPE: $(grep TFLOP /tmp/pmc.log)"
    python3 - <<'PY'
import csv, glob
f = glob.glob("/tmp/pmc/**/*counter_collection.csv", recursive=True)
by = {}
for r in csv.DictReader(open(f[0])):
    if "conv2d_bf16x3" in r["Kernel_Name"] and r["Counter_Name"] == "FETCH_SIZE":
        by[int(r["Dispatch_Id"])] = by.get(int(r["Dispatch_Id"]), 0.0) + float(r["Counter_Value"])
print("   FETCH_SIZE x2 per launch (MB):", [round(2 * by[k] * 1024 / 1e6) for k in sorted(by)])
PY
  done
done > $GRAFT_REPO_ROOT/$O/pre_order.txt 2>&1
cat $GRAFT_REPO_ROOT/$O/pre_order.txt
