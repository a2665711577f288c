#!/bin/bash
# CCVS_CONV_ABLATE variants of tools/conv_one.py under rocprofv3 --pmc GRBM_GUI_ACTIVE: CYCLES per launch (sum over the 8 XCDs / 8), which
# the chip's clock management does not distort (an ablation that feeds the matrix pipe zeros runs at a higher clock: time alone misleads).
cd /tmp && export TMPDIR=/tmp
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
for shape in "128 64 3 256 120 p8" "195 128 3 256 120" "99 128 3 256 120"; do
  for a in 0 1 128 2 3 16384 16387 4096; do
    rm -rf /tmp/pmc
    CCVS_CONV_ABLATE=$a rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE -d /tmp/pmc -o p --output-format csv -- python3 $ROOT/tools/conv_one.py $shape > /tmp/pmc.log 2>&1
    python3 - "$shape" $a <<'PY'
import csv,glob,sys
f=glob.glob("/tmp/pmc/**/*counter_collection.csv",recursive=True)
t=glob.glob("/tmp/pmc/**/*kernel_trace.csv",recursive=True)
if not f or not t: print(sys.argv[1], "ablate", sys.argv[2], "no data"); raise SystemExit
rows=[r for r in csv.DictReader(open(f[0])) if "pc_kernel" in r["Kernel_Name"] and r["Counter_Name"]=="GRBM_GUI_ACTIVE"]
cyc=[float(r["Counter_Value"])/8 for r in rows][-3:]
d=[int(r["End_Timestamp"])-int(r["Start_Timestamp"]) for r in csv.DictReader(open(t[0])) if "pc_kernel" in r["Kernel_Name"]][-3:]
c=sum(cyc)/len(cyc); ms=sum(d)/len(d)/1e6
print(f"{sys.argv[1]:24s} ablate={sys.argv[2]:6s} {c/1e6:8.2f} Mcycles  {ms:7.3f} ms  {c/ms/1e6:5.2f} GHz")
PY
  done
done
