#!/bin/bash
# HBM traffic of the convolution kernels over the encode + decode of one BAIR batch of 16 (the conv
# launches of one bench.py step; tools/decode_only.py avoids the hipGraph replays PMC cannot follow).
# FETCH_SIZE and WRITE_SIZE are collected in SEPARATE passes (TCC slot limit), kernel-trace only.
# Usage (GPU box): bash tools/pmc_conv_traffic.sh   -> gpurun_out/conv_traffic_raw.json
set -e
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
for C in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmc_$C
  rocprofv3 --kernel-trace --pmc $C --output-format csv -d /tmp/pmc_$C -- python3 $ROOT/tools/decode_only.py 16 > /tmp/pmc_$C.log 2>&1
done
python3 - <<PY
import csv, glob, json, collections
out = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    f = sorted(glob.glob(f"/tmp/pmc_{c}/*/*counter_collection.csv"))[-1]
    tot = collections.defaultdict(float); n = collections.defaultdict(int)
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == c and "conv2d" in r["Kernel_Name"]:
            name = r["Kernel_Name"].split("(")[0]
            tot[name] += float(r["Counter_Value"]); n[name] += 1
    out[c] = {"per_kernel_KiB": dict(tot), "launches": dict(n), "total_KiB": sum(tot.values()), "total_launches": sum(n.values())}
json.dump(out, open("$ROOT/gpurun_out/conv_traffic_raw.json", "w"), indent=1)
print(json.dumps({k: (v["total_KiB"], v["total_launches"]) for k, v in out.items()}))
PY
