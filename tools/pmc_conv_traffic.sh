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
import csv, glob, json, collections, re
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
# MI355X_MICROARCH.md, HBM section: on gfx950 FETCH_SIZE reports HALF the bytes of 16-byte-per-lane streaming reads
# (global_load_dwordx4 and LDS-DMA alike); WRITE_SIZE is exact.  The instantiations <TW, MB, 1|3|-8> stage activations
# with aligned dwordx4 loads / LDS-DMA and weights by LDS-DMA: x2; the scalar-staging ones (<.., 0|-2>, the synchronous
# kernel) read dwords: x1.
def wide(name):
    m = re.search(r"pc_kernel<\\s*\\d+,\\s*\\d+,\\s*(-?\\d+)(?:,\\s*\\d+)*>", name)
    return bool(m) and int(m.group(1)) in (1, 3, -8)
fetch = sum(v * (2.0 if wide(k) else 1.0) for k, v in out["FETCH_SIZE"]["per_kernel_KiB"].items()) * 1024
write = out["WRITE_SIZE"]["total_KiB"] * 1024
launches = out["FETCH_SIZE"]["total_launches"]
summary = {"bair-b16-bf16x3": {
    "bytes_per_launch": (fetch + write) / launches, "fetch_bytes_per_launch": fetch / launches, "write_bytes_per_launch": write / launches,
    "launches": launches,
    "source": "rocprofv3 --kernel-trace --pmc FETCH_SIZE / WRITE_SIZE (separate passes) -- python3 tools/decode_only.py 16 (tools/pmc_conv_traffic.sh); "
              "KiB counters x1024 summed over the conv launches of one batch; FETCH_SIZE x2 for the instantiations that read 16 bytes per lane "
              "(aligned dwordx4 / LDS-DMA staging), x1 for the dword-staging ones, per the gfx950 correction of MI355X_MICROARCH.md"}}
json.dump(summary, open("$ROOT/gpurun_out/conv_traffic.json", "w"), indent=1)
print(json.dumps(summary["bair-b16-bf16x3"], indent=1)[:600])
PY
