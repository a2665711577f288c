#!/bin/bash
# round 6, GPU call 27: the rate of bf16 matrix instructions with nothing else busy, with LDS operand fetches, with a staging wave beside (tools/micro/mfma_rate_probe.hip),
# under tools/power_trace.py (socket power and clock while it runs)
cd /root/repo
O=gpurun_out/r06aa; mkdir -p $O
timeout 300 python3 tools/power_trace.py $O/power_probe.json -- tools/micro/mfma_rate_probe 400000 > $O/mfma_rate_probe.txt 2> $O/mfma_rate_probe.err
cat $O/mfma_rate_probe.txt; grep power_trace $O/mfma_rate_probe.err
python3 - <<'PY'
import json
d = json.load(open("gpurun_out/r06aa/power_probe.json"))
for s in d["samples"]:
    if "power_w" in s: print(s["t"], s["power_w"], s.get("sclk_mhz"))
PY
