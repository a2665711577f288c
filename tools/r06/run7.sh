#!/bin/bash
# Round 6, GPU call 7: the round's final collection on the final tree (tools/collect_profiles.sh r06 quick + the corrected stencil
# table + the convolution traffic passes + the whole GPU suite).
cd $GRAFT_REPO_ROOT
bash tools/collect_profiles.sh r06 quick > gpurun_out/r06_collect.log 2>&1
timeout 600 python3 bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-strict-f32 --schedule stream > gpurun_out/r06_bench_stream.json 2>/dev/null
timeout 900 bash tools/pmc_decoder_kernels.sh > gpurun_out/r06_pmc_decoder_kernels.txt 2>&1
timeout 900 bash tools/pmc_conv_traffic.sh > gpurun_out/r06_pmc_conv_traffic.log 2>&1
cp gpurun_out/conv_traffic.json gpurun_out/r06_conv_traffic.json; cp gpurun_out/conv_traffic_raw.json gpurun_out/r06_conv_traffic_raw.json; cp gpurun_out/conv_traffic_table.txt gpurun_out/r06_conv_traffic_table.txt
( time timeout 1800 python -m pytest tests -q -m gpu ) > gpurun_out/r06_gpu_tests.log 2>&1
tail -n 5 gpurun_out/r06_gpu_tests.log
cat gpurun_out/r06_pmc_decoder_kernels.txt | tail -n 12
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r06_bench_*.json")):
    try:
        r = json.loads([l for l in open(f) if l.startswith("{")][-1])
        print(f, "fps %.1f" % r["value"], (r.get("self_check") or {}).get("pipelined_equals_serial"))
    except Exception as e:
        print(f, "NO LINE", e)
PY
