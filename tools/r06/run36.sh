#!/bin/bash
# round 6, GPU call 36: the driver's command on the round's final commit
cd /root/repo
O=gpurun_out/r06ai; mkdir -p $O
timeout 900 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_default.json 2> $O/bench_default.err
python3 - <<'PY'
import json
d = json.loads(open("gpurun_out/r06ai/bench_default.json").read().strip().splitlines()[-1])
r = d["roofline"]
print("fps", d["value"], "self_check", d["self_check"]["pipelined_equals_serial"], "conv alone", r["achieved"], "frac", r["frac"], "in-run", r["frac_in_run"], "traffic", r["traffic"], r["traffic_over_algorithmic"],
      "single", d["single_call"]["serial_frames_per_s"], d["single_call"]["stream_frames_per_s"], "power-limited frac", r["power_limited_peak"]["frac_of_power_limited"], "cpu", d["cpu_baseline"]["value"])
PY
