#!/bin/bash
# round 6, GPU call 24: the tree after the persistent-tile work (switch off by default): smoke, the whole GPU suite, the driver's command
cd /root/repo
O=gpurun_out/r06x; mkdir -p $O
timeout 600 python3 -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?" >> $O/smoke.log; tail -3 $O/smoke.log
timeout 1500 python3 -m pytest tests -x -q -m gpu 2>&1 | tail -6 > $O/gpu_tests.log; cat $O/gpu_tests.log
timeout 900 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_default.json 2> $O/bench_default.err
python3 - <<'PY'
import json
d = json.loads(open("gpurun_out/r06x/bench_default.json").read().strip().splitlines()[-1])
r = d["roofline"]
print("fps", d["value"], "self_check", d["self_check"]["pipelined_equals_serial"], "conv alone", r["achieved"], "frac", r["frac"], "in-run", r["frac_in_run"], "single", d["single_call"]["serial_frames_per_s"], d["single_call"]["stream_frames_per_s"])
PY
