#!/bin/bash
# round 6, GPU call 31: after the warm-up change of bench.py / PipelinedRun -- the pipeline tests and the driver's command once more
cd /root/repo
O=gpurun_out/r06ae; mkdir -p $O
timeout 1200 python3 -m pytest tests/test_pipeline_gpu.py tests/test_host_noise_gpu.py tests/test_capi_host.py -x -q -m gpu 2>&1 | tail -4 > $O/pipeline_tests.log; cat $O/pipeline_tests.log
timeout 900 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_default.json 2> $O/bench_default.err
python3 - <<'PY'
import json
d = json.loads(open("gpurun_out/r06ae/bench_default.json").read().strip().splitlines()[-1])
r = d["roofline"]
print("fps", d["value"], "self_check", d["self_check"]["pipelined_equals_serial"], "conv alone", r["achieved"], "frac", r["frac"], "in-run", r["frac_in_run"], "single", d["single_call"]["serial_frames_per_s"], d["single_call"]["stream_frames_per_s"], "hbm", d.get("hbm_peak_allocated_gb"))
PY
