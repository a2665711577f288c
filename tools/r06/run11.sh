#!/bin/bash
# Round 6, GPU call 11: the driver's command on the fixed bench.py (the timed run's token groups in the line), the touched tests.
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06j
O=gpurun_out/r06j
( time timeout 900 python -m pytest tests/test_pipeline_gpu.py tests/test_host_noise_gpu.py -x -q ) > $O/tests_pipeline.log 2>&1
tail -n 3 $O/tests_pipeline.log
( time timeout 1500 python bench.py --gpus 1 --steps 20 --warmup 5 ) > $O/bench_default.json 2> $O/bench_default.err
python - <<'PY'
import json
r = json.loads([l for l in open("gpurun_out/r06j/bench_default.json") if l.startswith("{")][-1])
tl = r["roofline_token_loop"]
print("fps %.1f" % r["value"], "self_check", r["self_check"]["pipelined_equals_serial"], "groups", tl["batches_per_token_group"], "step %.3f ms alone %.3f" % (tl["ms_per_step"], tl["alone"]["ms_per_step"]),
      "chains", tl["concurrent_token_loops"], "gemm16 alone rows", r["roofline_gemm16"]["alone"]["rows"], "%.1f us" % r["roofline_gemm16"]["alone"]["avg_launch_us"])
print(r["config"]["schedule"][:200])
print("single_call", {k: v for k, v in r["single_call"].items() if k != "note"})
PY
