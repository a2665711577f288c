#!/bin/bash
# Round 6, GPU call 6: the whole GPU suite with generate_vid streaming by default, smoke(), the driver's bench command.
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06f
O=gpurun_out/r06f
( time timeout 1800 python -m pytest tests -x -q -m gpu ) > $O/gpu_tests.log 2>&1
tail -n 6 $O/gpu_tests.log
( time timeout 600 python -c "import __graft_entry__ as g; g.smoke()" ) > $O/smoke.log 2>&1
tail -n 5 $O/smoke.log
( time timeout 1500 python bench.py --gpus 1 --steps 20 --warmup 5 ) > $O/bench_default.json 2> $O/bench_default.err
tail -n 4 $O/bench_default.err
python - <<'PY'
import json
r = json.loads([l for l in open("gpurun_out/r06f/bench_default.json") if l.startswith("{")][-1])
print("fps %.1f" % r["value"], "self_check", r["self_check"]["pipelined_equals_serial"], "single_call", {k: v for k, v in r["single_call"].items() if k != "note"})
print("roofline frac %.4f in_run %.4f traffic %.0f x%.3f" % (r["roofline"]["frac"], r["roofline"]["frac_in_run"], r["roofline"]["traffic"], r["roofline"]["traffic_over_algorithmic"]))
print("strict_f32 %.1f  device noise %.1f  encode_cond %.1f  cpu %.4f" % (r["strict_f32"]["frames_per_s"], r["sampling_device_noise"]["value"], r["encode_cond_only"]["value"], r["cpu_baseline"]["value"]))
PY
