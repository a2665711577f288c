#!/bin/bash
# Round 6, GPU call 15: the whole GPU suite, smoke() and the driver's command on the round's last tree.
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06n
O=gpurun_out/r06n
( time timeout 1800 python -m pytest tests -q -m gpu ) > $O/gpu_tests.log 2>&1
tail -n 4 $O/gpu_tests.log
( time timeout 600 python -c "import __graft_entry__ as g; g.smoke()" ) > $O/smoke.log 2>&1
tail -n 4 $O/smoke.log
( time timeout 1500 python bench.py --gpus 1 --steps 20 --warmup 5 ) > $O/bench_default.json 2> $O/bench_default.err
python - <<'PY'
import json
r = json.loads([l for l in open("gpurun_out/r06n/bench_default.json") if l.startswith("{")][-1])
tl = r["roofline_token_loop"]
print("fps %.1f" % r["value"], "self_check", r["self_check"]["pipelined_equals_serial"], "groups", tl["batches_per_token_group"], "step %.3f ms alone %.3f" % (tl["ms_per_step"], tl["alone"]["ms_per_step"]), "hbm %.0f GB" % r["hbm_peak_allocated_gb"])
print("single_call", {k: v for k, v in r["single_call"].items() if k != "note"})
PY
