#!/bin/bash
# Round 6, GPU call 13: the timeline of 5 lanes x 2 chains (20 batches = two groups of five per chain).
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06l
O=gpurun_out/r06l
LEGS="--no-other-noise-leg --no-strict-f32 --no-encode-cond-leg --no-cpu-baseline"
( time timeout 900 python bench.py --steps 20 --warmup 5 --lanes 5 --chains 2 $LEGS ) > $O/bench_5x2.json 2> $O/bench_5x2.err
python - <<'PY'
import json
r = json.loads([l for l in open("gpurun_out/r06l/bench_5x2.json") if l.startswith("{")][-1])
t = r["timeline_ms"]
print("fps %.1f" % r["value"], "groups", r["roofline_token_loop"]["batches_per_token_group"], "step %.3f" % r["roofline_token_loop"]["ms_per_step"], "hbm %.0f" % r["hbm_peak_allocated_gb"], "stage", {k: round(v) for k, v in r["stage_ms_per_step"].items()})
for i, x in enumerate(t):
    print(i, x)
PY
