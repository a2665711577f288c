#!/bin/bash
# round 4, GPU call 35: Kinetics with two batches per token group (128 stacked rows)
cd "${GRAFT_REPO_ROOT:-.}"; O=gpurun_out/r04; mkdir -p $O
run() {
  local label=$1; shift
  env "$@" timeout 500 python bench.py --steps 12 --warmup 2 --no-cpu-baseline --no-strict-f32 --no-encode-cond-leg --config kinetics --batch 64 > $O/b35.json 2> $O/b35.err || tail -5 $O/b35.err
  python - "$label" <<'PY'
import json,sys
try:
    d=json.loads(open("gpurun_out/r04/b35.json").read().strip().splitlines()[-1])
    print(sys.argv[1],"fps",round(d["value"],2),"stages",{k:round(v) for k,v in d["stage_ms_per_step"].items()},"groups",d["roofline_token_loop"]["batches_per_token_group"],"peak GB",round(d["hbm_peak_allocated_gb"],1), flush=True)
except Exception as e: print(sys.argv[1],"failed",e, flush=True)
PY
}
run "kinetics 1x3 (12 batches)" CCVS_PIPELINE_CHAINS=3
run "kinetics 2x2 rows 128" CCVS_PIPELINE_MAX_ROWS=128 CCVS_PIPELINE_LANES=2 CCVS_PIPELINE_CHAINS=2
run "kinetics 2x3 rows 128" CCVS_PIPELINE_MAX_ROWS=128 CCVS_PIPELINE_LANES=2 CCVS_PIPELINE_CHAINS=3
run "kinetics 3x2 rows 192" CCVS_PIPELINE_MAX_ROWS=192 CCVS_PIPELINE_LANES=3 CCVS_PIPELINE_CHAINS=2
run "kinetics 1x4" CCVS_PIPELINE_CHAINS=4
