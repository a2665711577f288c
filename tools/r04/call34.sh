#!/bin/bash
# round 4, GPU call 34: Drums (batch 8) with more lanes / chains
cd "${GRAFT_REPO_ROOT:-.}"; O=gpurun_out/r04; mkdir -p $O
run() {
  local label=$1; shift
  env "$@" timeout 500 python bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-strict-f32 --no-encode-cond-leg --config drums --batch 8 > $O/b34.json 2> $O/b34.err || tail -5 $O/b34.err
  python - "$label" <<'PY'
import json,sys
try:
    d=json.loads(open("gpurun_out/r04/b34.json").read().strip().splitlines()[-1])
    print(sys.argv[1],"fps",round(d["value"],2),"stages",{k:round(v) for k,v in d["stage_ms_per_step"].items()},"peak GB",round(d["hbm_peak_allocated_gb"],1), flush=True)
except Exception as e: print(sys.argv[1],"failed",e, flush=True)
PY
}
run "drums 4x2 (default)"
run "drums 8x1" CCVS_PIPELINE_LANES=8 CCVS_PIPELINE_CHAINS=1
run "drums 4x3" CCVS_PIPELINE_CHAINS=3
run "drums 2x2" CCVS_PIPELINE_LANES=2
run "drums 4x2, 1 dec" CCVS_PIPELINE_DEC_STREAMS=1
