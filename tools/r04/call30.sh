#!/bin/bash
# round 4, GPU call 30: lanes x chains and priorities with two decode streams
cd "${GRAFT_REPO_ROOT:-.}"; O=gpurun_out/r04; mkdir -p $O
run() {
  local label=$1; shift
  env "$@" timeout 400 python bench.py --no-cpu-baseline --no-strict-f32 --no-encode-cond-leg > $O/b30.json 2> $O/b30.err || tail -5 $O/b30.err
  python - "$label" <<'PY'
import json,sys
try:
    d=json.loads(open("gpurun_out/r04/b30.json").read().strip().splitlines()[-1])
    tl=d["timeline_ms"]
    print(sys.argv[1],"fps",round(d["value"],2),"stages",{k:round(v) for k,v in d["stage_ms_per_step"].items()},
          "last d1",max(t["d1"] for t in tl),"last t1",max(t["t1"] for t in tl),"peak GB",round(d["hbm_peak_allocated_gb"],1), flush=True)
except Exception as e: print(sys.argv[1],"failed",e, flush=True)
PY
}
export CCVS_PIPELINE_DEC_STREAMS=2
run "2 dec, 4x2" CCVS_PIPELINE_LANES=4
run "2 dec, 5x2" CCVS_PIPELINE_LANES=5 CCVS_PIPELINE_MAX_ROWS=96
run "2 dec, 6x2" CCVS_PIPELINE_LANES=6 CCVS_PIPELINE_MAX_ROWS=96
run "2 dec, 4x2 prio 0,0" CCVS_PIPELINE_LANES=4 CCVS_PIPELINE_PRIORITIES=0,0
run "2 dec, 4x2 prio -1,0" CCVS_PIPELINE_LANES=4 CCVS_PIPELINE_PRIORITIES=-1,0
run "2 dec, 6x1" CCVS_PIPELINE_LANES=6 CCVS_PIPELINE_MAX_ROWS=96 CCVS_PIPELINE_CHAINS=1
run "2 dec, 8x1" CCVS_PIPELINE_LANES=8 CCVS_PIPELINE_MAX_ROWS=128 CCVS_PIPELINE_CHAINS=1
run "2 dec, 4x2 again" CCVS_PIPELINE_LANES=4
