#!/bin/bash
# round 4, GPU call 17: stream priorities now that the decoder follows the token loops frame by frame
cd "${GRAFT_REPO_ROOT:-.}"; O=gpurun_out/r04; mkdir -p $O
run() {  # label, env...
  local label=$1; shift
  env "$@" timeout 400 python bench.py --no-cpu-baseline --no-strict-f32 --no-encode-cond-leg > $O/b17.json 2> $O/b17.err || tail -5 $O/b17.err
  python - "$label" <<'PY'
import json,sys
try:
    d=json.loads(open("gpurun_out/r04/b17.json").read().strip().splitlines()[-1])
    tl=d["timeline_ms"]
    print(sys.argv[1],"fps",round(d["value"],2),"stages",{k:round(v) for k,v in d["stage_ms_per_step"].items()},"in-run TF",round(d["roofline"]["in_timed_region"]["achieved"],1),
          "first d0",tl[0]["d0"],"last d1",tl[-1]["d1"],"last t1",max(t["t1"] for t in tl),"peak GB",round(d["hbm_peak_allocated_gb"],1), flush=True)
except Exception as e: print(sys.argv[1],"failed",e, flush=True)
PY
}
run "prio -1,0 3x2" CCVS_PIPELINE_PRIORITIES=-1,0
run "prio 0,0 3x2" CCVS_PIPELINE_PRIORITIES=0,0
run "prio 0,-1 3x2" CCVS_PIPELINE_PRIORITIES=0,-1
run "prio 0,0 4x2" CCVS_PIPELINE_PRIORITIES=0,0 CCVS_PIPELINE_LANES=4
run "prio 0,-1 4x2" CCVS_PIPELINE_PRIORITIES=0,-1 CCVS_PIPELINE_LANES=4
run "prio 0,-1 6x1" CCVS_PIPELINE_PRIORITIES=0,-1 CCVS_PIPELINE_LANES=6 CCVS_PIPELINE_MAX_ROWS=96 CCVS_PIPELINE_CHAINS=1
run "prio -1,0 3x2 again" CCVS_PIPELINE_PRIORITIES=-1,0
