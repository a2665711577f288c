#!/bin/bash
# round 4, GPU call 33: encodes on a stream of their own; three chains with two decode streams
cd "${GRAFT_REPO_ROOT:-.}"; O=gpurun_out/r04; mkdir -p $O
CCVS_PIPELINE_ENC_STREAM=1 timeout 900 python -m pytest tests/test_pipeline_gpu.py -x -q 2>&1 | tail -2
run() {
  local label=$1; shift
  env "$@" timeout 400 python bench.py --no-cpu-baseline --no-strict-f32 --no-encode-cond-leg > $O/b33.json 2> $O/b33.err || tail -5 $O/b33.err
  python - "$label" <<'PY'
import json,sys
try:
    d=json.loads(open("gpurun_out/r04/b33.json").read().strip().splitlines()[-1])
    tl=d["timeline_ms"]
    print(sys.argv[1],"fps",round(d["value"],2),"stages",{k:round(v) for k,v in d["stage_ms_per_step"].items()},
          "last d1",max(t["d1"] for t in tl),"last t1",max(t["t1"] for t in tl),"peak GB",round(d["hbm_peak_allocated_gb"],1), flush=True)
except Exception as e: print(sys.argv[1],"failed",e, flush=True)
PY
}
run "default (enc on dec stream 0)"
run "enc stream" CCVS_PIPELINE_ENC_STREAM=1
run "4x3" CCVS_PIPELINE_CHAINS=3
run "enc stream, depth 3" CCVS_PIPELINE_ENC_STREAM=1 CCVS_PIPELINE_DEPTH=3
run "depth 1" CCVS_PIPELINE_DEPTH=1
run "default again"
run "enc stream again" CCVS_PIPELINE_ENC_STREAM=1
