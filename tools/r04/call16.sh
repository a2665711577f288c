#!/bin/bash
# round 4, GPU call 16: the decoder takes a batch's tokens frame by frame while its token loop runs (CCVS_PIPELINE_STREAM)
cd "${GRAFT_REPO_ROOT:-.}"; O=gpurun_out/r04; mkdir -p $O
timeout 900 python -m pytest tests/test_pipeline_gpu.py -x -q > $O/tests16.log 2>&1; tail -5 $O/tests16.log
run() {  # label, env...
  local label=$1; shift
  env "$@" timeout 400 python bench.py --no-cpu-baseline --no-strict-f32 --no-encode-cond-leg > $O/b16.json 2> $O/b16.err || tail -5 $O/b16.err
  python - "$label" <<'PY'
import json,sys
try:
    d=json.loads(open("gpurun_out/r04/b16.json").read().strip().splitlines()[-1])
    tl=d["timeline_ms"]
    print(sys.argv[1],"fps",round(d["value"],2),"stages",{k:round(v) for k,v in d["stage_ms_per_step"].items()},"in-run TF",round(d["roofline"]["in_timed_region"]["achieved"],1),
          "first d0",tl[0]["d0"],"last d1",tl[-1]["d1"],"last t1",max(t["t1"] for t in tl),"peak GB",round(d["hbm_peak_allocated_gb"],1), flush=True)
except Exception as e: print(sys.argv[1],"failed",e, flush=True)
PY
}
run "stream0 3x2" CCVS_PIPELINE_STREAM=0
run "stream1 3x2" CCVS_PIPELINE_STREAM=1
run "stream1 4x2" CCVS_PIPELINE_STREAM=1 CCVS_PIPELINE_LANES=4
run "stream1 5x2" CCVS_PIPELINE_STREAM=1 CCVS_PIPELINE_LANES=5 CCVS_PIPELINE_MAX_ROWS=96
run "stream1 3x3" CCVS_PIPELINE_STREAM=1 CCVS_PIPELINE_CHAINS=3
run "stream1 4x3" CCVS_PIPELINE_STREAM=1 CCVS_PIPELINE_LANES=4 CCVS_PIPELINE_CHAINS=3
run "stream1 6x1" CCVS_PIPELINE_STREAM=1 CCVS_PIPELINE_LANES=6 CCVS_PIPELINE_MAX_ROWS=96 CCVS_PIPELINE_CHAINS=1
run "stream1 3x2 depth3" CCVS_PIPELINE_STREAM=1 CCVS_PIPELINE_DEPTH=3
run "stream1 3x2 depth1" CCVS_PIPELINE_STREAM=1 CCVS_PIPELINE_DEPTH=1
run "stream1 3x2 again" CCVS_PIPELINE_STREAM=1
