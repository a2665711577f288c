#!/bin/bash
# round 4, GPU call 29: two decode streams (the decode of batch i on stream i % 2)
cd "${GRAFT_REPO_ROOT:-.}"; O=gpurun_out/r04; mkdir -p $O
timeout 900 python -m pytest tests/test_pipeline_gpu.py -x -q 2>&1 | tail -3
run() {
  local label=$1; shift
  env "$@" timeout 400 python bench.py --no-cpu-baseline --no-strict-f32 --no-encode-cond-leg > $O/b29.json 2> $O/b29.err || tail -5 $O/b29.err
  python - "$label" <<'PY'
import json,sys
try:
    d=json.loads(open("gpurun_out/r04/b29.json").read().strip().splitlines()[-1])
    tl=d["timeline_ms"]
    print(sys.argv[1],"fps",round(d["value"],2),"stages",{k:round(v) for k,v in d["stage_ms_per_step"].items()},"in-run TF",round(d["roofline"]["in_timed_region"]["achieved"],1),
          "last d1",tl[-1]["d1"],"last t1",max(t["t1"] for t in tl),"peak GB",round(d["hbm_peak_allocated_gb"],1), flush=True)
except Exception as e: print(sys.argv[1],"failed",e, flush=True)
PY
}
run "dec streams 1" CCVS_PIPELINE_DEC_STREAMS=1
run "dec streams 2" CCVS_PIPELINE_DEC_STREAMS=2
run "dec streams 3" CCVS_PIPELINE_DEC_STREAMS=3
run "dec streams 2, 4x2" CCVS_PIPELINE_DEC_STREAMS=2 CCVS_PIPELINE_LANES=4
run "dec streams 2, 3x3" CCVS_PIPELINE_DEC_STREAMS=2 CCVS_PIPELINE_CHAINS=3
run "dec streams 1 again" CCVS_PIPELINE_DEC_STREAMS=1
run "dec streams 2 again" CCVS_PIPELINE_DEC_STREAMS=2
