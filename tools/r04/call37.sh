#!/bin/bash
# round 4, GPU call 37: cache policy of the decode step's once-read streams on the final schedule (bit 0: keys / values, bit 1: weights)
cd "${GRAFT_REPO_ROOT:-.}"; O=gpurun_out/r04; mkdir -p $O
run() {
  local label=$1; shift
  env "$@" timeout 400 python bench.py --no-cpu-baseline --no-strict-f32 --no-encode-cond-leg > $O/b37.json 2> $O/b37.err || tail -5 $O/b37.err
  python - "$label" <<'PY'
import json,sys
try:
    d=json.loads(open("gpurun_out/r04/b37.json").read().strip().splitlines()[-1])
    tl=d["timeline_ms"]
    print(sys.argv[1],"fps",round(d["value"],2),"stages",{k:round(v) for k,v in d["stage_ms_per_step"].items()},"last d1",max(t["d1"] for t in tl),"last t1",max(t["t1"] for t in tl), flush=True)
except Exception as e: print(sys.argv[1],"failed",e, flush=True)
PY
}
run "DECODE_NT=1 (default)"
run "DECODE_NT=0" CCVS_DECODE_NT=0
run "DECODE_NT=3" CCVS_DECODE_NT=3
run "DECODE_NT=1 again"
