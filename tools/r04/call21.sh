#!/bin/bash
# round 4, GPU call 21: do the convolution forms that were picked alone still win beside the token loops?
cd "${GRAFT_REPO_ROOT:-.}"; O=gpurun_out/r04; mkdir -p $O
run() {
  local label=$1; shift
  env "$@" timeout 400 python bench.py --no-cpu-baseline --no-strict-f32 --no-encode-cond-leg > $O/b21.json 2> $O/b21.err || tail -5 $O/b21.err
  python - "$label" <<'PY'
import json,sys
try:
    d=json.loads(open("gpurun_out/r04/b21.json").read().strip().splitlines()[-1])
    print(sys.argv[1],"fps",round(d["value"],2),"stages",{k:round(v) for k,v in d["stage_ms_per_step"].items()},"alone TF",round(d["roofline"]["achieved"],1),"in-run TF",round(d["roofline"]["in_timed_region"]["achieved"],1), flush=True)
except Exception as e: print(sys.argv[1],"failed",e, flush=True)
PY
}
run "default"
run "WPC2=0" CCVS_CONV_WPC2=0
run "WPC2=128" CCVS_CONV_WPC2=128
run "PP4=0" CCVS_CONV_PP4=0
run "PRIO=2" CCVS_CONV_PRIO=2
run "P8=0" CCVS_CONV_P8=0
run "default again"
