#!/bin/bash
# round 4, GPU call 7: token-group size / chains / ramp with the faster decoder (20 and 40 timed batches)
cd "${GRAFT_REPO_ROOT:-.}"; O=gpurun_out/r04; mkdir -p $O; rm -f $O/sweep7.txt
run() {  # lanes chains ramp steps
  CCVS_PIPELINE_MAX_ROWS=256 timeout 400 python bench.py --lanes $1 --chains $2 ${3:+--ramp $3} --steps $4 --no-cpu-baseline --no-strict-f32 --no-encode-cond-leg > $O/sw.json 2> $O/sw.err
  python - "$@" <<'PY' >> gpurun_out/r04/sweep7.txt
import json,sys
a=sys.argv[1:]
try:
    d=json.loads(open("gpurun_out/r04/sw.json").read().strip().splitlines()[-1])
    tl=d["timeline_ms"]
    print("lanes",a[0],"chains",a[1],"ramp",a[2] if len(a)>3 else "-","steps",a[-1],"fps",round(d["value"],2),"first decode at",round(tl[0]["d0"]),"tok step ms",round(d["roofline_token_loop"]["ms_per_step"],2),"stages",{k:round(v) for k,v in d["stage_ms_per_step"].items()},"in-run TF",round(d["roofline"]["in_timed_region"]["achieved"],1))
except Exception as e: print(a,"failed",e)
PY
  tail -1 $O/sweep7.txt
}
run 3 2 "" 20
run 4 2 "1,2" 20
run 6 1 "1,2,3" 20
run 6 2 "1,2,3,4" 20
run 8 1 "1,2,4" 20
run 3 2 "" 40
run 6 1 "1,2,3" 40
run 6 2 "1,2,3,4" 40
run 8 2 "1,2,4,6" 40
