#!/bin/bash
# round 4, GPU call 11: packed epilogue staged pixel-major
cd "${GRAFT_REPO_ROOT:-.}"; O=gpurun_out/r04; mkdir -p $O
timeout 300 python -m pytest tests/test_ops_gpu.py -x -q -k "packed or conv" > $O/tests12a.log 2>&1; tail -3 $O/tests12a.log
CCVS_CONV_P8=1 timeout 600 python -m pytest tests/test_e2e_gpu.py tests/test_pipeline_gpu.py -x -q > $O/tests12_p8.log 2>&1; tail -3 $O/tests12_p8.log
CCVS_CONV_P8=1 timeout 300 python tools/conv_shape_census.py 2>&1 | grep -v "amdgpu.ids\|Loading" | head -14 > $O/census12_p8.txt; cat $O/census12_p8.txt
for p8 in 1 0 1; do
  CCVS_CONV_P8=$p8 timeout 400 python bench.py --no-cpu-baseline --no-strict-f32 --no-encode-cond-leg > $O/b12.json 2> $O/b12.err
  python - $p8 <<'PY'
import json,sys
try:
    d=json.loads(open("gpurun_out/r04/b12.json").read().strip().splitlines()[-1])
    print("P8",sys.argv[1],"fps",round(d["value"],2),"stages",{k:round(v) for k,v in d["stage_ms_per_step"].items()},"alone TF",round(d["roofline"]["achieved"],1),"in-run",round(d["roofline"]["in_timed_region"]["achieved"],1))
except Exception as e: print("failed",e)
PY
done
