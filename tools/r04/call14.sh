#!/bin/bash
# round 4, GPU call 14: back-warp into the packed input of the first Subpixel convolution
cd "${GRAFT_REPO_ROOT:-.}"; O=gpurun_out/r04; mkdir -p $O
timeout 300 python -m pytest tests/test_ops_gpu.py -x -q -k "backwarp or warp or packed" > $O/tests14a.log 2>&1; tail -3 $O/tests14a.log
timeout 900 python -m pytest tests/test_e2e_gpu.py tests/test_pipeline_gpu.py tests/test_real_geometry_gpu.py -x -q -k "not gpt" > $O/tests14b.log 2>&1; tail -3 $O/tests14b.log
for pw in 1 0; do
  CCVS_P8_WARP=$pw timeout 300 python tools/conv_shape_census.py 2>&1 | grep -v "amdgpu.ids\|Loading" | head -8 > $O/census14_pw$pw.txt; cat $O/census14_pw$pw.txt
done
for pw in 1 0 1; do
  CCVS_P8_WARP=$pw timeout 400 python bench.py --no-cpu-baseline --no-strict-f32 --no-encode-cond-leg > $O/b14.json 2> $O/b14.err
  python - $pw <<'PY'
import json,sys
try:
    d=json.loads(open("gpurun_out/r04/b14.json").read().strip().splitlines()[-1])
    print("P8_WARP",sys.argv[1],"fps",round(d["value"],2),"stages",{k:round(v) for k,v in d["stage_ms_per_step"].items()},"alone TF",round(d["roofline"]["achieved"],1),"in-run",round(d["roofline"]["in_timed_region"]["achieved"],1))
except Exception as e: print("failed",e)
PY
done
