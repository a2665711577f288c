#!/bin/bash
# round 4, GPU call 4: the epilogue without vector-memory waits between its stores
cd "${GRAFT_REPO_ROOT:-.}"; O=gpurun_out/r04; mkdir -p $O
timeout 900 python -m pytest tests/test_ops_gpu.py tests/test_pipeline_gpu.py tests/test_e2e_gpu.py -x -q > $O/tests5.log 2>&1; tail -3 $O/tests5.log
for shape in "195 128 3 256 120" "49 128 3 256 120" "99 128 3 256 120" "128 64 3 256 120" "64 32 3 256 120"; do
  for ab in 0 3 16387; do
    echo -n "ablate=$ab  " >> $O/ablate5.txt
    CCVS_CONV_ABLATE=$ab timeout 120 python tools/conv_one.py $shape 2>&1 | tail -1 >> $O/ablate5.txt
  done
done
cat $O/ablate5.txt
timeout 300 python tools/conv_shape_census.py 2>&1 | grep -v "amdgpu.ids\|Loading" | head -24 > $O/census5.txt; cat $O/census5.txt
CCVS_CPU_BASELINE_BUDGET=1 timeout 600 python bench.py > $O/bench5.json 2> $O/bench5.err
python - <<'PY'
import json
try:
    d=json.loads(open("gpurun_out/r04/bench5.json").read().strip().splitlines()[-1])
    print("fps",round(d["value"],2),"alone TF",round(d["roofline"]["achieved"],1),"in-run TF",round(d["roofline"]["in_timed_region"]["achieved"],1),d.get("stage_ms_per_step"), "tok step ms", d["roofline_token_loop"]["ms_per_step"])
except Exception as e: print("failed",e)
PY
