#!/bin/bash
# round 4, GPU call 8: correlation with chunk-ahead fetches, warp + projection with two channels in flight (A/B against round 3's
# library on the same box), then the whole GPU suite and the default line
cd "${GRAFT_REPO_ROOT:-.}"; O=gpurun_out/r04; mkdir -p $O
OLD=$PWD/ccvs_amd/csrc/libccvs_hip_r03.so
timeout 300 python tools/mem_bench.py 2>&1 | grep -v amdgpu.ids > $O/mem_new.txt
CCVS_LIB=$OLD timeout 300 python tools/mem_bench.py 2>&1 | grep -v amdgpu.ids > $O/mem_old.txt
paste -d'\n' $O/mem_new.txt $O/mem_old.txt
timeout 1200 python -m pytest tests -m gpu -x -q > $O/tests8.log 2>&1; tail -4 $O/tests8.log
CCVS_CPU_BASELINE_BUDGET=1 timeout 600 python bench.py > $O/bench8.json 2> $O/bench8.err
python - <<'PY'
import json
try:
    d=json.loads(open("gpurun_out/r04/bench8.json").read().strip().splitlines()[-1])
    print("fps",round(d["value"],2),"alone TF",round(d["roofline"]["achieved"],1),"in-run TF",round(d["roofline"]["in_timed_region"]["achieved"],1),d.get("stage_ms_per_step"), "cond-only", d["encode_cond_only"]["value"])
except Exception as e: print("failed",e)
PY
