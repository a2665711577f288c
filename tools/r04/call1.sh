#!/bin/bash
# round 4, GPU call 1: the whole GPU suite, the per-shape conv census and the default bench line, new library vs round 3's
cd "${GRAFT_REPO_ROOT:-.}"; O=gpurun_out/r04; mkdir -p $O
OLD=$PWD/ccvs_amd/csrc/libccvs_hip_r03.so
timeout 900 python -m pytest tests -m gpu -x -q > $O/tests1.log 2>&1; echo "rc=$?" >> $O/tests1.log
tail -5 $O/tests1.log
timeout 300 python tools/conv_shape_census.py > $O/census_new.txt 2>&1
CCVS_LIB=$OLD timeout 300 python tools/conv_shape_census.py > $O/census_old.txt 2>&1
head -12 $O/census_new.txt; head -12 $O/census_old.txt
timeout 600 python bench.py > $O/bench_new.json 2> $O/bench_new.err
CCVS_LIB=$OLD timeout 600 python bench.py > $O/bench_old.json 2> $O/bench_old.err
python - <<'PY'
import json
for n in ("new","old"):
    try:
        d=json.loads(open(f"gpurun_out/r04/bench_{n}.json").read().strip().splitlines()[-1])
        print(n, d["value"], d["ms_per_step"], d["roofline"]["achieved"], d["roofline"].get("in_timed_region"), d.get("stage_ms_per_step"))
    except Exception as e: print(n, "failed", e)
PY
