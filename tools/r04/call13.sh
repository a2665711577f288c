#!/bin/bash
# round 4, GPU call 13: everything with the defaults of the round (packed activations on): the whole GPU suite, the driver's
# command, the serial line
cd "${GRAFT_REPO_ROOT:-.}"; O=gpurun_out/r04; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -x -q > $O/tests13.log 2>&1; tail -4 $O/tests13.log
timeout 900 python bench.py > $O/bench13_default.json 2> $O/bench13_default.err
python - <<'PY'
import json
try:
    d=json.loads(open("gpurun_out/r04/bench13_default.json").read().strip().splitlines()[-1])
    print("fps",round(d["value"],2),"ms/step",round(d["ms_per_step"],1),"alone TF",round(d["roofline"]["achieved"],1),"in-run TF",round(d["roofline"]["in_timed_region"]["achieved"],1),d.get("stage_ms_per_step"), "cond-only", round(d["encode_cond_only"]["value"],2), "strict", d.get("strict_f32",{}).get("value"), "cpu", d["cpu_baseline"]["value"], d.get("supervisor"))
except Exception as e: print("failed",e)
PY
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
