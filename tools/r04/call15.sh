#!/bin/bash
# round 4, GPU call 15: the whole GPU suite, smoke() and the driver's command on the final defaults
cd "${GRAFT_REPO_ROOT:-.}"; O=gpurun_out/r04; mkdir -p $O
timeout 2400 python -m pytest tests -m gpu -x -q > $O/gpu_tests_final.log 2>&1; tail -3 $O/gpu_tests_final.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
timeout 600 python bench.py > $O/b15.json 2> $O/b15.err; tail -c 600 $O/b15.json | head -c 300; echo
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r04/b15.json").read().strip().splitlines()[-1])
print("fps",round(d["value"],2),"alone TF",round(d["roofline"]["achieved"],1),"in-run",round(d["roofline"]["in_timed_region"]["achieved"],1),"traffic",d["roofline"]["traffic"],"cpu",d.get("cpu_baseline",{}).get("value"),d["config"].get("conv_intermediates"))
PY
