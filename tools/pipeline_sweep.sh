#!/bin/bash
# Pipelined-schedule sweep on the GPU box: frames/s for (lanes, chains, ramp) combinations of Generator.run_pipelined.
# usage: bash tools/pipeline_sweep.sh "L,C,RAMP L,C,RAMP ..." [steps]     (RAMP: sizes of the first groups joined by '+', or '-')
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT
STEPS=${2:-20}
export CCVS_BENCH_SUPERVISE=0
for combo in $1; do
  IFS=, read L C R <<< "$combo"
  RAMP=""; [ "$R" != "-" ] && RAMP="--ramp ${R//+/,}"
  timeout 900 python3 bench.py --steps $STEPS --warmup 4 --no-cpu-baseline --lanes $L --chains $C $RAMP 2>/dev/null | tail -1 > /tmp/line.json
  python3 - "$combo" <<'PY'
import json, sys
try:
    d = json.loads(open("/tmp/line.json").read())
    t = d["roofline_token_loop"]
    print(f"{sys.argv[1]:>12}: {d['value']:.1f} frames/s  ms/step {d['ms_per_step']:.0f}  stages {({k: round(v) for k, v in d['stage_ms_per_step'].items()})}  "
          f"token ms/step {t['ms_per_step']:.2f} groups {t['batches_per_token_group']} agg_frac {t['aggregate_frac']:.3f}", flush=True)
    import os
    if os.environ.get("TIMELINE") == "1":
        for i, r in enumerate(d["timeline_ms"]):
            print(f"    batch {i:2d} (group of {r['group']}): encode {r['e0']:7.0f}-{r['e1']:7.0f}  tokens {r['t0']:7.0f}-{r['t1']:7.0f}  decode {r['d0']:7.0f}-{r['d1']:7.0f}")
except Exception as e:
    print(sys.argv[1], "failed", e, flush=True)
PY
done
