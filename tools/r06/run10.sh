#!/bin/bash
# Round 6, GPU call 10: a finite run of 20 batches over two token chains is 3 + 2 groups of four -- chain A's three sequential loops
# (3 x 7.0 s) are as long as the whole run (21.06 s).  Balanced group sizes (10 batches per chain) through --ramp, same box.
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06i
O=gpurun_out/r06i
LEGS="--no-other-noise-leg --no-strict-f32 --no-encode-cond-leg --no-cpu-baseline"
i=0
for RAMP in "" "4,4,3,3,3,3" "3,3,3,3,4,4" "3,3,4,4,3,3" "" "4,4,3,3,3,3"; do
  i=$((i+1))
  if [ -z "$RAMP" ]; then R=""; else R="--ramp $RAMP"; fi
  ( time timeout 900 python bench.py --steps 20 --warmup 5 $R $LEGS ) > $O/bench_${i}_ramp_${RAMP//,/_}.json 2> $O/bench_${i}.err
done
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r06i/bench_*.json")):
    try:
        r = json.loads([l for l in open(f) if l.startswith("{")][-1])
    except Exception as e:
        print(f, "NO LINE", e); continue
    tl = r.get("roofline_token_loop", {})
    t = r.get("timeline_ms") or []
    print(f.split("/")[-1], "fps %.1f" % r["value"], "step_ms %.3f" % tl.get("ms_per_step", 0), "groups", tl.get("batches_per_token_group"),
          "stage", {k: round(v) for k, v in r["stage_ms_per_step"].items()}, "token stages end %.0f ms, run %.0f ms" % (max(x["t1"] for x in t), max(x["d1"] for x in t)), "self_check", (r.get("self_check") or {}).get("pipelined_equals_serial"))
PY
