#!/bin/bash
# Round 6, GPU call 14: ONE token chain of 10 batches (160 rows per step; 20 batches = two loops) with at most N batches in decode at a
# time (their context rings are what the memory went to): CCVS_PIPELINE_MAX_DECODING.
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06m
O=gpurun_out/r06m
LEGS="--no-other-noise-leg --no-strict-f32 --no-encode-cond-leg --no-cpu-baseline"
export CCVS_PIPELINE_MEM_FRAC=1.4 CCVS_PIPELINE_MAX_ROWS=256
i=0
for CFG in "10:1:8" "10:1:6" "7:1:8"; do
  i=$((i+1))
  IFS=: read L C D <<< "$CFG"
  ( time CCVS_PIPELINE_MAX_DECODING=$D timeout 900 python bench.py --steps 20 --warmup $L --lanes $L --chains $C $LEGS ) > $O/bench_${i}_${L}x${C}_dec${D}.json 2> $O/bench_${i}.err
  tail -n 3 $O/bench_${i}.err | head -n 1
done
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r06m/bench_*.json")):
    try:
        r = json.loads([l for l in open(f) if l.startswith("{")][-1])
    except Exception as e:
        print(f, "NO LINE", e); continue
    tl = r.get("roofline_token_loop", {})
    t = r.get("timeline_ms") or []
    print(f.split("/")[-1], "fps %.1f" % r["value"], "groups", tl.get("batches_per_token_group"), "step %.3f" % tl["ms_per_step"], "stage", {k: round(v) for k, v in r["stage_ms_per_step"].items()},
          "tokens end %.2f s, run %.2f s" % (max(x["t1"] for x in t) / 1e3, max(x["d1"] for x in t) / 1e3), "self_check", (r.get("self_check") or {}).get("pipelined_equals_serial"), "hbm %.0f GB" % r["hbm_peak_allocated_gb"])
PY
