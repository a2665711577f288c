#!/bin/bash
# Round 6, GPU call 12: the tail of the finite run -- chain A's third group (6.0 s) runs alone while chain B idles and the decoder starves.
# Larger LAST groups instead (two groups per chain, the second one bigger): tokens end earlier, the decoder finishes without token loops beside it.
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06k
O=gpurun_out/r06k
LEGS="--no-other-noise-leg --no-strict-f32 --no-encode-cond-leg --no-cpu-baseline"
export CCVS_PIPELINE_MEM_FRAC=1.4
i=0
for CFG in "4:" "8:4,4,4,8" "6:4,4,6,6" "7:3,3,7,7" "8:4,4,4,8"; do
  i=$((i+1))
  L=${CFG%%:*}; RAMP=${CFG#*:}
  if [ -z "$RAMP" ]; then R=""; else R="--ramp $RAMP"; fi
  ( time timeout 900 python bench.py --steps 20 --warmup 6 --lanes $L --chains 2 $R $LEGS ) > $O/bench_${i}_lanes${L}_ramp_${RAMP//,/_}.json 2> $O/bench_${i}.err
  tail -n 2 $O/bench_${i}.err | head -n 1
done
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r06k/bench_*.json")):
    try:
        r = json.loads([l for l in open(f) if l.startswith("{")][-1])
    except Exception as e:
        print(f, "NO LINE", e); continue
    tl = r.get("roofline_token_loop", {})
    t = r.get("timeline_ms") or []
    print(f.split("/")[-1], "fps %.1f" % r["value"], "groups", tl.get("batches_per_token_group"), "stage", {k: round(v) for k, v in r["stage_ms_per_step"].items()},
          "tokens end %.2f s, run %.2f s" % (max(x["t1"] for x in t) / 1e3, max(x["d1"] for x in t) / 1e3), "self_check", (r.get("self_check") or {}).get("pipelined_equals_serial"), "hbm %.0f GB" % r["hbm_peak_allocated_gb"])
PY
