#!/bin/bash
# A/B of library builds on one box: bench.py (the driver's K and W, no side legs) per argument; "default" = the in-tree library.
#   gpurun -- bash tools/r05/lib_ab.sh tools/r05/lib_prio3.so default tools/r05/lib_prio3.so default
F="--steps 20 --warmup 5 --no-cpu-baseline --no-other-noise-leg --no-strict-f32 --no-encode-cond-leg"
for lib in "$@"; do
    if [ "$lib" = default ]; then unset CCVS_LIB; else export CCVS_LIB=$PWD/$lib; fi
    python bench.py $F > /tmp/b.json 2>/tmp/err.log
    python - "$lib" <<'PY'
import json, sys
try:
    d = json.loads(open("/tmp/b.json").read().strip().splitlines()[-1]); t = d["roofline_token_loop"]; r = d["roofline"]
    print(f"{sys.argv[1]:26s} {d['value']:7.1f} fps  step in-run {t['ms_per_step']:.3f} ms alone {t['alone']['ms_per_step']:.3f}  conv in-run {r['in_timed_region']['achieved']:.1f} alone {r['achieved']:.1f}  decode {d['stage_ms_per_step']['decode']:.0f} ms  self_check {d['self_check']['pipelined_equals_serial']}", flush=True)
except Exception as e:
    print(sys.argv[1], "FAILED", e, open("/tmp/err.log").read()[-400:])
PY
done
