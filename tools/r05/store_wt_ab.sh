#!/bin/bash
# Convolution epilogue stores write-through (sc1) against plain: same box, bench.py without its side legs.
#   build: cp -r ccvs_amd/csrc /tmp/wt && make -C /tmp/wt ROOT=$PWD EXTRA=-DCB_STORE_WT && cp /tmp/wt/libccvs_hip.so tools/r05/lib_wt.so
F="--steps 20 --warmup 5 --no-cpu-baseline --no-other-noise-leg --no-strict-f32 --no-encode-cond-leg"
for lib in tools/r05/lib_wt.so "" tools/r05/lib_wt.so ""; do
    if [ -n "$lib" ]; then export CCVS_LIB=$PWD/$lib; else unset CCVS_LIB; fi
    python bench.py $F > /tmp/b.json 2>/tmp/err.log
    python - "${lib:-default}" <<'PY'
import json, sys
try:
    d = json.loads(open("/tmp/b.json").read().strip().splitlines()[-1]); t = d["roofline_token_loop"]; r = d["roofline"]
    print(f"{sys.argv[1]:24s} {d['value']:7.1f} fps  step in-run {t['ms_per_step']:.3f} ms alone {t['alone']['ms_per_step']:.3f}  conv in-run {r['in_timed_region']['achieved']:.1f} alone {r['achieved']:.1f} {r.get('alone_passes')}  self_check {d['self_check']['pipelined_equals_serial']}", flush=True)
except Exception as e:
    print(sys.argv[1], "FAILED", e, open("/tmp/err.log").read()[-400:])
PY
done
