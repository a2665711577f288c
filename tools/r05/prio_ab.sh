#!/bin/bash
# Issue priorities of the convolution waves (CCVS_CONV_PRIO, s_setprio in the producer / consumer kernels) and of the token kernels
# (library built with -DCCVS_TOKEN_PRIO=3: tools/r05/lib_prio3.so), one box.
F="--steps 20 --warmup 5 --no-cpu-baseline --no-other-noise-leg --no-strict-f32 --no-encode-cond-leg"
run() {
    tag=$1; shift
    env "$@" python bench.py $F > /tmp/b.json 2>/tmp/err.log
    python - "$tag" <<'PY'
import json, sys
try:
    d = json.loads(open("/tmp/b.json").read().strip().splitlines()[-1]); t = d["roofline_token_loop"]; r = d["roofline"]
    print(f"{sys.argv[1]:34s} {d['value']:7.1f} fps  step in-run {t['ms_per_step']:.3f} ms  conv in-run {r['in_timed_region']['achieved']:.1f} alone {r['achieved']:.1f}  decode {d['stage_ms_per_step']['decode']:.0f} ms", flush=True)
except Exception as e:
    print(sys.argv[1], "FAILED", e, open("/tmp/err.log").read()[-400:])
PY
}
run "conv prio 2, tokens 0" CCVS_CONV_PRIO=2
run "conv prio 0, tokens 0" CCVS_CONV_PRIO=0
run "conv prio 3, tokens 0" CCVS_CONV_PRIO=3
run "conv prio 1, tokens 3" CCVS_CONV_PRIO=1 CCVS_LIB=$PWD/tools/r05/lib_prio3.so
run "conv prio 2, tokens 0" CCVS_CONV_PRIO=2
