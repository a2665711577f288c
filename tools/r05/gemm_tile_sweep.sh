#!/bin/bash
# The decode GEMM's block tile and the token-group shape, one box (bench.py without its side legs).
#   gpurun -- bash tools/r05/gemm_tile_sweep.sh
mkdir -p gpurun_out/r05
F="--no-cpu-baseline --no-other-noise-leg --no-strict-f32 --no-encode-cond-leg"
show() { python - "$@" <<'PY'
import json, sys
tag, f = sys.argv[1], sys.argv[2]
try:
    d = json.loads(open(f).read().strip().splitlines()[-1]); t = d["roofline_token_loop"]; r = d["roofline"]
    print(f"{tag:28s} {d['value']:7.1f} fps  step in-run {t['ms_per_step']:.3f} ms alone {t['alone']['ms_per_step']:.3f} ms  conv in-run {r['in_timed_region']['achieved']:.1f} alone {r['achieved']:.1f} {r.get('alone_passes')} self_check {d['self_check']['pipelined_equals_serial']}", flush=True)
except Exception as e:
    print(tag, "FAILED", e)
PY
}
for cfg in "1 4 2" "2 4 2" "3 4 2" "0 4 2" "1 6 2" "1 8 2" "1 4 3" "1 4 2"; do
    set -- $cfg
    out=gpurun_out/r05/sweep_t$1_l$2_c$3_$RANDOM.json
    CCVS_GEMM_TILE2=$1 python bench.py $F --lanes $2 --chains $3 > $out 2>> gpurun_out/r05/sweep_err.log
    show "tile $1 lanes $2 chains $3" $out
done
