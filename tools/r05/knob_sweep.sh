#!/bin/bash
# Schedule knobs once more on the 2 x 2 decode GEMM (the balance between the token loops and the decoder moved): one box.
F="--steps 20 --warmup 5 --no-cpu-baseline --no-other-noise-leg --no-strict-f32 --no-encode-cond-leg"
one() {  # tag, env...
    tag=$1; shift
    env "$@" python bench.py $F > /tmp/b.json 2>/tmp/err.log
    python - "$tag" <<'PY'
import json, sys
try:
    d = json.loads(open("/tmp/b.json").read().strip().splitlines()[-1]); t = d["roofline_token_loop"]; s = d["stage_ms_per_step"]
    print(f"{sys.argv[1]:34s} {d['value']:7.1f} fps  step {t['ms_per_step']:.3f} ms  decode {s['decode']:.0f} ms/batch  conv in-run {d['roofline']['in_timed_region']['achieved']:.1f}", flush=True)
except Exception as e:
    print(sys.argv[1], "FAILED", e, open("/tmp/err.log").read()[-300:])
PY
}
one "default" X=1
one "dec streams 3" CCVS_PIPELINE_DEC_STREAMS=3
one "dec streams 1" CCVS_PIPELINE_DEC_STREAMS=1
one "priorities 0,0" CCVS_PIPELINE_PRIORITIES=0,0
one "priorities -1,0" CCVS_PIPELINE_PRIORITIES=-1,0
one "cu limit 224" CCVS_PIPELINE_CU_LIMIT=224
one "kz mink 1024" CCVS_GEMM_KZ_MINK=1024
one "default" X=1
