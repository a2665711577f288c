#!/bin/bash
# round 6, GPU call 21: persistent-tile convolutions on fewer CUs than the chip has (the rest stays free for the token kernels)
cd /root/repo
O=gpurun_out/r06u; mkdir -p $O
run() { # name, env...
  local name=$1; shift
  env "$@" timeout 900 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_$name.json 2> $O/bench_$name.err
  python3 - $O/bench_$name.json $name <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    r = d.get("roofline", {})
    print(sys.argv[2], "fps %.1f" % d["value"], "self_check", d.get("self_check", {}).get("pipelined_equals_serial"), "conv alone TF %.1f" % r.get("achieved"), "in-run frac %.4f" % r.get("frac_in_run"),
          "stages", d.get("stages", {}).get("per_batch_ms") or {k: v for k, v in d.get("stages", {}).items() if k in ("encode", "transformer", "decode")})
except Exception as e:
    print(sys.argv[2], "FAILED", e)
PY
}
run A0 CCVS_CONV_PT=0
run P1_240 CCVS_CONV_PT=1 CCVS_CONV_PT_CUS=240
run P1_224 CCVS_CONV_PT=1 CCVS_CONV_PT_CUS=224
run P3_240 CCVS_CONV_PT=3 CCVS_CONV_PT_CUS=240
