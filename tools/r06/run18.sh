#!/bin/bash
# round 6, GPU call 18: the driver's line with two workgroups per CU on the packed-input 64-channel layers and on 49->128 with packed output (A / B / A on one box)
cd /root/repo
O=gpurun_out/r06r; mkdir -p $O
run() { # name, env...
  local name=$1; shift
  env "$@" timeout 900 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_$name.json 2> $O/bench_$name.err
  python3 - $O/bench_$name.json $name <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    r = d.get("roofline", {})
    print(sys.argv[2], "fps", d["value"], "self_check", d.get("self_check"), "conv alone TF", r.get("achieved"), "in-run frac", r.get("frac_in_run"))
except Exception as e:
    print(sys.argv[2], "FAILED", e)
PY
}
run A0 CCVS_X=0
run B1 CCVS_CONV_P8_WPC2=1 CCVS_CONV_WPC2_P8OUT=1
run A1 CCVS_X=0
run B2 CCVS_CONV_P8_WPC2=1 CCVS_CONV_WPC2_P8OUT=0
tail -3 $O/*.err
