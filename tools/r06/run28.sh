#!/bin/bash
# round 6, GPU call 28: the driver's line with persistent tiles after the weight DMA moved into the first tap's shadow (A / B / B' / A on one box)
cd /root/repo
O=gpurun_out/r06ab; mkdir -p $O
run() { # name, env...
  local name=$1; shift
  env "$@" timeout 900 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_$name.json 2> $O/bench_$name.err
  python3 - $O/bench_$name.json $name <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    r = d.get("roofline", {}); sc = d.get("single_call", {})
    print(sys.argv[2], "fps %.1f" % d["value"], "self_check", d.get("self_check", {}).get("pipelined_equals_serial"), "conv alone TF %.1f" % r.get("achieved"), "in-run frac %.4f" % r.get("frac_in_run"),
          "serial %.1f stream %.1f" % (sc.get("serial_frames_per_s"), sc.get("stream_frames_per_s")))
except Exception as e:
    print(sys.argv[2], "FAILED", e)
PY
}
run A0 CCVS_CONV_PT=0
run B3 CCVS_CONV_PT=3
run C1 CCVS_CONV_PT=1
run A1 CCVS_CONV_PT=0
