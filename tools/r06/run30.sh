#!/bin/bash
# round 6, GPU call 30: the secondary configurations on the round's last tree (persistent tiles in their alone passes)
cd /root/repo
O=gpurun_out/r06ad; mkdir -p $O
timeout 600 python3 bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-strict-f32 --no-encode-cond-leg --config kinetics --batch 64 --chains 3 > $O/r06_bench_kinetics.json 2> $O/r06_bench_kinetics.err
timeout 600 python3 bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-strict-f32 --no-encode-cond-leg --config bair-p2p > $O/r06_bench_bair_p2p.json 2> $O/r06_bench_bair_p2p.err
timeout 600 python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-strict-f32 --no-encode-cond-leg --config drums --batch 8 > $O/r06_bench_drums.json 2> $O/r06_bench_drums.err
python3 - <<'PY'
import json
for n in ("kinetics", "bair_p2p", "drums"):
    try:
        d = json.loads(open(f"gpurun_out/r06ad/r06_bench_{n}.json").read().strip().splitlines()[-1])
        print(n, "fps %.1f" % d["value"], "self_check", (d.get("self_check") or {}).get("pipelined_equals_serial"), "conv alone %.1f" % d["roofline"]["achieved"])
    except Exception as e:
        print(n, "FAILED", e)
PY
for f in $O/*.err; do tail -n 2 $f; done
