#!/bin/bash
# Round 6, GPU call 2: the persistent decode step -- bit-identity tests, the step alone by rows (chain vs persistent), and the
# one-chain schedules (8 x 1, 7 x 1, 6 x 1) with the launch chain and with the persistent step.
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06b
O=gpurun_out/r06b
( time timeout 1500 python -m pytest tests/test_persistent_step_gpu.py -x -q ) > $O/test_persistent.log 2>&1
tail -5 $O/test_persistent.log
for P in 0 1; do
  ( time CCVS_DECODE_PERSISTENT=$P timeout 600 python tools/token_step_probe.py 300 16 64 96 128 ) > $O/token_step_probe_p$P.txt 2>&1
  grep rows $O/token_step_probe_p$P.txt
done
LEGS="--no-other-noise-leg --no-strict-f32 --no-encode-cond-leg --no-cpu-baseline"
export CCVS_PIPELINE_MEM_FRAC=0.95
( time timeout 900 python bench.py --steps 20 --warmup 8 --lanes 8 --chains 1 $LEGS ) > $O/bench_8x1_chain.json 2> $O/bench_8x1_chain.err
( time CCVS_DECODE_PERSISTENT=1 timeout 900 python bench.py --steps 20 --warmup 8 --lanes 8 --chains 1 $LEGS ) > $O/bench_8x1_pers.json 2> $O/bench_8x1_pers.err
( time CCVS_DECODE_PERSISTENT=1 timeout 900 python bench.py --steps 24 --warmup 6 --lanes 6 --chains 1 $LEGS ) > $O/bench_6x1_24_pers.json 2> $O/bench_6x1_24_pers.err
( time timeout 900 python bench.py --steps 20 --warmup 7 --lanes 7 --chains 1 $LEGS ) > $O/bench_7x1_chain.json 2> $O/bench_7x1_chain.err
( time CCVS_DECODE_PERSISTENT=1 timeout 900 python bench.py --steps 20 --warmup 7 --lanes 7 --chains 1 $LEGS ) > $O/bench_7x1_pers.json 2> $O/bench_7x1_pers.err
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r06b/bench_*.json")):
    try:
        r = json.loads([l for l in open(f) if l.startswith("{")][-1])
    except Exception as e:
        print(f, "NO LINE", e); continue
    tl = r.get("roofline_token_loop", {})
    print(f, "fps %.1f" % r["value"], "step_ms %.3f" % tl.get("ms_per_step", 0), "alone", (tl.get("alone") or {}).get("ms_per_step"), "groups", tl.get("batches_per_token_group"),
          "stage", {k: round(v) for k, v in r["stage_ms_per_step"].items()}, "conv alone %.1f in-run %.1f" % (r["roofline"]["achieved"], (r["roofline"]["in_timed_region"] or {}).get("achieved", 0)),
          "self_check", (r.get("self_check") or {}).get("pipelined_equals_serial"), "hbm_gb %.0f" % r.get("hbm_peak_allocated_gb", 0))
PY
for f in $O/*.err; do echo "== $f"; tail -n 3 $f; done | tail -n 40
