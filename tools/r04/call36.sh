#!/bin/bash
# round 4, GPU call 36: the round's final tree -- whole GPU suite, smoke(), the driver's command, secondary configurations
ROOT=${GRAFT_REPO_ROOT:-/root/repo}; cd $ROOT; OUT=$ROOT/gpurun_out; TAG=r04e; mkdir -p $OUT
timeout 2400 python -m pytest tests -m gpu -x -q > $OUT/${TAG}_gpu_tests.log 2>&1; tail -3 $OUT/${TAG}_gpu_tests.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
timeout 900 python3 bench.py > $OUT/${TAG}_bench_default.json 2> $OUT/${TAG}_bench_default.err
timeout 600 python3 bench.py --steps 12 --warmup 2 --no-cpu-baseline --no-strict-f32 --no-encode-cond-leg --config kinetics --batch 64 --chains 3 > $OUT/${TAG}_bench_kinetics.json 2> $OUT/${TAG}_bench_kinetics.err
timeout 600 python3 bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-strict-f32 --no-encode-cond-leg --config drums --batch 8 > $OUT/${TAG}_bench_drums.json 2> $OUT/${TAG}_bench_drums.err
timeout 600 python3 bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-strict-f32 --no-encode-cond-leg --config bair-p2p > $OUT/${TAG}_bench_bair_p2p.json 2> $OUT/${TAG}_bench_bair_p2p.err
python3 - <<'PY'
import json
for f in ["default","kinetics","drums","bair_p2p"]:
    try:
        d=json.loads(open(f"gpurun_out/r04e_bench_{f}.json").read().strip().splitlines()[-1])
        print(f, round(d["value"],1), {k:round(v) for k,v in d["stage_ms_per_step"].items()}, "alone TF", round(d["roofline"]["achieved"],1), "peak GB", round(d["hbm_peak_allocated_gb"],1), (d.get("encode_cond_only") or {}).get("value"), flush=True)
    except Exception as e: print(f,"failed",e, flush=True)
PY
