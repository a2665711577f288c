#!/bin/bash
# round 4, GPU call 31: the new defaults (4 lanes x 2 chains, two decode streams): pipeline tests, soak, secondary configurations
ROOT=${GRAFT_REPO_ROOT:-/root/repo}; cd $ROOT; OUT=$ROOT/gpurun_out; TAG=r04c; mkdir -p $OUT
timeout 900 python -m pytest tests/test_pipeline_gpu.py tests/test_e2e_gpu.py -x -q 2>&1 | tail -2
timeout 900 python tools/soak_pipelined.py 2 120 2>&1 | grep -E "^run|^soak" | tee $OUT/${TAG}_soak.txt
timeout 600 python3 bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-strict-f32 --no-encode-cond-leg --config kinetics --batch 64 --chains 3 > $OUT/${TAG}_bench_kinetics.json 2> $OUT/${TAG}_bench_kinetics.err
timeout 600 python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-strict-f32 --no-encode-cond-leg --config drums --batch 8 > $OUT/${TAG}_bench_drums.json 2> $OUT/${TAG}_bench_drums.err
timeout 600 python3 bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-strict-f32 --no-encode-cond-leg --config bair-p2p > $OUT/${TAG}_bench_bair_p2p.json 2> $OUT/${TAG}_bench_bair_p2p.err
CCVS_PIPELINE_DEC_STREAMS=1 timeout 600 python3 bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-strict-f32 --no-encode-cond-leg --config kinetics --batch 64 --chains 3 > $OUT/${TAG}_bench_kinetics_1dec.json 2>/dev/null
python3 - <<'PY'
import json
for f in ["kinetics","kinetics_1dec","drums","bair_p2p"]:
    try:
        d=json.loads(open(f"gpurun_out/r04c_bench_{f}.json").read().strip().splitlines()[-1])
        print(f, round(d["value"],1), {k:round(v) for k,v in d["stage_ms_per_step"].items()}, round(d["hbm_peak_allocated_gb"],1), flush=True)
    except Exception as e: print(f,"failed",e, flush=True)
PY
