#!/bin/bash
# round 4, GPU call 18: bench lines and the kernel trace again on the frame-by-frame schedule (the kernels are unchanged since
# tools/collect_profiles.sh r04), then the whole GPU suite
ROOT=${GRAFT_REPO_ROOT:-/root/repo}; cd $ROOT; OUT=$ROOT/gpurun_out; TAG=r04b; mkdir -p $OUT
timeout 900 python3 bench.py --steps 20 --warmup 5 > $OUT/${TAG}_bench_default.json 2> $OUT/${TAG}_bench_default.err
timeout 600 python3 bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-strict-f32 --no-encode-cond-leg --config kinetics --batch 64 --chains 3 > $OUT/${TAG}_bench_kinetics.json 2> $OUT/${TAG}_bench_kinetics.err
timeout 600 python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-strict-f32 --no-encode-cond-leg --config drums --batch 8 > $OUT/${TAG}_bench_drums.json 2> $OUT/${TAG}_bench_drums.err
timeout 600 python3 bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-strict-f32 --no-encode-cond-leg --config bair-p2p > $OUT/${TAG}_bench_bair_p2p.json 2> $OUT/${TAG}_bench_bair_p2p.err
python3 - <<'PY'
import json
for f in ["default","kinetics","drums","bair_p2p"]:
    try:
        d=json.loads(open(f"gpurun_out/r04b_bench_{f}.json").read().strip().splitlines()[-1])
        print(f, round(d["value"],1), {k:round(v) for k,v in d["stage_ms_per_step"].items()}, d.get("decoder_stream",{}).get("idle_frac"), d.get("pipeline_hbm",{}).get("GBps_of_these_two"), (d.get("encode_cond_only") or {}).get("value"), flush=True)
    except Exception as e: print(f,"failed",e, flush=True)
PY
cd /tmp && export TMPDIR=/tmp
export CCVS_BENCH_SUPERVISE=0
rm -rf /tmp/prof_p
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_p -- python3 $ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-strict-f32 --no-encode-cond-leg --schedule pipelined > /tmp/prof_p.log 2>&1
cp $(ls /tmp/prof_p/*/*kernel_stats.csv | head -1) $OUT/${TAG}_bench_pipelined_kernel_stats.csv
grep "^{" /tmp/prof_p.log | tail -1 > $OUT/${TAG}_bench_pipelined_under_rocprof.json
unset CCVS_BENCH_SUPERVISE
cd $ROOT
timeout 2400 python -m pytest tests -m gpu -x -q > $OUT/${TAG}_gpu_tests.log 2>&1; tail -3 $OUT/${TAG}_gpu_tests.log
