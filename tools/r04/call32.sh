#!/bin/bash
# round 4, GPU call 32: the round's final tree -- whole GPU suite, smoke(), the driver's command, the kernel trace of the pipelined bench
ROOT=${GRAFT_REPO_ROOT:-/root/repo}; cd $ROOT; OUT=$ROOT/gpurun_out; TAG=r04d; mkdir -p $OUT
timeout 2400 python -m pytest tests -m gpu -x -q > $OUT/${TAG}_gpu_tests.log 2>&1; tail -3 $OUT/${TAG}_gpu_tests.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
timeout 900 python3 bench.py > $OUT/${TAG}_bench_default.json 2> $OUT/${TAG}_bench_default.err
python3 - <<'PY'
import json
d=json.loads(open("gpurun_out/r04d_bench_default.json").read().strip().splitlines()[-1])
print("fps",round(d["value"],2),"alone TF",round(d["roofline"]["achieved"],1),"in-run",round(d["roofline"]["in_timed_region"]["achieved"],1),"cond-only",round(d["encode_cond_only"]["value"],1),"strict",round(d["strict_f32"]["frames_per_s"],1),"cpu",d["cpu_baseline"]["value"],"peak GB",round(d["hbm_peak_allocated_gb"],1), d["decoder_stream"]["decode_streams"], flush=True)
PY
cd /tmp && export TMPDIR=/tmp
export CCVS_BENCH_SUPERVISE=0
rm -rf /tmp/prof_p
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_p -- python3 $ROOT/bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-strict-f32 --no-encode-cond-leg --schedule pipelined > /tmp/prof_p.log 2>&1
cp $(ls /tmp/prof_p/*/*kernel_stats.csv | head -1) $OUT/${TAG}_bench_pipelined_kernel_stats.csv
grep "^{" /tmp/prof_p.log | tail -1 > $OUT/${TAG}_bench_pipelined_under_rocprof.json
