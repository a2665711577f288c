#!/bin/bash
# Everything profiles/ holds for a round, collected on the GPU box into gpurun_out/<tag>_*: bench lines (default pipelined
# schedule with cpu_baseline, serial, fp32-MFMA convolutions, Kinetics), rocprofv3 kernel stats of the bench command, the
# convolution and attention PMC passes.  Every command has its own time limit (a stalled one must not eat the whole call).
# usage: bash tools/collect_profiles.sh r02
TAG=${1:-rXX}
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd $ROOT
timeout 900 python3 bench.py > $OUT/${TAG}_bench_default.json 2> $OUT/${TAG}_bench_default.err
timeout 600 python3 bench.py --steps 8 --warmup 2 --no-cpu-baseline --schedule serial > $OUT/${TAG}_bench_serial.json 2>/dev/null
timeout 600 python3 bench.py --steps 20 --warmup 2 --no-cpu-baseline > $OUT/${TAG}_bench_pipelined_20.json 2>/dev/null
timeout 600 python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --schedule serial --conv-precision f32 > $OUT/${TAG}_bench_serial_f32conv.json 2>/dev/null
timeout 600 python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --config kinetics --batch 64 > $OUT/${TAG}_bench_kinetics.json 2> $OUT/${TAG}_bench_kinetics.err
timeout 600 python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --rec-pass --schedule serial > $OUT/${TAG}_bench_serial_recpass.json 2>/dev/null
cd /tmp && export TMPDIR=/tmp
export CCVS_BENCH_SUPERVISE=0   # under rocprofv3 the measuring process is the profiled one (no supervising parent)
for SCHED in pipelined serial; do
  rm -rf /tmp/prof_$SCHED
  timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$SCHED -- python3 $ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --schedule $SCHED > /tmp/prof_$SCHED.log 2>&1
  cp $(ls /tmp/prof_$SCHED/*/*kernel_stats.csv | head -1) $OUT/${TAG}_bench_${SCHED}_kernel_stats.csv
  tail -1 /tmp/prof_$SCHED.log > $OUT/${TAG}_bench_${SCHED}_under_rocprof.json
done
unset CCVS_BENCH_SUPERVISE
timeout 900 bash $ROOT/tools/pmc_conv_counters.sh > $OUT/${TAG}_pmc_conv_counters.txt 2>&1
timeout 600 bash $ROOT/tools/pmc_attention.sh > $OUT/${TAG}_pmc_attention.txt 2>&1
timeout 900 bash $ROOT/tools/pmc_conv_traffic.sh > $OUT/${TAG}_pmc_conv_traffic.log 2>&1
cp $OUT/conv_traffic.json $OUT/${TAG}_conv_traffic.json 2>/dev/null
python3 $ROOT/tools/attn_prefill_one.py 16 1024 20 > $OUT/${TAG}_attn_prefill.txt 2>&1
python3 $ROOT/tools/token_hog_probe.py 300 2>&1 | grep -v Loading > $OUT/${TAG}_token_hog_probe.txt
ls -la $OUT | tail -30
