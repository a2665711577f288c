#!/bin/bash
# round 6, GPU call 29: on the round's final tree -- rocprofv3 kernel stats of the bench command (both schedules), the serial line, the per-shape census
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r06ac; mkdir -p $OUT
cd $ROOT
timeout 600 python3 bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-strict-f32 --schedule serial > $OUT/r06_bench_serial.json 2>/dev/null
python3 tools/conv_shape_census.py 2>&1 | grep -v Loading > $OUT/r06_conv_shape_census.txt
head -12 $OUT/r06_conv_shape_census.txt
cd /tmp && export TMPDIR=/tmp
export CCVS_BENCH_SUPERVISE=0
for SCHED in serial pipelined; do
  for TRY in 1 2 3; do
  rm -rf /tmp/prof_$SCHED
  timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$SCHED -- python3 $ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-strict-f32 --no-encode-cond-leg --schedule $SCHED > /tmp/prof_$SCHED.log 2>&1
  ls /tmp/prof_$SCHED/*/*kernel_stats.csv > /dev/null 2>&1 && break
  done
  cp $(ls /tmp/prof_$SCHED/*/*kernel_stats.csv | head -1) $OUT/r06_bench_${SCHED}_kernel_stats.csv
  grep "^{" /tmp/prof_$SCHED.log | tail -1 > $OUT/r06_bench_${SCHED}_under_rocprof.json
  head -8 $OUT/r06_bench_${SCHED}_kernel_stats.csv | cut -c1-160
done
