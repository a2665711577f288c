#!/bin/bash
# Everything profiles/ holds for a round, collected on the GPU box into gpurun_out/<tag>_*: bench lines (default pipelined
# schedule with cpu_baseline and strict_f32, serial, Kinetics, point-to-point, Drums), rocprofv3 kernel stats of the bench
# command, the convolution and attention PMC passes, the per-shape census and the footprint / step probes.  Every command has
# its own time limit (a stalled one must not eat the whole call).
# usage: bash tools/collect_profiles.sh r03 [quick]
TAG=${1:-rXX}
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd $ROOT
timeout 900 python3 bench.py --steps 20 --warmup 5 > $OUT/${TAG}_bench_default.json 2> $OUT/${TAG}_bench_default.err
timeout 600 python3 bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-strict-f32 --schedule serial > $OUT/${TAG}_bench_serial.json 2>/dev/null
timeout 600 python3 bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-strict-f32 --no-encode-cond-leg --config kinetics --batch 64 --chains 3 > $OUT/${TAG}_bench_kinetics.json 2> $OUT/${TAG}_bench_kinetics.err
timeout 600 python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-strict-f32 --no-encode-cond-leg --config drums --batch 8 > $OUT/${TAG}_bench_drums.json 2> $OUT/${TAG}_bench_drums.err
timeout 600 python3 bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-strict-f32 --no-encode-cond-leg --config bair-p2p > $OUT/${TAG}_bench_bair_p2p.json 2> $OUT/${TAG}_bench_bair_p2p.err
timeout 300 python3 tools/mem_bench.py 2>&1 | grep -v amdgpu.ids > $OUT/${TAG}_mem_bench.txt
python3 tools/conv_shape_census.py 2>&1 | grep -v Loading > $OUT/${TAG}_conv_shape_census.txt
timeout 300 python3 tools/token_step_probe.py 480 16 32 48 64 2>&1 | grep rows > $OUT/${TAG}_token_step_probe.txt
QUICK=${2:-}
cd /tmp && export TMPDIR=/tmp
export CCVS_BENCH_SUPERVISE=0   # (bench.py also recognises the profiler's preload by itself)
for SCHED in pipelined serial; do
  rm -rf /tmp/prof_$SCHED
  for TRY in 1 2 3; do   # (rocprofv3 itself segfaults at start-up every other time on this image: the bench never ran then)
  rm -rf /tmp/prof_$SCHED
  timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$SCHED -- python3 $ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-strict-f32 --no-encode-cond-leg --schedule $SCHED > /tmp/prof_$SCHED.log 2>&1
  ls /tmp/prof_$SCHED/*/*kernel_stats.csv > /dev/null 2>&1 && break
  done
  cp $(ls /tmp/prof_$SCHED/*/*kernel_stats.csv | head -1) $OUT/${TAG}_bench_${SCHED}_kernel_stats.csv
  grep "^{" /tmp/prof_$SCHED.log | tail -1 > $OUT/${TAG}_bench_${SCHED}_under_rocprof.json
done
unset CCVS_BENCH_SUPERVISE
if [ "$QUICK" = "quick" ]; then ls -la $OUT | tail -20; exit 0; fi   # the PMC passes below: only when their kernels changed
SHAPE="195 128 3 256 240" timeout 900 bash $ROOT/tools/pmc_conv_counters.sh > $OUT/${TAG}_pmc_conv_counters.txt 2>&1
SHAPE="128 64 3 256 240" timeout 900 bash $ROOT/tools/pmc_conv_counters.sh > $OUT/${TAG}_pmc_conv_counters_128to64.txt 2>&1
SHAPE="128 64 3 256 240 p8" timeout 900 bash $ROOT/tools/pmc_conv_counters.sh > $OUT/${TAG}_pmc_conv_counters_128to64_packed.txt 2>&1
timeout 600 bash $ROOT/tools/pmc_attention.sh > $OUT/${TAG}_pmc_attention.txt 2>&1
timeout 600 bash $ROOT/tools/pmc_gemm_seq.sh 3072 20480 > $OUT/${TAG}_pmc_gemm_seq.txt 2>&1
timeout 900 bash $ROOT/tools/pmc_decoder_kernels.sh > $OUT/${TAG}_pmc_decoder_kernels.txt 2>&1
timeout 900 bash $ROOT/tools/pmc_conv_traffic.sh > $OUT/${TAG}_pmc_conv_traffic.log 2>&1
cp $OUT/conv_traffic.json $OUT/${TAG}_conv_traffic.json 2>/dev/null
cp $OUT/conv_traffic_raw.json $OUT/${TAG}_conv_traffic_raw.json 2>/dev/null
ls -la $OUT | tail -30
