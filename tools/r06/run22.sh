#!/bin/bash
# round 6, GPU call 22: socket power and shader clock during the driver's line (is the timed region power-managed?), with and without persistent-tile convolutions
cd /root/repo
O=gpurun_out/r06v; mkdir -p $O
/opt/rocm/bin/rocm-smi -P -c -t --showmaxpower --json > $O/smi_idle.json 2>&1
CCVS_CONV_PT=0 timeout 900 python3 tools/power_trace.py $O/power_pt0.json -- python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_pt0.json 2> $O/bench_pt0.err
CCVS_CONV_PT=3 timeout 900 python3 tools/power_trace.py $O/power_pt3.json -- python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_pt3.json 2> $O/bench_pt3.err
grep power_trace $O/*.err
cat $O/smi_idle.json | head -c 1500
