#!/bin/bash
# round 4, GPU call 3: wave priority of the conv kernels, two workgroups per CU for the small-Cin layers, the dense prefill GEMM
cd "${GRAFT_REPO_ROOT:-.}"; O=gpurun_out/r04; mkdir -p $O
H="HBM stream x256,L2 re-read x256,VALU spin x256,idle x256"
for pr in 0 1 3; do
  echo "== CCVS_CONV_PRIO=$pr" >> $O/prio_probe.txt
  CCVS_CONV_PRIO=$pr timeout 300 python tools/conv_contention_probe.py 120 "$H" "195,128,64,32" 2>&1 | grep -v amdgpu.ids >> $O/prio_probe.txt
done
cat $O/prio_probe.txt
for w in 0 112; do
  for shape in "49 128 3 256 120" "99 128 3 256 120" "49 128 3 128 240" "99 128 3 128 240"; do
    echo -n "WPC2=$w  " >> $O/wpc2.txt
    CCVS_CONV_WPC2=$w timeout 120 python tools/conv_one.py $shape 2>&1 | tail -1 >> $O/wpc2.txt
  done
done
cat $O/wpc2.txt
CCVS_CONV_WPC2=112 timeout 600 python -m pytest tests/test_ops_gpu.py tests/test_pipeline_gpu.py -x -q -k "conv or gemm or rows or decode_step" > $O/tests3_wpc2.log 2>&1; tail -3 $O/tests3_wpc2.log
CCVS_CONV_WPC2=112 timeout 300 python tools/conv_shape_census.py 2>&1 | head -14 > $O/census_wpc2.txt; cat $O/census_wpc2.txt
timeout 300 python tools/gemm_seq_bench.py > $O/gemm_seq_dense.txt 2>&1; cat $O/gemm_seq_dense.txt
CCVS_GEMM_SEQ_DENSE=0 timeout 300 python tools/gemm_seq_bench.py > $O/gemm_seq_rb.txt 2>&1; grep "four GEMMs" $O/gemm_seq_rb.txt
timeout 900 python -m pytest tests/test_real_geometry_gpu.py tests/test_e2e_gpu.py -x -q -k "gpt or golden or generator" > $O/tests3_gpt.log 2>&1; tail -3 $O/tests3_gpt.log
for cfg in "0 0" "3 0" "3 112" "1 0"; do
  set -- $cfg
  CCVS_CONV_PRIO=$1 CCVS_CONV_WPC2=$2 CCVS_CPU_BASELINE_BUDGET=1 timeout 600 python bench.py > $O/bench_p$1_w$2.json 2> $O/bench_p$1_w$2.err
  python - "$1" "$2" <<'PY'
import json,sys
p,w=sys.argv[1:3]
try:
    d=json.loads(open(f"gpurun_out/r04/bench_p{p}_w{w}.json").read().strip().splitlines()[-1])
    print("prio",p,"wpc2",w,"fps",round(d["value"],2),"alone TF",round(d["roofline"]["achieved"],1),"in-run TF",round(d["roofline"]["in_timed_region"]["achieved"],1),d.get("stage_ms_per_step"), "tok step ms", d["roofline_token_loop"]["ms_per_step"])
except Exception as e: print(p,w,"failed",e)
PY
done
