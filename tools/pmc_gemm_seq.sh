#!/bin/bash
# MFMA utilisation of the whole-sequence GEMM (gemm_seq_kernel, fp32 matrix cores): SQ_VALU_MFMA_BUSY_CYCLES (cycles the matrix
# pipe of a SIMD is busy, summed over SIMDs) against GRBM_GUI_ACTIVE (summed over the 8 XCDs) x 1024 SIMDs / 8 -- separate --pmc
# passes, kernel trace only.  The four GEMMs of one GPT layer at M rows (default 3072 = a BAIR token-group prefill, and 20480).
# usage (GPU box): bash tools/pmc_gemm_seq.sh [M ...]
cd /tmp && export TMPDIR=/tmp
MS=${@:-3072 20480}
for c in "GRBM_GUI_ACTIVE" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES"; do
  rm -rf /tmp/pmcg; rocprofv3 --kernel-trace --pmc $c -d /tmp/pmcg -o p --output-format csv -- python3 $GRAFT_REPO_ROOT/tools/gemm_seq_bench.py $MS > /tmp/pmcg.log 2>&1
  python3 - <<PY
import csv,glob,collections
f=glob.glob("/tmp/pmcg/**/*counter_collection.csv",recursive=True)
if not f: print("no counter file"); print(open("/tmp/pmcg.log").read()[-800:]); raise SystemExit
agg=collections.defaultdict(float); n=collections.defaultdict(int)
for r in csv.DictReader(open(f[0])):
    if "gemm_seq_kernel" in r["Kernel_Name"]: agg[r["Counter_Name"]]+=float(r["Counter_Value"]); n[r["Counter_Name"]]+=1
for k,v in agg.items(): print(k, "launches=%d sum=%.6g"%(n[k], v))
t=glob.glob("/tmp/pmcg/**/*kernel_trace.csv",recursive=True)
if t:
    d=[int(r["End_Timestamp"])-int(r["Start_Timestamp"]) for r in csv.DictReader(open(t[0])) if "gemm_seq_kernel" in r["Kernel_Name"]]
    print("gemm_seq_kernel: %d launches, %.4f ms in all"%(len(d), sum(d)/1e6))
if "SQ_VALU_MFMA_BUSY_CYCLES" in agg and "SQ_BUSY_CYCLES" in agg:
    print("MFMA busy / (busy cycles x 4 SIMDs per CU-unit): see the GRBM pass for the wall-clock form")
PY
done
grep "TFLOP" /tmp/pmcg.log | tail -12
