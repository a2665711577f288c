#!/bin/bash
# HBM traffic of the convolution kernels over the encode + decode of one BAIR batch of 16 (the conv
# launches of one bench.py step; tools/decode_only.py avoids the hipGraph replays PMC cannot follow).
# FETCH_SIZE and WRITE_SIZE are collected in SEPARATE passes (TCC slot limit), kernel-trace only; a third pass counts the
# CUs' read requests to L2 (TCP_TCC_READ_REQ_sum: what the kernels move through the CU -> L2 path, hits included).
# The reduction (tools/pmc_conv_traffic_reduce.py) asks the LIBRARY which instantiations read 16 bytes per lane (FETCH_SIZE x2 on
# gfx950) and stops on a kernel name it cannot classify; with the per-launch algorithmic bytes of the same command it prints
# the over-fetch per instantiation.
# Usage (GPU box): bash tools/pmc_conv_traffic.sh   -> gpurun_out/conv_traffic_raw.json, conv_traffic.json, conv_traffic_table.txt
set -e
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
PASSES=""
for C in FETCH_SIZE WRITE_SIZE TCP_TCC_READ_REQ_sum; do
  rm -rf /tmp/pmc_$C
  if [ $C = FETCH_SIZE ]; then export CCVS_DUMP_CONV_BYTES=/tmp/conv_launch_bytes.json; else unset CCVS_DUMP_CONV_BYTES; fi
  rocprofv3 --kernel-trace --pmc $C --output-format csv -d /tmp/pmc_$C -- python3 $ROOT/tools/decode_only.py 16 > /tmp/pmc_$C.log 2>&1 || { tail -5 /tmp/pmc_$C.log; [ $C = TCP_TCC_READ_REQ_sum ] && continue; exit 1; }
  PASSES="$PASSES /tmp/pmc_$C"
done
mkdir -p $ROOT/gpurun_out
python3 $ROOT/tools/pmc_conv_traffic_reduce.py --passes $PASSES --launch-bytes /tmp/conv_launch_bytes.json --out-dir $ROOT/gpurun_out | tee $ROOT/gpurun_out/conv_traffic_table.txt
