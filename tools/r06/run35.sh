#!/bin/bash
# round 6, GPU call 35: the single generate call with persistent tiles for the fp32-input layers (mode 1, the default) against all layers (mode 3) and none (0)
cd /root/repo
O=gpurun_out/r06ah; mkdir -p $O
for v in 1 3 0 1 3; do
  echo "## CCVS_CONV_PT=$v"
  CCVS_CONV_PT=$v timeout 400 python3 tools/r06/single_call_probe.py 4 2>&1 | grep -v "Loading\|amdgpu.ids" | tail -4
done > $O/single_call_pt.txt 2>&1
cat $O/single_call_pt.txt
