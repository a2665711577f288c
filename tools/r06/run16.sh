#!/bin/bash
# Round 6, GPU call 16: how much of the persistent step's loss is residency (one workgroup per CU) and how much the grid barrier:
# the step ALONE with 1 / 2 / 4 workgroups per CU (all resident only because nothing else runs).
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06p
for W in 1 2 4; do
  echo "## CCVS_DECODE_PERSISTENT=1 CCVS_STEP_WGS_PER_CU=$W"
  CCVS_DECODE_PERSISTENT=1 CCVS_STEP_WGS_PER_CU=$W timeout 600 python tools/token_step_probe.py 300 16 64 128 2>&1 | grep rows
done > gpurun_out/r06p/persistent_wgs_per_cu.txt
echo "## launch chain" >> gpurun_out/r06p/persistent_wgs_per_cu.txt
timeout 600 python tools/token_step_probe.py 300 16 64 128 2>&1 | grep rows >> gpurun_out/r06p/persistent_wgs_per_cu.txt
cat gpurun_out/r06p/persistent_wgs_per_cu.txt
