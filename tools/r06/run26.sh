#!/bin/bash
# round 6, GPU call 26: matrix-pipe busy cycles and launch cycles of the convolution on three shapes, producer / consumer kernel against persistent tiles
# (separate --pmc passes with --kernel-trace only, tools/pmc_conv_counters.sh)
cd /root/repo
O=$GRAFT_REPO_ROOT/gpurun_out/r06z; mkdir -p $O
{
for shape in "195 128 3 256 240" "49 128 3 256 240 bf16x3 p8out" "99 128 3 256 240 bf16x3 p8out pre=15" "128 64 3 256 240 p8"; do
  for v in 0 3; do
    echo "## CCVS_CONV_PT=$v"
    CCVS_CONV_PT=$v SHAPE="$shape" bash $GRAFT_REPO_ROOT/tools/pmc_conv_counters.sh "GRBM_GUI_ACTIVE" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES"
  done
done
} > $O/pmc_conv_pt.txt 2>&1
cat $O/pmc_conv_pt.txt
