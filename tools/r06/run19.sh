#!/bin/bash
# round 6, GPU call 19: persistent tiles (CCVS_CONV_PT) against the producer / consumer kernel, per shape, with checksums of the outputs
cd /root/repo
export CCVS_CONV_ONE_SUM=1
O=gpurun_out/r06s; mkdir -p $O
{
for shape in "128 64 3 256 240 p8" "99 128 3 256 240 bf16x3 p8out pre=15" "99 128 3 256 240" "49 128 3 256 240 bf16x3 p8out" "195 128 3 64 240" "128 64 3 128 240 p8" "104 128 3 256 240 p8" "128 128 3 256 16" "96 128 3 256 16"; do
  for v in 0 3; do
    echo "## CCVS_CONV_PT=$v  $shape"
    CCVS_CONV_PT=$v timeout 120 python3 tools/conv_one.py $shape 2>&1 | grep -v amdgpu.ids
    rc=$?; [ $rc -ne 0 ] && echo "rc=$rc"
  done
done
} > $O/pt_shapes.txt 2>&1
cat $O/pt_shapes.txt
