#!/bin/bash
# round 6, GPU call 17: two workgroups per CU on packed input (CCVS_CONV_P8_WPC2) and with packed output (CCVS_CONV_WPC2_P8OUT), per shape, with checksums
cd /root/repo
export CCVS_CONV_ONE_SUM=1
O=gpurun_out/r06q; mkdir -p $O
{
for shape in "128 64 3 256 240 p8" "128 64 3 128 240 p8" "104 128 3 256 240 p8" "64 32 3 256 240 p8"; do
  for v in 0 1; do
    echo "## CCVS_CONV_P8_WPC2=$v  $shape"
    CCVS_CONV_P8_WPC2=$v timeout 300 python3 tools/conv_one.py $shape
  done
done
for shape in "49 128 3 256 240 bf16x3 p8out" "99 128 3 256 240 bf16x3 p8out"; do
  for v in 0 1; do
    echo "## CCVS_CONV_WPC2_P8OUT=$v CCVS_CONV_WPC2=128  $shape"
    CCVS_CONV_WPC2=128 CCVS_CONV_WPC2_P8OUT=$v timeout 300 python3 tools/conv_one.py $shape
  done
done
} > $O/wpc2_p8.txt 2>&1
cat $O/wpc2_p8.txt
