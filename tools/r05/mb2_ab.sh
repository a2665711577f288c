#!/bin/bash
cd "$(dirname "$0")/../.."
for shape in "99 128 3 256 240 p8out" "49 128 3 256 240 p8out" "195 128 3 256 120 p8out" "99 128 3 128 480 p8out" "128 128 3 256 16" "96 128 3 256 16" "128 128 3 128 32"; do
  for v in 0 1 2; do
    echo -n "MB2_128=$v  "
    CCVS_CONV_MB2_128=$v python tools/conv_one.py $shape 2>/dev/null | tail -1
  done
done
