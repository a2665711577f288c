#!/bin/bash
cd "$(dirname "$0")/../.."
for shape in "128 64 3 256 120 p8" "64 32 3 256 120 p8" "128 64 3 128 480 p8"; do
  for x in 0 1 2; do
    echo -n "xspread=$x  "
    CCVS_CONV_XSPREAD=$x python tools/conv_one.py $shape 2>/dev/null | tail -1
  done
done
