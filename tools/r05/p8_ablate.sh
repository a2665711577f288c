#!/bin/bash
# Where does a tile of the packed-input layers spend its time?  CCVS_CONV_ABLATE bits: 1 no activation DMA, 128 no weight DMA, 2 no MFMA,
# 16384 no epilogue, 4096 prologue only, 32768 dispatch only (results are wrong with any bit set).
cd "$(dirname "$0")/../.."
for shape in "128 64 3 256 120 p8" "64 32 3 256 120 p8"; do
  for a in 0 1 128 129 2 3 131 16384 16387 16515 4096 32768; do
    echo -n "ablate=$a  "
    CCVS_CONV_ABLATE=$a python tools/conv_one.py $shape 2>/dev/null | tail -1
  done
done
