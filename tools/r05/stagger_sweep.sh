#!/bin/bash
# Experiment (round 5): do the CUs run their convolution tiles in lock-step?  CCVS_CONV_STAGGER=s delays the first workgroup of
# every CU by ph x s x ~3.8 us (ph = 0..7): if the prologue / epilogue bursts of 256 CUs coincide, spreading the phases shortens
# the launch.  One shape per line, N = 120 images at 256^2.
cd "$(dirname "$0")/../.."
for shape in "195 128 3 256 120" "99 128 3 256 120" "49 128 3 256 120" "128 64 3 256 120 p8" "64 32 3 256 120 p8" "128 64 3 256 120"; do
  for s in 0 1 2 4 8; do
    echo -n "stagger=$s  "
    CCVS_CONV_STAGGER=$s python tools/conv_one.py $shape 2>/dev/null | tail -1
  done
done
