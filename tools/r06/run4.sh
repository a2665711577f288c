#!/bin/bash
# Round 6, GPU call 4: the single call with the decoder following the token loop; where the 99 -> 128 convolution over-fetches;
# FETCH_SIZE calibrated on the gather kernels' access pattern.
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06d
O=gpurun_out/r06d
( time timeout 900 python tools/r06/single_call_probe.py 3 ) > $O/single_call_probe.txt 2>&1
grep -v Loading $O/single_call_probe.txt | tail -n 8
( time SHAPES="99 128 3 256 240;99 128 3 256 240 pre=15;96 128 3 256 16" XCDS="1" bash tools/r06/pmc_overfetch.sh ) > $O/pmc_overfetch_99.txt 2>&1
grep -v '^$' $O/pmc_overfetch_99.txt | tail -n 20
( time bash tools/r06/pmc_gather_calibrate.sh ) > $O/pmc_gather_calibrate.txt 2>&1
cat $O/pmc_gather_calibrate.txt | tail -n 12
