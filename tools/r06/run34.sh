#!/bin/bash
# round 6, GPU call 34: HBM-side traffic of the convolutions of one batch on the round's final library (one process: persistent tiles where they apply) --
# the kernels of the bench line's alone passes
cd /root/repo
timeout 1200 bash tools/pmc_conv_traffic.sh > gpurun_out/conv_traffic_call34.log 2>&1; echo rc=$?
tail -40 gpurun_out/conv_traffic_table.txt
