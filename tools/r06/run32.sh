#!/bin/bash
# round 6, GPU call 32: soak of the round's last tree (two runs of 60 batches back to back in one process, RCCL group alive)
cd /root/repo
O=gpurun_out/r06af; mkdir -p $O
timeout 1200 python3 tools/soak_pipelined.py 2 60 > $O/soak.txt 2>&1; echo "rc=$?" >> $O/soak.txt
tail -12 $O/soak.txt
