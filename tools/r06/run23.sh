#!/bin/bash
# round 6, GPU call 23: the persistent-tile bit-identity test, then the whole GPU suite with CCVS_CONV_PT=3 (every parity test through the persistent form where it applies)
cd /root/repo
O=gpurun_out/r06w; mkdir -p $O
timeout 900 python3 -m pytest tests/test_ops_gpu.py -q -m gpu -k persistent_tiles 2>&1 | tail -5 > $O/test_pt.log
cat $O/test_pt.log
CCVS_CONV_PT=3 timeout 1500 python3 -m pytest tests -x -q -m gpu 2>&1 | tail -8 > $O/gpu_tests_pt3.log
cat $O/gpu_tests_pt3.log
