#!/bin/bash
# round 6, GPU call 25: the encoder's normalize_out test on the GPU
cd /root/repo
O=gpurun_out/r06y; mkdir -p $O
timeout 600 python3 -m pytest tests/test_ops_gpu.py tests/test_capi_host.py -q -m gpu -k "normalize or persistent_tiles" 2>&1 | tail -5 > $O/test_normalize.log; cat $O/test_normalize.log
