#!/bin/bash
# round 4, GPU call 25: fusion / blend tail through an LDS window
cd "${GRAFT_REPO_ROOT:-.}"; O=gpurun_out/r04; mkdir -p $O
timeout 600 python -m pytest tests/test_ops_gpu.py -x -q -k "warp" 2>&1 | tail -5
for fs in 0.02 0.05 0.1; do for m in 0 1 2; do
  echo "FLOW_SCALE=$fs CCVS_FUSE_WIN=$m: $(FLOW_SCALE=$fs CCVS_FUSE_WIN=$m timeout 300 python tools/mem_bench.py 2>&1 | grep warp_fuse_blend)"
done; done
