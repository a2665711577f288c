#!/bin/bash
# round 4, GPU call 26: the four-pixel warp kernels with 64 x 16 tiles per workgroup (CCVS_WARP_TILED)
cd "${GRAFT_REPO_ROOT:-.}"; O=gpurun_out/r04; mkdir -p $O
timeout 600 python -m pytest tests/test_ops_gpu.py -x -q -k "warp" 2>&1 | tail -3
for sm in 0 1; do for fs in 0.02 0.05 0.1 0.2; do for m in 0 1; do
  echo "--- smooth=$sm FLOW_SCALE=$fs (x32 px) CCVS_WARP_TILED=$m"
  FLOW_SMOOTH=$sm FLOW_SCALE=$fs CCVS_WARP_TILED=$m timeout 300 python tools/mem_bench.py 2>&1 | grep -E "^backwarp 120|warp_fuse_blend|warp \+ 1x1"
done; done; done
for m in 0 1; do for cch in 4 8; do
  echo "--- smooth FLOW_SCALE=0.1 TILED=$m FUSE_CCH=$cch: $(FLOW_SMOOTH=1 FLOW_SCALE=0.1 CCVS_WARP_TILED=$m CCVS_FUSE_CCH=$cch timeout 300 python tools/mem_bench.py 2>&1 | grep warp_fuse_blend)"
done; done
echo "--- window kernel, smooth 0.1: $(FLOW_SMOOTH=1 FLOW_SCALE=0.1 CCVS_FUSE_WIN=1 timeout 300 python tools/mem_bench.py 2>&1 | grep warp_fuse_blend)"
