#!/bin/bash
# round 4, GPU call 27: tile shapes of the four-pixel warp kernels (threads along a tile row: 8 = 32 x 32, 16 = 64 x 16, 32 = 128 x 8, 64 = 256 x 4 = plain at W = 256)
cd "${GRAFT_REPO_ROOT:-.}"; O=gpurun_out/r04; mkdir -p $O
timeout 600 python -m pytest tests/test_ops_gpu.py -x -q -k "warp" 2>&1 | tail -2
for sm in 1 0; do for fs in 0.05 0.1; do for m in 0 8 16 32; do
  echo "--- smooth=$sm FLOW_SCALE=$fs (x32 px) CCVS_WARP_TILED=$m"
  FLOW_SMOOTH=$sm FLOW_SCALE=$fs CCVS_WARP_TILED=$m timeout 300 python tools/mem_bench.py 2>&1 | grep -E "^backwarp 120|warp_fuse_blend|warp \+ 1x1"
done; done; done
for t in 0 8 16; do
  CCVS_WARP_TILED=$t timeout 400 python bench.py --no-cpu-baseline --no-strict-f32 --no-encode-cond-leg > $O/b27.json 2> $O/b27.err
  python - "TILED=$t" <<'PY'
import json,sys
try:
    d=json.loads(open("gpurun_out/r04/b27.json").read().strip().splitlines()[-1])
    print(sys.argv[1],"fps",round(d["value"],2),"stages",{k:round(v) for k,v in d["stage_ms_per_step"].items()},"alone TF",round(d["roofline"]["achieved"],1),"in-run TF",round(d["roofline"]["in_timed_region"]["achieved"],1), flush=True)
except Exception as e: print(sys.argv[1],"failed",e, flush=True)
PY
done
