#!/bin/bash
# Weak-scaling sweep on ONE node: bench.py at N = 1, 2, 4, 8 GPUs back to back (each bare invocation starts its own ranks
# through torch.distributed.run), one JSON line per N into gpurun_out/scale_N.json, and the efficiency the driver would
# derive: value(N) / (N * value(1)).   usage: bash tools/scale_sweep.sh [steps] [warmup]      (needs an N-GPU box)
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $ROOT
STEPS=${1:-20}; WARM=${2:-5}
mkdir -p gpurun_out
NGPU=$(python3 -c 'import torch; print(torch.cuda.device_count())')
for N in 1 2 4 8; do
  [ $N -le $NGPU ] || { echo "N=$N: only $NGPU GPU(s) visible, skipped"; continue; }
  python3 bench.py --gpus $N --steps $STEPS --warmup $WARM --no-cpu-baseline --no-strict-f32 | tail -1 > gpurun_out/scale_$N.json
done
python3 - <<'PY'
import glob, json
rows = {}
for f in sorted(glob.glob("gpurun_out/scale_*.json")):
    try:
        d = json.loads(open(f).read())
        rows[d["n_gpus"]] = d
    except Exception as e:
        print(f, "unreadable:", e)
if 1 in rows:
    base = rows[1]["value"]
    for n, d in sorted(rows.items()):
        print(f"N={n}: {d['value']:.1f} frames/s whole job, {d['value'] / n:.1f} per GPU, weak-scaling efficiency {d['value'] / (n * base):.3f}, "
              f"ranks {d['multi_gpu']['rccl_ranks']}")
PY
