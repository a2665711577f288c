#!/bin/bash
# Round 6, GPU call 9: the N-rank launch path of bench.py on a one-GPU box (torch.distributed.run + RCCL with one rank).
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06h
( time CCVS_BENCH_FORCE_LAUNCHER=1 timeout 900 python bench.py --gpus 1 --steps 6 --warmup 2 --no-cpu-baseline --no-strict-f32 --no-encode-cond-leg --no-other-noise-leg ) > gpurun_out/r06h/bench_launcher.json 2> gpurun_out/r06h/bench_launcher.err
tail -n 3 gpurun_out/r06h/bench_launcher.err
python - <<'PY'
import json
r = json.loads([l for l in open("gpurun_out/r06h/bench_launcher.json") if l.startswith("{")][-1])
print("fps %.1f" % r["value"], "ranks", r["multi_gpu"]["rccl_ranks"], r["multi_gpu"]["backend"], "seed", r["multi_gpu"]["noise_seed"]["rank_0"], "self_check", r["self_check"]["pipelined_equals_serial"], "single_call", r["single_call"]["stream_frames_per_s"])
PY
