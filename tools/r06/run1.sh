#!/bin/bash
# Round 6, GPU call 1: corrected conv traffic (PMC), host stress on the box's host, the baseline line with the new fields,
# one-chain schedule probes, the new GEMM tile-switch test, token step alone by rows.
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06
O=gpurun_out/r06
( time bash tools/pmc_conv_traffic.sh ) > $O/pmc_conv_traffic.log 2>&1
cp gpurun_out/conv_traffic.json gpurun_out/conv_traffic_raw.json gpurun_out/conv_traffic_table.txt $O/ 2>/dev/null
( time timeout 900 python tools/host_stress.py --ranks 1 8 --batches 10 ) > $O/host_stress.txt 2>&1
( time timeout 900 python -m pytest tests/test_pipeline_gpu.py -x -q -k "gemm_block_tile or decode_gemm_rows" ) > $O/test_gemm_tile.log 2>&1
( time timeout 600 python tools/token_step_probe.py 300 16 64 80 96 128 ) > $O/token_step_probe.txt 2>&1
( time timeout 1200 python bench.py --steps 20 --warmup 5 ) > $O/bench_default.json 2> $O/bench_default.err
LEGS="--no-other-noise-leg --no-strict-f32 --no-encode-cond-leg --no-cpu-baseline"
( time timeout 900 python bench.py --steps 20 --warmup 5 --lanes 5 --chains 1 $LEGS ) > $O/bench_5x1.json 2> $O/bench_5x1.err
( time timeout 900 python bench.py --steps 24 --warmup 6 --lanes 6 --chains 1 $LEGS ) > $O/bench_6x1_24.json 2> $O/bench_6x1_24.err
( time timeout 900 python bench.py --steps 24 --warmup 4 --lanes 4 --chains 2 $LEGS ) > $O/bench_4x2_24.json 2> $O/bench_4x2_24.err
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r06/bench_*.json")):
    try:
        r = json.loads([l for l in open(f) if l.startswith("{")][-1])
    except Exception as e:
        print(f, "NO LINE", e); continue
    tl = r.get("roofline_token_loop", {})
    print(f, "fps %.1f" % r["value"], "step_ms %.3f" % tl.get("ms_per_step", 0), "alone", (tl.get("alone") or {}).get("ms_per_step"), "groups", tl.get("batches_per_token_group"),
          "stage", {k: round(v) for k, v in r["stage_ms_per_step"].items()}, "conv alone %.1f in-run %.1f" % (r["roofline"]["achieved"], (r["roofline"]["in_timed_region"] or {}).get("achieved", 0)),
          "self_check", (r.get("self_check") or {}).get("pipelined_equals_serial"), "hbm_gb %.0f" % r.get("hbm_peak_allocated_gb", 0))
PY
