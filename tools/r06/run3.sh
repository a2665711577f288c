#!/bin/bash
# Round 6, GPU call 3: the whole GPU suite on the refactored decode kernels, the over-fetch experiment, and the gemm16 register
# attribute as a second library (build/nv48, -DCCVS_GEMM16_NUM_VGPR=48) against the default one.
# The second library is not kept in the tree: cp -r ccvs_amd/csrc build/nv48 && make -C build/nv48 clean all EXTRA=-DCCVS_GEMM16_NUM_VGPR=48
# (ROOT in its Makefile pointing at the repository) before the call; the result is profiles/r06_gemm16_num_vgpr_ab.txt.
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06c
O=gpurun_out/r06c
( time timeout 1500 python -m pytest tests -x -q -m gpu ) > $O/gpu_tests.log 2>&1
tail -n 4 $O/gpu_tests.log
( time bash tools/r06/pmc_overfetch.sh ) > $O/pmc_overfetch.txt 2>&1
cat $O/pmc_overfetch.txt | grep -v '^$' | tail -n 30
LEGS="--no-other-noise-leg --no-strict-f32 --no-encode-cond-leg --no-cpu-baseline"
NV=$GRAFT_REPO_ROOT/build/nv48/libccvs_hip.so
for tag in default nv48; do
  if [ $tag = nv48 ]; then export CCVS_LIB=$NV; else unset CCVS_LIB; fi
  ( time timeout 600 python tools/token_step_probe.py 300 16 64 ) > $O/token_step_probe_$tag.txt 2>&1
  grep rows $O/token_step_probe_$tag.txt
  ( time timeout 900 python bench.py --schedule serial --steps 6 --warmup 2 $LEGS ) > $O/bench_serial_$tag.json 2> $O/bench_serial_$tag.err
  ( time timeout 900 python bench.py --steps 20 --warmup 5 $LEGS ) > $O/bench_pipelined_$tag.json 2> $O/bench_pipelined_$tag.err
done
unset CCVS_LIB
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r06c/bench_*.json")):
    try:
        r = json.loads([l for l in open(f) if l.startswith("{")][-1])
    except Exception as e:
        print(f, "NO LINE", e); continue
    tl = r.get("roofline_token_loop", {})
    print(f, "fps %.1f" % r["value"], "step_ms %.3f" % tl.get("ms_per_step", 0), "stage", {k: round(v) for k, v in r["stage_ms_per_step"].items()},
          "conv alone %.1f" % r["roofline"]["achieved"], "self_check", (r.get("self_check") or {}).get("pipelined_equals_serial"))
PY
