#!/bin/bash
# Round 6, GPU call 8: kernel trace of the pipelined bench (steady-state window) -> the in-run duration of the decode GEMM for
# profiles/gemm16_inrun.json, and the per-queue anatomy of round 5 once more on the final library.
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06g
O=$GRAFT_REPO_ROOT/gpurun_out/r06g
cd /tmp && export TMPDIR=/tmp
export CCVS_BENCH_SUPERVISE=0
for TRY in 1 2 3; do
  rm -rf /tmp/trace
  timeout 900 rocprofv3 --kernel-trace --output-format csv -d /tmp/trace -- python3 $GRAFT_REPO_ROOT/bench.py --steps 12 --warmup 4 --no-cpu-baseline --no-strict-f32 --no-encode-cond-leg --no-other-noise-leg > /tmp/trace.log 2>&1
  ls /tmp/trace/*/*kernel_trace.csv > /dev/null 2>&1 && break
done
grep "^{" /tmp/trace.log | tail -n 1 > $O/bench_under_trace.json
python3 $GRAFT_REPO_ROOT/tools/r05/trace_gaps.py /tmp/trace 0.30 0.50 > $O/trace_gaps.txt 2>&1
python3 $GRAFT_REPO_ROOT/tools/r05/trace_beside.py /tmp/trace 0.30 0.50 > $O/trace_beside.txt 2>&1
python3 - > $O/gemm16_inrun_window.json <<'PY'
import csv, glob, json
csv.field_size_limit(1 << 30)
rows = []
for f in glob.glob("/tmp/trace/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Queue_Id"], r["Kernel_Name"].split("(")[0]))
rows.sort()
t0, t1 = rows[0][0], rows[-1][1]
w0, w1 = t0 + 0.30 * (t1 - t0), t0 + 0.50 * (t1 - t0)
out = {}
for s, e, q, n in rows:
    if w0 <= s <= w1 and ("gemm16_kernel<0, 2, 2, 1>" in n or "attention_decode_kernel" in n):
        k = "gemm16" if "gemm16" in n else "attention"
        d = out.setdefault(k, {}).setdefault(q, [0, 0])
        d[0] += 1; d[1] += e - s
print(json.dumps({k: {q: {"launches": v[0], "avg_us": v[1] / v[0] / 1e3} for q, v in d.items()} for k, d in out.items()}, indent=1))
PY
cat $O/gemm16_inrun_window.json; head -n 20 $O/trace_gaps.txt
