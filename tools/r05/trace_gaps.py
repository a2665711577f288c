#!/usr/bin/env python3
"""Kernel-trace anatomy of the pipelined run: per hardware queue, how much of a token step is kernels and how much is the time
between a kernel's end and the next one's start (dependent launches of a replayed hipGraph).
    python tools/r05/trace_gaps.py <dir with *_kernel_trace.csv> [lo hi]     (window as fractions of the trace, default 0.35 0.65)"""
import csv
import glob
import sys
from collections import defaultdict

csv.field_size_limit(1 << 30)
root = sys.argv[1]
lo, hi = (float(sys.argv[2]), float(sys.argv[3])) if len(sys.argv) > 3 else (0.35, 0.65)
files = glob.glob(root + "/**/*kernel_trace.csv", recursive=True)
rows = []
for f in files:
    with open(f) as fh:
        for r in csv.DictReader(fh):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Queue_Id"], r["Kernel_Name"].split("(")[0][:48]))
rows.sort()
t0, t1 = rows[0][0], rows[-1][1]
w0, w1 = t0 + lo * (t1 - t0), t0 + hi * (t1 - t0)
print(f"{len(rows)} dispatches over {(t1 - t0) / 1e9:.2f} s; window {(w1 - w0) / 1e9:.2f} s")
byq = defaultdict(list)
for s, e, q, n in rows:
    if w0 <= s <= w1:
        byq[q].append((s, e, n))
for q, rr in sorted(byq.items(), key=lambda kv: -len(kv[1])):
    if len(rr) < 50:
        continue
    busy = sum(e - s for s, e, _ in rr)
    span = rr[-1][1] - rr[0][0]
    stat = defaultdict(lambda: [0, 0, 0, []])
    for i in range(1, len(rr)):
        s, e, n = rr[i]
        gap = s - rr[i - 1][1]
        st = stat[n]
        st[0] += 1; st[1] += e - s; st[2] += max(gap, 0); st[3].append(gap)
    print(f"queue {q}: {len(rr)} dispatches, span {span / 1e6:.1f} ms, kernels {100.0 * busy / span:.1f} % of it")
    edges = [0, 5e3, 20e3, 100e3, 1e6, 10e6, 1e12]
    hist = [[0, 0] for _ in edges[:-1]]
    for i in range(1, len(rr)):
        gap = rr[i][0] - rr[i - 1][1]
        if gap <= 0:
            continue
        for b in range(len(hist)):
            if edges[b] <= gap < edges[b + 1]:
                hist[b][0] += 1; hist[b][1] += gap
                break
    print("    gaps between consecutive kernels (<5 us, 5-20, 20-100, 100 us-1 ms, 1-10 ms, >10 ms): "
          + "   ".join(f"{c} x = {100.0 * t / span:.1f} %" for c, t in hist))
    for n, (c, d, g, gl) in sorted(stat.items(), key=lambda kv: -(kv[1][1] + kv[1][2]))[:8]:
        gl.sort()
        print(f"    {n:48s} {c:7d} x  kernel {d / c / 1e3:8.1f} us   gap before it: mean {g / c / 1e3:7.1f} us, median {gl[len(gl) // 2] / 1e3:6.1f}, p90 {gl[int(0.9 * len(gl))] / 1e3:6.1f}")
