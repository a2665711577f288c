#!/usr/bin/env python3
"""Kernel trace of the pipelined run: how long a token kernel takes by WHAT runs beside it on the decode queues.
    python tools/r05/trace_beside.py <dir with *_kernel_trace.csv> [lo hi]"""
import bisect
import csv
import glob
import re
import sys
from collections import defaultdict

csv.field_size_limit(1 << 30)
root = sys.argv[1]
lo, hi = (float(sys.argv[2]), float(sys.argv[3])) if len(sys.argv) > 3 else (0.3, 0.6)


def short(name):
    m = re.match(r"(?:void )?(\w+)(<[^>]*>)?", name)
    return (m.group(1) + (m.group(2) or "")) if m else name[:40]


rows = []
for f in glob.glob(root + "/**/*kernel_trace.csv", recursive=True):
    with open(f) as fh:
        for r in csv.DictReader(fh):
            n = short(r["Kernel_Name"])
            if n.startswith("gemm16"):   # the GEMM of a decode step by its grid: workgroups in x (column tiles) . y (row blocks) . z (K slices)
                n = "gemm16 grid %d.%d.%d" % (int(r["Grid_Size_X"]) // int(r["Workgroup_Size_X"]), int(r["Grid_Size_Y"]), int(r["Grid_Size_Z"]))
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Queue_Id"], n))
rows.sort()
t0, t1 = rows[0][0], rows[-1][1]
w0, w1 = t0 + lo * (t1 - t0), t0 + hi * (t1 - t0)
rows = [r for r in rows if w0 <= r[0] <= w1]
byq = defaultdict(list)
for r in rows:
    byq[r[2]].append(r)
tokq = [q for q, rr in byq.items() if sum(1 for r in rr if r[3].startswith("gemm16")) > len(rr) // 2]
decq = [q for q, rr in byq.items() if q not in tokq and sum(1 for r in rr if r[3].startswith("conv2d")) > len(rr) // 4]
print(f"window {(w1 - w0) / 1e9:.2f} s; token queues {tokq}, decode queues {decq}")
starts = {q: [r[0] for r in byq[q]] for q in decq + tokq}


def active(q, t):
    i = bisect.bisect_right(starts[q], t) - 1
    if i >= 0 and byq[q][i][1] >= t:
        return byq[q][i][3]
    return "-"


shapes = defaultdict(lambda: defaultdict(lambda: [0, 0.0]))
for q in tokq:
    for s_, e_, _, n in byq[q]:
        if n.startswith("gemm16 grid"):
            mid = (s_ + e_) // 2
            busy = sum(1 for d in decq if active(d, mid).startswith("conv2d"))
            a = shapes[n][busy]
            a[0] += 1; a[1] += e_ - s_
print("\ngemm16 by grid (column tiles . row blocks . K slices): launches x mean us, with 0 / 1 / 2 decode queues inside a convolution")
for n, by in sorted(shapes.items(), key=lambda kv: -sum(a[0] for a in kv[1].values())):
    print(f"  {n:28s} " + "   ".join(f"{k}: {by[k][0]:6d} x {by[k][1] / max(by[k][0], 1) / 1e3:6.1f}" for k in (0, 1, 2)))

for kind in ("gemm16", "attention_decode"):
    agg = defaultdict(lambda: [0, 0.0])
    for q in tokq:
        other = [o for o in tokq if o != q]
        for s, e, _, n in byq[q]:
            if not n.startswith(kind):
                continue
            mid = (s + e) // 2
            key = tuple(sorted(active(d, mid) for d in decq)) + tuple("tok:" + ("busy" if active(o, mid) != "-" else "-") for o in other)
            a = agg[key]
            a[0] += 1; a[1] += e - s
    tot = sum(a[0] for a in agg.values())
    print(f"\n{kind}: {tot} launches, mean {sum(a[1] for a in agg.values()) / max(tot, 1) / 1e3:.1f} us; by what runs beside (decode queues..., other token queue)")
    for key, (c, d) in sorted(agg.items(), key=lambda kv: -kv[1][0])[:22]:
        print(f"  {c:6d} x {d / c / 1e3:7.1f} us   " + " | ".join(key))
