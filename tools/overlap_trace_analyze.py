#!/usr/bin/env python3
"""Reads a rocprofv3 kernel trace (CSV) of a pipelined bench run and reports how the token stream and the decoder stream
actually share the chip: busy time of each, their overlap, and the launch-to-launch gaps of the token kernels split by
whether a decoder kernel was running.   python tools/overlap_trace_analyze.py <kernel_trace.csv>"""
import csv
import sys
from bisect import bisect_right

TOKEN = ("gemm16", "attention_decode", "sample_topk", "gpt_embed")
rows = []
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", "?")))
rows.sort()
tok = [r for r in rows if any(t in r[2] for t in TOKEN)]
dec = [r for r in rows if not any(t in r[2] for t in TOKEN)]
print(f"{len(rows)} kernels: {len(tok)} token-loop, {len(dec)} others; queues token {sorted(set(r[3] for r in tok))} others {sorted(set(r[3] for r in dec))}")


def union(iv):
    out = []
    for s, e, *_ in sorted(iv):
        if out and s <= out[-1][1]:
            out[-1][1] = max(out[-1][1], e)
        else:
            out.append([s, e])
    return out


ud, ut = union(dec), union(tok)
busy_d, busy_t = sum(e - s for s, e in ud), sum(e - s for s, e in ut)
# overlap of the two unions
i = j = 0
ov = 0
while i < len(ud) and j < len(ut):
    s, e = max(ud[i][0], ut[j][0]), min(ud[i][1], ut[j][1])
    if s < e:
        ov += e - s
    if ud[i][1] < ut[j][1]:
        i += 1
    else:
        j += 1
t0, t1 = rows[0][0], max(r[1] for r in rows)
print(f"span {1e-6 * (t1 - t0):.0f} ms; decoder-stream busy {1e-6 * busy_d:.0f} ms, token-stream busy {1e-6 * busy_t:.0f} ms, both at once {1e-6 * ov:.0f} ms")
starts = [s for s, e in ud]


def dec_running(t):
    k = bisect_right(starts, t) - 1
    return k >= 0 and ud[k][1] > t


gaps = {True: [], False: []}
durs = {True: [], False: []}
for a, b in zip(tok, tok[1:]):
    gap = b[0] - a[1]
    if 0 <= gap < 200000:           # inside one run of decode steps
        gaps[dec_running(a[1])].append(gap)
    durs[dec_running(a[0])].append(a[1] - a[0])
for flag in (False, True):
    g, d = sorted(gaps[flag]), durs[flag]
    if g:
        print(f"token kernels while the decoder stream is {'BUSY' if flag else 'idle'}: {len(g)} gaps, mean {1e-3 * sum(g) / len(g):.2f} us, "
              f"median {1e-3 * g[len(g) // 2]:.2f}, p90 {1e-3 * g[int(0.9 * len(g))]:.2f} us; kernel duration mean {1e-3 * sum(d) / max(len(d), 1):.2f} us")
# decoder kernels: duration of the big convolution chunks when token kernels run beside them
for name in ("pc_kernel<32, 4, 3>", "backwarp_kernel"):
    d = [e - s for s, e, n, q in dec if name in n]
    if d:
        print(f"{name}: {len(d)} launches, mean {1e-3 * sum(d) / len(d):.1f} us")

# Is a delayed token kernel released by the END of a decoder-stream kernel?  For every token launch that waited > 10 us:
# time from the nearest preceding decoder-kernel end to the token kernel's start, and what was running on the decoder stream.
ends = sorted((e, s, n) for s, e, n, q in dec)
end_times = [e for e, s, n in ends]
deltas, same_q = [], 0
examples = []
for a, b in zip(tok, tok[1:]):
    gap = b[0] - a[1]
    if 10000 < gap < 200000:
        k = bisect_right(end_times, b[0]) - 1
        if k >= 0:
            d = b[0] - end_times[k]
            deltas.append(d)
            if len(examples) < 12:
                running = [n.split("(")[0][:40] for s, e, n, q in dec[max(0, bisect_right([r[0] for r in dec], b[0]) - 6):bisect_right([r[0] for r in dec], b[0])] if e > b[0]]
                examples.append((gap, d, ends[k][2].split("(")[0][:40], b[3], running))
if deltas:
    ds = sorted(deltas)
    print(f"{len(ds)} token launches waited > 10 us; time from the last decoder-kernel END before their start: median {1e-3 * ds[len(ds) // 2]:.2f} us, "
          f"p25 {1e-3 * ds[len(ds) // 4]:.2f}, p75 {1e-3 * ds[3 * len(ds) // 4]:.2f} us; share below 3 us: {sum(1 for d in ds if d < 3000) / len(ds):.2f}")
    for ex in examples:
        print("   gap %.1f us, started %.2f us after the end of %s (token queue %s); decoder kernels still running then: %s" % (1e-3 * ex[0], 1e-3 * ex[1], ex[2], ex[3], ex[4]))
qpairs = {}
for s, e, n, q in rows:
    key = ("token" if any(t in n for t in TOKEN) else "other", q)
    qpairs[key] = qpairs.get(key, 0) + 1
print("launches per (class, queue):", sorted(qpairs.items()))
