#!/usr/bin/env python3
"""Top kernels of a rocprofv3 --stats run.   python tools/kernel_stats_top.py <dir or kernel_stats.csv> [n]"""
import csv
import glob
import os
import sys

path = sys.argv[1]
if os.path.isdir(path):
    path = sorted(glob.glob(os.path.join(path, "**", "*kernel_stats.csv"), recursive=True))[0]
n = int(sys.argv[2]) if len(sys.argv) > 2 else 16
rows = list(csv.DictReader(open(path)))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"]))[:n]:
    print("%-64s calls=%7s tot_ms=%9.1f avg_us=%9.1f %5.1f%%" % (r["Name"][:64], r["Calls"], float(r["TotalDurationNs"]) / 1e6,
                                                                   float(r["AverageNs"]) / 1e3, 100 * float(r["TotalDurationNs"]) / tot))
