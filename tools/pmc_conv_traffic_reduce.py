#!/usr/bin/env python3
"""Reduce the rocprofv3 --pmc passes of tools/pmc_conv_traffic.sh (or an already committed raw file) to the convolution kernels'
HBM traffic per launch, with gfx950's FETCH_SIZE correction applied PER INSTANTIATION by tools/pmc_widths.py (the library
classifies its own kernels; an unknown kernel name stops the script).

    python3 tools/pmc_conv_traffic_reduce.py --passes /tmp/pmc_FETCH_SIZE /tmp/pmc_WRITE_SIZE [/tmp/pmc_TCP_TCC_READ_REQ_sum ...] \
            [--launch-bytes conv_bytes.json] --out-dir gpurun_out [--tag r06]
    python3 tools/pmc_conv_traffic_reduce.py --raw profiles/r05_conv_traffic_raw.json --out-dir /tmp/x      # re-reduce old counters

Outputs: <out>/conv_traffic_raw.json (per kernel name: launches + summed counters), <out>/conv_traffic.json (the record bench.py
prints as roofline.traffic) and a per-instantiation table on stdout."""
import argparse
import collections
import csv
import glob
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import pmc_widths  # noqa: E402

L2_REQ_BYTES = 128   # one TCP -> TCC read request = one 128-byte line (profiles/r05_gemm_tile_ab.txt section 5: 524 k requests = 67 MB)


def read_passes(dirs):
    """{counter: {kernel base name: [value per dispatch, in dispatch order]}} for the convolution kernels."""
    out = {}
    for d in dirs:
        files = sorted(glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True))
        if not files:
            raise SystemExit(f"no counter_collection.csv under {d}")
        rows = collections.defaultdict(dict)   # dispatch id -> {counter: value}, name
        for r in csv.DictReader(open(files[-1])):
            if not pmc_widths.is_conv(r["Kernel_Name"]):
                continue
            did = int(r["Dispatch_Id"])
            rows[did]["name"] = pmc_widths.base_name(r["Kernel_Name"])
            rows[did][r["Counter_Name"]] = rows[did].get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
        for did in sorted(rows):
            row = rows[did]
            for c, v in row.items():
                if c == "name":
                    continue
                out.setdefault(c, []).append((row["name"], v))
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--passes", nargs="*", default=[])
    ap.add_argument("--raw", default=None, help="an already reduced raw file (per-kernel sums) instead of --passes")
    ap.add_argument("--launch-bytes", default=None, help="CCVS_DUMP_CONV_BYTES output of the same command: algorithmic bytes per launch, in order")
    ap.add_argument("--out-dir", default="gpurun_out")
    ap.add_argument("--key", default="bair-b16-bf16x3")
    ap.add_argument("--note", default="")
    args = ap.parse_args()

    per_launch = None
    if args.raw:
        raw = json.load(open(args.raw))
        for c in raw:      # (round <= 5 raw files carry rocprofv3's names: "void name<...>(args)")
            for fld in ("per_kernel", "per_kernel_KiB", "launches"):
                if fld in raw[c]:
                    raw[c][fld] = {pmc_widths.base_name(k): v for k, v in raw[c][fld].items()}
    else:
        seq = read_passes(args.passes)
        raw = {}
        for c, lst in seq.items():
            tot, n = collections.defaultdict(float), collections.defaultdict(int)
            for name, v in lst:
                tot[name] += v
                n[name] += 1
            raw[c] = {"per_kernel": dict(tot), "launches": dict(n), "total": sum(tot.values()), "total_launches": len(lst)}
        if args.launch_bytes and "FETCH_SIZE" in seq:
            recs = json.load(open(args.launch_bytes))
            names = [nm for nm, _ in seq["FETCH_SIZE"]]
            if len(recs) == len(names):
                per_launch = collections.defaultdict(lambda: [0.0, 0.0])
                for nm, r in zip(names, recs):
                    per_launch[nm][0] += r["bytes"]
                    per_launch[nm][1] += r["side_bytes"]
            else:
                print(f"# {len(recs)} timed launches against {len(names)} counter rows: no per-instantiation algorithmic bytes", file=sys.stderr)
    # (round <= 5 raw files: keys per_kernel_KiB / total_KiB)
    def per_kernel(c):
        r = raw[c]
        return r.get("per_kernel", r.get("per_kernel_KiB"))

    fetch_k, write_k = per_kernel("FETCH_SIZE"), per_kernel("WRITE_SIZE")
    launches = raw["FETCH_SIZE"]["launches"]
    l2 = per_kernel("TCP_TCC_READ_REQ_sum") if "TCP_TCC_READ_REQ_sum" in raw else None
    table, lo_tot, hi_tot, raw_tot = [], 0.0, 0.0, 0.0
    for name in sorted(fetch_k, key=lambda k: -fetch_k[k]):
        lo, hi = pmc_widths.fetch_scale(name)          # raises on a kernel nobody classifies
        f_raw = fetch_k[name] * 1024
        raw_tot += f_raw
        lo_tot += f_raw * lo
        hi_tot += f_raw * hi
        row = {"kernel": name, "launches": launches[name], "read_bytes_per_lane": pmc_widths.read_bytes_per_lane(name),
               "fetch_raw_GB": f_raw / 1e9, "fetch_GB": f_raw * hi / 1e9, "write_GB": write_k.get(name, 0.0) * 1024 / 1e9}
        if l2:
            row["l2_read_GB"] = l2.get(name, 0.0) * L2_REQ_BYTES / 1e9
        if per_launch and name in per_launch:
            row["algorithmic_GB"], row["side_operands_GB"] = per_launch[name][0] / 1e9, per_launch[name][1] / 1e9
            row["hbm_over_algorithmic"] = (row["fetch_GB"] + row["write_GB"]) / row["algorithmic_GB"]
            row["hbm_over_algorithmic_with_side_operands"] = (row["fetch_GB"] + row["write_GB"]) / (row["algorithmic_GB"] + row["side_operands_GB"])
        table.append(row)
    assert lo_tot == hi_tot, "a convolution kernel with an uncalibrated access width"
    write = sum(write_k.values()) * 1024
    n = raw["FETCH_SIZE"]["total_launches"]
    rec = {"bytes_per_launch": (hi_tot + write) / n, "fetch_bytes_per_launch": hi_tot / n, "write_bytes_per_launch": write / n,
           "fetch_raw_bytes_per_launch": raw_tot / n, "launches": n,
           "per_instantiation": table,
           "source": "rocprofv3 --kernel-trace --pmc FETCH_SIZE / WRITE_SIZE (separate passes) -- python3 tools/decode_only.py 16 "
                     "(tools/pmc_conv_traffic.sh -> tools/pmc_conv_traffic_reduce.py); KiB counters x1024 summed over the conv launches of one batch; "
                     "FETCH_SIZE x2 for every instantiation the LIBRARY reports as reading 16 bytes per lane (ccvs_conv_fetch_bytes_per_lane: the "
                     "VEC staging modes NTY 1 | 3 and the packed-input modes NTY -8 | -83), x1 for the dword-staging ones (NTY 0 | -2, the synchronous "
                     "kernel), per the gfx950 correction of MI355X_MICROARCH.md; an unclassified kernel name stops the reduction"}
    if per_launch:
        a = sum(v[0] for v in per_launch.values())
        sd = sum(v[1] for v in per_launch.values())
        rec["algorithmic_bytes_per_launch"] = a / n
        rec["side_operand_bytes_per_launch"] = sd / n
        rec["hbm_over_algorithmic"] = (hi_tot + write) / a
        rec["hbm_over_algorithmic_with_side_operands"] = (hi_tot + write) / (a + sd)
    if args.note:
        rec["note"] = args.note
    os.makedirs(args.out_dir, exist_ok=True)
    json.dump(raw, open(os.path.join(args.out_dir, "conv_traffic_raw.json"), "w"), indent=1)
    json.dump({args.key: rec}, open(os.path.join(args.out_dir, "conv_traffic.json"), "w"), indent=1)
    hdr = f"{'kernel':52s} {'n':>5s} {'B/lane':>6s} {'fetch raw':>10s} {'fetch':>9s} {'write':>9s}" + (f" {'L2 reads':>9s}" if l2 else "") + \
          (f" {'algorithmic':>11s} {'side ops':>9s} {'HBM/alg':>8s} {'/(alg+side)':>11s}" if per_launch else "")
    print(hdr + "   (GB over the launches of one batch)")
    for r in table:
        line = f"{r['kernel']:52s} {r['launches']:5d} {r['read_bytes_per_lane']:6d} {r['fetch_raw_GB']:10.2f} {r['fetch_GB']:9.2f} {r['write_GB']:9.2f}"
        if l2:
            line += f" {r['l2_read_GB']:9.2f}"
        if "algorithmic_GB" in r:
            line += f" {r['algorithmic_GB']:11.2f} {r['side_operands_GB']:9.2f} {r['hbm_over_algorithmic']:8.3f} {r['hbm_over_algorithmic_with_side_operands']:11.3f}"
        print(line)
    print(f"TOTAL per launch: fetch {hi_tot / n / 1e6:.1f} MB (raw {raw_tot / n / 1e6:.1f}) + write {write / n / 1e6:.1f} MB = {(hi_tot + write) / n / 1e6:.1f} MB over {n} launches"
          + (f"; algorithmic {rec['algorithmic_bytes_per_launch'] / 1e6:.1f} MB (+ {rec['side_operand_bytes_per_launch'] / 1e6:.1f} MB side operands): "
             f"x{rec['hbm_over_algorithmic']:.3f} / x{rec['hbm_over_algorithmic_with_side_operands']:.3f}" if per_launch else ""))


if __name__ == "__main__":
    main()
