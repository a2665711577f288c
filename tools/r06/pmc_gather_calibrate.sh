#!/bin/bash
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pg; timeout 300 rocprofv3 --kernel-trace --pmc $c -d /tmp/pg -o p --output-format csv -- python3 $GRAFT_REPO_ROOT/tools/r06/gather_calibrate.py > /tmp/pg.log 2>&1
  grep ALG_ /tmp/pg.log
  python3 - <<PY
import csv, glob
f = glob.glob("/tmp/pg/**/*counter_collection.csv", recursive=True)
rows = [r for r in csv.DictReader(open(f[0])) if "backwarp" in r["Kernel_Name"] and r["Counter_Name"] == "$c"]
by = {}
for r in rows: by[int(r["Dispatch_Id"])] = by.get(int(r["Dispatch_Id"]), 0.0) + float(r["Counter_Value"])
vals = [by[k] for k in sorted(by)]
print("$c per launch (KiB):", [round(v) for v in vals], " kernel:", rows[0]["Kernel_Name"].split("(")[0] if rows else None, " -> bytes", [round(v * 1024 / 1e6, 1) for v in vals], "MB")
PY
done
