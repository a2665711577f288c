cd /tmp && export TMPDIR=/tmp
for p8 in 1 0; do
  rm -rf /tmp/prof; CCVS_CONV_P8=$p8 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof -o b -- python3 $GRAFT_REPO_ROOT/tools/decode_only.py 16 > /tmp/b.log 2>&1
  echo "== CCVS_CONV_P8=$p8"; tail -2 /tmp/b.log
  python3 - <<PY
import csv,glob
f=glob.glob("/tmp/prof/*kernel_stats.csv")[0]
rows=list(csv.DictReader(open(f)))
tot=0
for r in rows:
    if "conv2d" in r["Name"] or "tap_shift" in r["Name"]:
        print(r["Name"][:60].ljust(60), r["Calls"].rjust(6), "%8.1f ms"%(float(r["TotalDurationNs"])/1e6))
        tot+=float(r["TotalDurationNs"])/1e6
print("conv total %.1f ms"%tot)
PY
done
