#!/bin/bash
# Where does the fp32-input 3 x 3 instantiation (conv2d_bf16x3_pc_kernel<32, 4, 3, 2, 1>: 1.29 x its algorithmic bytes through HBM,
# profiles/r06_conv_traffic_table.txt) fetch more than it needs?  Its VEC staging reads the halo tile widened to 4-pixel boundaries:
# 40 x 10 pixels per 32 x 8 output pixels = 1.5625 x the input if no neighbour's fetch is ever an L2 hit.  Per shape: FETCH_SIZE
# (x2: 16 bytes per lane), L2 hits / misses, L2 read requests -- with the XCD-aware tile order and with the plain one.
cd /tmp && export TMPDIR=/tmp
IFS=';' read -ra SHAPE_LIST <<< "${SHAPES:-195 128 3 256 48;128 64 3 256 48 p8}"     # SHAPES="Cin Cout k H N [p8] [pre=K];..."
for SHAPE in "${SHAPE_LIST[@]}"; do
  for XCD in ${XCDS:-1 0}; do
    echo "== shape (Cin Cout k H N) $SHAPE   CCVS_CONV_XCD=$XCD"
    for c in "FETCH_SIZE" "TCC_HIT_sum TCC_MISS_sum"; do
      rm -rf /tmp/pmc; CCVS_CONV_XCD=$XCD timeout 300 rocprofv3 --kernel-trace --pmc $c -d /tmp/pmc -o p --output-format csv -- python3 $GRAFT_REPO_ROOT/tools/conv_one.py $SHAPE > /tmp/pmc.log 2>&1
      SHAPE="$SHAPE" python3 - <<'PY'
import csv, glob, collections, os
sh = os.environ["SHAPE"].split()
cin, cout, k, h, n = [int(v) for v in sh[:5]]
f = glob.glob("/tmp/pmc/**/*counter_collection.csv", recursive=True)
if not f:
    print("no counter file"); raise SystemExit
rows = list(csv.DictReader(open(f[0])))
by = collections.defaultdict(lambda: collections.defaultdict(float))
for r in rows:
    if "conv2d_bf16x3" in r["Kernel_Name"]:
        by[int(r["Dispatch_Id"])][r["Counter_Name"]] += float(r["Counter_Value"]); by[int(r["Dispatch_Id"])]["name"] = r["Kernel_Name"].split("(")[0]
last = by[max(by)]          # the last launch = the measured shape (packed input: the first ones build it)
alg_in = 4.0 * n * cin * h * h
for c, v in last.items():
    if c == "name": continue
    extra = ""
    if c == "FETCH_SIZE": extra = f"  -> x2 = {2 * v * 1024 / 1e6:.1f} MB = {2 * v * 1024 / alg_in:.3f} x the input ({alg_in / 1e6:.1f} MB)"
    if c == "TCP_TCC_READ_REQ_sum": extra = f"  -> x128 B = {v * 128 / 1e6:.1f} MB = {v * 128 / alg_in:.2f} x the input"
    print(f"   {last['name']}  {c} = {v:.4g}{extra}")
if "TCC_HIT_sum" in last: print(f"   L2 hit rate {last['TCC_HIT_sum'] / (last['TCC_HIT_sum'] + last['TCC_MISS_sum']):.3f}")
PY
    done
  done
done
