#!/bin/bash
# Counter evidence for the decode GEMM's block tile: L2 read requests of the CUs (TCP_TCC_READ_REQ_sum), HBM-side fetch (FETCH_SIZE,
# KiB) and cycles (GRBM_GUI_ACTIVE) per gemm16 launch of a 64-row decode step, 2 x 2 blocks per workgroup (CCVS_GEMM_TILE2=1)
# against one (0).  Eager steps (the counter pass cannot follow graph replays), separate --pmc passes, kernel trace only.
#   gpurun -- bash tools/r05/pmc_gemm_tile.sh [rows] [tokens]
ROWS=${1:-64}; TOK=${2:-100}
cd /tmp && export TMPDIR=/tmp
export CCVS_PROBE_EAGER=1
for tile in 1 0; do
  for c in TCP_TCC_READ_REQ_sum FETCH_SIZE GRBM_GUI_ACTIVE; do
    rm -rf /tmp/pg_${tile}_$c
    for try in 1 2 3; do
      CCVS_GEMM_TILE2=$tile timeout 300 rocprofv3 --kernel-trace --pmc $c -d /tmp/pg_${tile}_$c -o p --output-format csv -- python3 $GRAFT_REPO_ROOT/tools/token_step_probe.py $TOK $ROWS > /tmp/pg_${tile}_$c.log 2>&1 && break
    done
  done
done
python3 - <<'PY'
import csv, glob, collections
for tile in (1, 0):
    cnt = collections.defaultdict(lambda: collections.defaultdict(float))
    n = collections.defaultdict(int)
    dur = collections.defaultdict(list)
    for d in glob.glob(f"/tmp/pg_{tile}_*/"):
        for f in glob.glob(d + "**/*counter_collection.csv", recursive=True):
            for r in csv.DictReader(open(f)):
                if "gemm16_kernel" not in r["Kernel_Name"]:
                    continue
                key = "%d.%d.%d" % (int(r["Grid_Size_X"]) // int(r["Workgroup_Size_X"]), int(r["Grid_Size_Y"]), int(r["Grid_Size_Z"])) if "Grid_Size_X" in r else "all"
                cnt[key][r["Counter_Name"]] += float(r["Counter_Value"])
                if r["Counter_Name"] == "GRBM_GUI_ACTIVE":
                    n[key] += 1
    print(f"CCVS_GEMM_TILE2={tile}: gemm16 launches by grid (column tiles . row blocks . K slices), per launch")
    for key in sorted(cnt, key=lambda k: -n[k]):
        c = cnt[key]; m = max(n[key], 1)
        print(f"   grid {key:10s} {n[key]:6d} x   L2 read requests {c['TCP_TCC_READ_REQ_sum'] / m / 1e3:9.1f} k   HBM fetch (x2) {2 * c['FETCH_SIZE'] * 1024 / m / 1e6:7.2f} MB   "
              f"{c['GRBM_GUI_ACTIVE'] / 8 / m / 1e3:7.1f} kcycles")
PY
