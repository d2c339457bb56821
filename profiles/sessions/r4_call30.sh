#!/bin/bash
# round 4, session 30: L2 hit rate and memory-side bytes of the 256-row r/k/v/g-sized launch on the 64-token pipelined tile (shape 11) and of the 2048-row
# launch on the 128-token tile (shape 10): is the launch bound behind the L2?
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
export TMPDIR=/tmp
cd /tmp
for pass in "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "FETCH_SIZE WRITE_SIZE" "TCP_TCC_READ_REQ_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum"; do
  tag=$(echo $pass | cut -d' ' -f1)
  rm -rf $O/pmc_tile_$tag
  SHAPES=11 TS=256 timeout 300 rocprofv3 --kernel-trace --pmc $pass --output-format csv -d $O/pmc_tile_$tag -o p -- python $R/scripts/tile_by_rows.py > $O/pmc_tile_$tag.log 2>&1
  SHAPES=10 TS=2048 timeout 300 rocprofv3 --kernel-trace --pmc $pass --output-format csv -d $O/pmc_tile2k_$tag -o p -- python $R/scripts/tile_by_rows.py > $O/pmc_tile2k_$tag.log 2>&1
done
python - $O <<'PY' | tee $O/r4c30_tile_l2_counters.log
import csv, glob, collections, sys
O = sys.argv[1]
for run in ("pmc_tile_", "pmc_tile2k_"):
    acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(collections.Counter)
    for f in glob.glob(O + "/" + run + "*/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"]
            if "gemm_tile3" not in k: continue
            key = (k[:40], r.get("Grid_Size", ""))
            acc[key][r["Counter_Name"]] += float(r["Counter_Value"]); n[key][r["Counter_Name"]] += 1
    print("==", run)
    for key, c in acc.items():
        print(key, {name: round(v / n[key][name], 1) for name, v in c.items()}, "launches", max(n[key].values()))
PY
