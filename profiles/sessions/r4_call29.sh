#!/bin/bash
# round 4, session 29: what a K split of the 256-row non-linear launches would buy — the r/k/v/g-sized launch (10240 x 2560) on the 64-token pipelined
# tile as 1 / 2 / 3 / 4 K copies writing partial slabs (no consumer sums them here: an upper bound on the gain)
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
O=$R/gpurun_out
: > $O/r4c29_ksplit_256_rows.log
for k in 1 2 3 4; do
  echo "== RWKV_BENCH_KSB=$k" >> $O/r4c29_ksplit_256_rows.log
  RWKV_BENCH_KSB=$k SHAPES=11,10 TS=256,384,512 timeout 300 python scripts/tile_by_rows.py 2>&1 | grep -v "^#" >> $O/r4c29_ksplit_256_rows.log
done
cat $O/r4c29_ksplit_256_rows.log
