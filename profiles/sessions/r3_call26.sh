#!/bin/bash
# round 3, session 26: 128-row x 64-token tile shapes (8 waves) against the 64x64 shapes on steps of 256 / 512 rows
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O
cd $R
timeout 600 python scripts/tile_bench3.py > $O/r3_tile_128x64.log 2>&1
cat $O/r3_tile_128x64.log
for sh in 11 12; do
  for cfg in "v6-3b 1 32 256 256" "v7-2.9b 2 32 256 256"; do
    RWKV_TILE_SHAPE=$sh timeout 300 python scripts/prefill_probe.py $cfg 2>&1 | tail -1 | sed "s/^/TILE_SHAPE=$sh /" | tee -a $O/r3_tile_128x64.log
  done
done
