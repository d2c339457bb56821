#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O
cd $R
TS=512,1024 SHAPES=4,7,10,11,12 timeout 400 python scripts/tile_bench.py > $O/tile8.log 2>&1; echo "tile rc=$?"
for V in 10 12; do
  echo "== RWKV_TILE_SHAPE=$V" >> $O/tile8.log
  RWKV_TILE_SHAPE=$V timeout 200 python scripts/prefill_probe.py v6-3b 1 32 256 512 >> $O/tile8.log 2>&1
done
