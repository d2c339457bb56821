#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O
cd $R
for x in 1 3 5 9 7 15; do echo "== RWKV_TILE_XCD=$x"; RWKV_TILE_XCD=$x SHAPES=10 TS=1024 timeout 300 python scripts/tile_bench.py 2>&1 | grep -v "^w1\|wo-"; done > $O/tile16.log 2>&1
cat $O/tile16.log
