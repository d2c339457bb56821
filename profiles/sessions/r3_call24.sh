#!/bin/bash
# round 3, session 24: Wo (320 tiles at 2048 rows of the 3 B model) on the pipelined tile kernel
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O
cd $R
: > $O/r3_tile3_wo.log
for rep in 1 2; do
  for v in "65 400" "60 300" "30 128"; do
    set -- $v
    for cfg in "v6-3b 1 32 256 2048" "v6-3b 1 32 256 1024" "v7-2.9b 2 32 256 2048" "v6-7b 0 8 2048 1024"; do
      RWKV_TILE3_FILL=$1 RWKV_TILE3_MIN_TILES=$2 timeout 300 python scripts/prefill_probe.py $cfg 2>&1 | tail -1 | sed "s/^/FILL=$1 MIN=$2 /" >> $O/r3_tile3_wo.log
    done
  done
done
cat $O/r3_tile3_wo.log
