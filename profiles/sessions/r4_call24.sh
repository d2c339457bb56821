#!/bin/bash
# round 4, session 24: where a pipelined-tile stage spends its time — ablations of the hand-scheduled build (results are wrong by construction, timing only):
# no s_barrier, no X DMA, no weight refill, no fragment reads
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
O=$R/gpurun_out
mkdir -p $O
: > $O/r4c24_tile3_ablation.log
for lib in t3pipe abl_NOBAR abl_NODMA abl_NOW abl_NOLDS; do
  export RWKV_HIP_LIB=$R/ai00_server_amd/librwkv_hip_$lib.so
  echo "== $lib" >> $O/r4c24_tile3_ablation.log
  SHAPES=10,11 TS=256,2048 timeout 300 python scripts/tile_by_rows.py 2>&1 | grep -v "^#" >> $O/r4c24_tile3_ablation.log
done
cat $O/r4c24_tile3_ablation.log
