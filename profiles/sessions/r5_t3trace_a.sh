#!/bin/bash
# where a stage of the pipelined tile kernel spends its cycles (dev build -DRWKV_T3_TRACE: s_memtime stamps, printed by a few waves)
cd $GRAFT_REPO_ROOT
for lib in t3trace t3trace2; do
  for shape in 11 12 10; do
    echo "== $lib shape $shape T=256"
    RWKV_HIP_LIB=$PWD/ai00_server_amd/librwkv_hip_$lib.so SHAPES=$shape TS=256 timeout 100 python scripts/tile_by_rows.py 2>&1 | sort | uniq -c | sort -rn | head -30
  done
done
echo "== $lib shape 10 T=2048"
RWKV_HIP_LIB=$PWD/ai00_server_amd/librwkv_hip_t3trace.so SHAPES=10 TS=2048 timeout 100 python scripts/tile_by_rows.py 2>&1 | sort | uniq -c | sort -rn | head -30
