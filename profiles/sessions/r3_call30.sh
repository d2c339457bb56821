#!/bin/bash
# round 3, session 30: V7 second-stage LoRA launch (K = 64..320) at 2048 / 1024 rows: the 128x64 GLDS shape (grid rule) against the 64x64 shapes (variant build -DRWKV_SHAPE7_BY_K)
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O
cd $R
: > $O/r3_shape7_by_k.log
for rep in 1 2; do
  for lib in librwkv_hip.so librwkv_hip_byk.so; do
    for cfg in "v7-2.9b 2 32 256 2048" "v7-2.9b 2 32 256 1024"; do
      RWKV_HIP_LIB=$R/ai00_server_amd/$lib timeout 300 python scripts/prefill_probe.py $cfg 2>&1 | tail -1 | sed "s/^/$lib /" >> $O/r3_shape7_by_k.log
    done
  done
done
cat $O/r3_shape7_by_k.log
