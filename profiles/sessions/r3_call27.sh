#!/bin/bash
# round 3, session 27: all-NF4 launches on 128-k chunks from 512 tiles: prefill A/B on V7-2.9B NF4
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O
cd $R
: > $O/r3_nf4_kc128.log
for rep in 1 2 3; do
  for m in 512 1000000; do
    for cfg in "v7-2.9b 2 32 256 256" "v7-2.9b 2 32 256 512"; do
      RWKV_NF4_KC128_MIN=$m timeout 300 python scripts/prefill_probe.py $cfg 2>&1 | tail -1 | sed "s/^/NF4_KC128_MIN=$m /" >> $O/r3_nf4_kc128.log
    done
  done
done
cat $O/r3_nf4_kc128.log
