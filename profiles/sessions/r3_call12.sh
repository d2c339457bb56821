#!/bin/bash
# round 3, session 12: wide fused mix — how many blocks (prefill A/B)
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O
cd $R
for nb in 256 640 1024 2048 4096; do for C in 2048 512 256; do echo -n "V6WIDE_BLOCKS=$nb "; RWKV_V6WIDE_BLOCKS=$nb timeout 200 python scripts/prefill_probe.py v6-3b 1 32 256 $C 2>&1 | tail -1; done; done > $O/r3_v6wide_ab.log 2>&1
for nb in 256 1024 2048; do echo -n "V6WIDE_BLOCKS=$nb "; RWKV_V6WIDE_BLOCKS=$nb timeout 300 python scripts/prefill_probe.py v6-7b 0 8 2048 1024 2>&1 | tail -1; done >> $O/r3_v6wide_ab.log 2>&1
cat $O/r3_v6wide_ab.log
