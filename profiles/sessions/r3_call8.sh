#!/bin/bash
# round 3, session 8: wide form of the fused V6 mix for 17..32 rows (40 blocks instead of 100), K-split rule check
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O
cd $R
timeout 600 python scripts/ab_bench.py "base::" "wide16:RWKV_V6MIX_WIDE_ABOVE=16:" "base2::" "wide16b:RWKV_V6MIX_WIDE_ABOVE=16:" > $O/r3_ab8.log 2>&1; cat $O/r3_ab8.log
for ks in 0 1; do for C in 512 256; do echo -n "KSPLIT=$ks "; RWKV_TILE_KSPLIT=$ks timeout 200 python scripts/prefill_probe.py v6-3b 1 32 256 $C 2>&1 | tail -1; done; done > $O/r3_ksplit_ab3.log 2>&1; cat $O/r3_ksplit_ab3.log
