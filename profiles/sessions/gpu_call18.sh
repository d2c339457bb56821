#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O
cd $R
: > $O/pf18.log
for F in 80 60; do
  echo "== RWKV_TILE3_FILL=$F" >> $O/pf18.log
  for C in 1024 2048; do RWKV_TILE3_FILL=$F timeout 200 python scripts/prefill_probe.py v6-3b 1 32 256 $C 2>&1 | tail -1 >> $O/pf18.log; done
  RWKV_TILE3_FILL=$F timeout 300 python scripts/prefill_probe.py v6-7b 0 8 1024 1024 2>&1 | tail -1 >> $O/pf18.log
done
cat $O/pf18.log
