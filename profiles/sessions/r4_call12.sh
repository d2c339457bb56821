#!/bin/bash
# round 4, session 12: shape 11 on the non-linear launches only, by RWKV_TILE3_64_MAX_T, chunk 256 / 512 / 1024, four engines
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O
cd $R
: > $O/r4c12_tile3_64_nonlinear.log
for mt in 0 256 512 1024; do
  export RWKV_TILE3_64_MAX_T=$mt
  echo "== RWKV_TILE3_64_MAX_T=$mt" >> $O/r4c12_tile3_64_nonlinear.log
  for chunk in 256 512 1024; do
    timeout 300 python scripts/prefill_probe.py v6-3b 1 32 256 $chunk 2>&1 | tail -1 >> $O/r4c12_tile3_64_nonlinear.log
    timeout 300 python scripts/prefill_probe.py v7-2.9b 2 32 256 $chunk 2>&1 | tail -1 >> $O/r4c12_tile3_64_nonlinear.log
    timeout 300 python scripts/prefill_probe.py v6-3b 0 32 256 $chunk 2>&1 | tail -1 >> $O/r4c12_tile3_64_nonlinear.log
    timeout 300 python scripts/prefill_probe.py v6-7b 0 8 1024 $chunk 2>&1 | tail -1 >> $O/r4c12_tile3_64_nonlinear.log
  done
done
cat $O/r4c12_tile3_64_nonlinear.log
