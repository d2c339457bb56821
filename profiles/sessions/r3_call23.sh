#!/bin/bash
# round 3, session 23: ln_shift rows contiguous per XCD; threshold of the two-launch wide mix: parity, prefill A/B
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_knobs.py -m gpu -q -x -k "wide_mix or TILE_XCD" > $O/r3_t23.log 2>&1; echo "tests rc=$?"; tail -5 $O/r3_t23.log
: > $O/r3_ln_xcd_rows.log
for rep in 1 2; do
  for x in 1 0; do
  for cfg in "v6-3b 1 32 256 2048" "v6-3b 1 32 256 256" "v7-2.9b 2 32 256 256" "v6-7b 0 8 2048 1024"; do
    RWKV_TILE_XCD=$x timeout 300 python scripts/prefill_probe.py $cfg 2>&1 | tail -1 | sed "s/^/TILE_XCD=$x /" >> $O/r3_ln_xcd_rows.log
  done
  done
  for m in 512 256; do
  for cfg in "v6-3b 1 32 256 512" "v6-3b 1 32 256 256"; do
    RWKV_V6_SPLIT_MIN_T=$m timeout 300 python scripts/prefill_probe.py $cfg 2>&1 | tail -1 | sed "s/^/V6_SPLIT_MIN_T=$m /" >> $O/r3_ln_xcd_rows.log
  done
  done
  timeout 300 python scripts/prefill_probe.py v6-3b 1 32 256 512 2>&1 | tail -1 | sed "s/^/default /" >> $O/r3_ln_xcd_rows.log
done
cat $O/r3_ln_xcd_rows.log
