#!/bin/bash
# round 3, session 28: touch-ahead of the next launch's weights by the 64x64 tile launches (steps <= 640 rows): parity, prefill A/B
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O
cd $R
timeout 600 python -m pytest tests/test_gpu_bench_paths.py -m gpu -q -x -k "tile_shape or bit_identical" > $O/r3_t28.log 2>&1; echo "tests rc=$?"; tail -4 $O/r3_t28.log
: > $O/r3_touch_ahead.log
for rep in 1 2; do
  for m in 640 0; do
    for cfg in "v6-3b 1 32 256 256" "v6-3b 1 32 256 512" "v7-2.9b 2 32 256 256" "v6-7b 0 8 512 256"; do
      RWKV_TOUCH_AHEAD_MAX_T=$m timeout 300 python scripts/prefill_probe.py $cfg 2>&1 | tail -1 | sed "s/^/TOUCH_AHEAD_MAX_T=$m /" >> $O/r3_touch_ahead.log
    done
  done
done
cat $O/r3_touch_ahead.log
