#!/bin/bash
# round 3, session 21: wide v6_mix at <= 128 VGPRs (two blocks per CU): parity, prefill A/B
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O
cd $R
timeout 1500 python -m pytest tests/test_gpu_knobs.py tests/test_gpu_bench_paths.py -m gpu -q -x > $O/r3_t21.log 2>&1; echo "tests rc=$?"; tail -5 $O/r3_t21.log
: > $O/r3_v6wide_2blk.log
for rep in 1 2; do
  for cfg in "v6-3b 1 32 256 2048" "v6-3b 1 32 256 1024" "v6-3b 1 32 256 512" "v6-3b 1 32 256 256" "v6-7b 0 8 2048 1024"; do
    timeout 300 python scripts/prefill_probe.py $cfg 2>&1 | tail -1 >> $O/r3_v6wide_2blk.log
  done
done
cat $O/r3_v6wide_2blk.log
