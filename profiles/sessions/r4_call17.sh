#!/bin/bash
# round 4, session 17: after the tile epilogue rewrite: parity of every tile shape, the 128x64 rule on / off by chunk, four engines
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
O=$R/gpurun_out
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_bench_paths.py -x -q 2>&1 | tail -3
: > $O/r4c17_rule.log
for v in 1 0; do
  export RWKV_TILE3_64=$v
  echo "== RWKV_TILE3_64=$v" >> $O/r4c17_rule.log
  for chunk in 256 512 1024 2048; do
    timeout 300 python scripts/prefill_probe.py v6-3b 1 32 256 $chunk 2>&1 | tail -1 >> $O/r4c17_rule.log
    timeout 300 python scripts/prefill_probe.py v7-2.9b 2 32 256 $chunk 2>&1 | tail -1 >> $O/r4c17_rule.log
    timeout 300 python scripts/prefill_probe.py v6-3b 0 32 256 $chunk 2>&1 | tail -1 >> $O/r4c17_rule.log
    timeout 300 python scripts/prefill_probe.py v6-7b 0 8 2048 $chunk 2>&1 | tail -1 >> $O/r4c17_rule.log
  done
done
cat $O/r4c17_rule.log
