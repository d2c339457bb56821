#!/bin/bash
# round 3, session 2: Infinity-Cache prefetch experiments (scripts/mall_bench.hip)
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O
cd $R
timeout 300 scripts/mall_bench.bin > $O/r3_mall_bench.log 2>&1; echo "mall rc=$?"
cat $O/r3_mall_bench.log
