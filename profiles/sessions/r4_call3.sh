#!/bin/bash
# round 4, session 3: cold-start price list of a decode-shaped kernel (code size, kernarg walks, block shape); full-depth parity in both precisions
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
O=$R/gpurun_out
mkdir -p $O
timeout 300 ./scripts/coldstart_bench.bin 2>&1 | tee $O/r4c3_coldstart.log
: > $O/full_depth_errors.jsonl
timeout 1500 python -m pytest tests/test_gpu_full_depth.py -q -s 2>&1 | grep -v "^$" | grep "full-depth\|passed\|failed\|Error\|assert" | tail -60 > $O/r4c3_full_depth.log
cat $O/r4c3_full_depth.log
