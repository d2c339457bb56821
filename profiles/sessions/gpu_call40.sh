#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O
cd $R
SHAPES=3,4,1 TS=256,512,1024 timeout 300 python scripts/tile_bench.py 2>&1 | grep -v "^w1" > $O/tile40.log; cat $O/tile40.log
timeout 900 python -m pytest tests/test_gpu_bench_paths.py tests/test_gpu_parity.py -q -x -k "tile_shape or prefill or pipelined" > $O/t40.log 2>&1; echo "tests rc=$?"; tail -3 $O/t40.log
for C in 256 512 2048; do timeout 200 python scripts/prefill_probe.py v6-3b 1 32 256 $C 2>&1 | tail -1; done
