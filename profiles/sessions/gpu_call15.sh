#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O
cd $R
SHAPES=4,7,10 TS=512,1024 timeout 300 python scripts/tile_bench.py > $O/tile15.log 2>&1; echo "tile rc=$?"
cat $O/tile15.log
timeout 600 python -m pytest tests/test_gpu_bench_paths.py -q -x -k "tile_shape and (10-0 or 10-1 or 10-2 or 4-1)" > $O/t15.log 2>&1; echo "tests rc=$?"
tail -15 $O/t15.log
