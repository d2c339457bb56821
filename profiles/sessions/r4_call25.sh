#!/bin/bash
# round 4, session 25: pipelined kernel on 256 rows x 128 tokens, EIGHT waves sharing one X ring (shape 12) — the weight refill is the largest
# single cost of a stage (session 24: -17 % Int8 / -32 % fp16 without it), so halve the X bytes per flop through the CU's vector memory path
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
O=$R/gpurun_out
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_bench_paths.py -x -q -k "tile_shape or bit_identical" 2>&1 | tail -4
SHAPES=10,12 TS=512,1024,2048,4096 timeout 300 python scripts/tile_by_rows.py 2>&1 | grep -v "^#" | tee $O/r4c25_shape12_by_rows.log
