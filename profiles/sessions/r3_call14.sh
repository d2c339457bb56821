#!/bin/bash
# round 3, session 14: 256x128 pipelined tile (shape 11, quantised weights): isolated GEMMs, parity, model-level prefill A/B
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O
cd $R
SHAPES=4,10,11 timeout 300 python scripts/tile_bench2.py > $O/r3_tile3b_bench.log 2>&1; cat $O/r3_tile3b_bench.log
timeout 900 python -m pytest tests/test_gpu_bench_paths.py -m gpu -q -k "tile_shape" > $O/r3_t14.log 2>&1; echo "tests rc=$?"; tail -4 $O/r3_t14.log
for v in 0 1; do for C in 2048 1024 512; do echo -n "TILE3B=$v "; RWKV_TILE3B=$v timeout 200 python scripts/prefill_probe.py v6-3b 1 32 256 $C 2>&1 | tail -1; done; done > $O/r3_tile3b_ab.log 2>&1
for v in 0 1; do echo -n "TILE3B=$v "; RWKV_TILE3B=$v timeout 300 python scripts/prefill_probe.py v7-2.9b 2 32 256 2048 2>&1 | tail -1; done >> $O/r3_tile3b_ab.log 2>&1
cat $O/r3_tile3b_ab.log
