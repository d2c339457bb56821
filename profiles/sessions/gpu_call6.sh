#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O
cd $R
for X in 0 1; do
  echo "== RWKV_TILE_XCD=$X" >> $O/tile6.log
  RWKV_TILE_XCD=$X TS=512,1024 SHAPES=3,4,6,7 timeout 300 python scripts/tile_bench.py >> $O/tile6.log 2>&1
  RWKV_TILE_XCD=$X timeout 200 python scripts/prefill_probe.py v6-3b 1 32 256 512 >> $O/tile6.log 2>&1
done
timeout 600 python -m pytest tests/test_gpu_parity.py -q -x -k "prefab or tile_gemm or chunk_size" > $O/t6.log 2>&1; echo "tests rc=$?"
timeout 600 python -m pytest tests/test_gpu_bench_paths.py -q -x -k "tile_shape or config5" >> $O/t6.log 2>&1; echo "tests2 rc=$?"
