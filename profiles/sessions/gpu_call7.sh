#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_bench_paths.py -q -x -k "second_form or config5" > $O/t7.log 2>&1; echo "tests rc=$?"
TS=512,1024 SHAPES=4,7,100,101,102,103 timeout 400 python scripts/tile_bench.py > $O/tile7.log 2>&1; echo "tile rc=$?"
for V in 0 1 2; do
  echo "== RWKV_TILE2=$V" >> $O/tile7.log
  RWKV_TILE2=$V timeout 200 python scripts/prefill_probe.py v6-3b 1 32 256 512 >> $O/tile7.log 2>&1
done
RWKV_TILE2=0 timeout 300 python scripts/prefill_probe.py v6-7b 0 8 1024 1024 >> $O/tile7.log 2>&1
timeout 300 python scripts/prefill_probe.py v6-7b 0 8 1024 1024 >> $O/tile7.log 2>&1
