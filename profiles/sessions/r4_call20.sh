#!/bin/bash
# round 4, session 20: the k-step-split eight-wave 128x64 tile (shape 12): parity + reproducibility, isolated launch by rows against shape 11, prefill A/B
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
O=$R/gpurun_out
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_bench_paths.py -x -q -k "tile_shape or bit_identical" 2>&1 | tail -4
SHAPES=12,11 TS=256,384,512,1024 timeout 300 python scripts/tile_by_rows.py 2>&1 | tee $O/r4c20_shape12_by_rows.log
: > $O/r4c20_shape12_prefill.log
for v in 1 2; do
  export RWKV_TILE3_64=$v
  echo "== RWKV_TILE3_64=$v" >> $O/r4c20_shape12_prefill.log
  for chunk in 256 512 1024; do
    timeout 300 python scripts/prefill_probe.py v6-3b 1 32 256 $chunk 2>&1 | tail -1 >> $O/r4c20_shape12_prefill.log
    timeout 300 python scripts/prefill_probe.py v7-2.9b 2 32 256 $chunk 2>&1 | tail -1 >> $O/r4c20_shape12_prefill.log
    timeout 300 python scripts/prefill_probe.py v6-7b 0 8 1024 $chunk 2>&1 | tail -1 >> $O/r4c20_shape12_prefill.log
  done
done
cat $O/r4c20_shape12_prefill.log
