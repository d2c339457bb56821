#!/bin/bash
# round 4, session 22: 192 x 64 six-wave blocks (shape 12): parity + bit identity, isolated launch, prefill A/B (RWKV_TILE3_64=2 default: 192-row rule on; =1: 128-row only)
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
O=$R/gpurun_out
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_bench_paths.py -x -q -k "tile_shape or bit_identical" 2>&1 | tail -4
SHAPES=12,11 TS=256,384,512 timeout 300 python scripts/tile_by_rows.py 2>&1 | tee $O/r4c22_shape192_by_rows.log
: > $O/r4c22_shape192_prefill.log
for v in 2 1; do
  export RWKV_TILE3_64=$v
  echo "== RWKV_TILE3_64=$v" >> $O/r4c22_shape192_prefill.log
  for chunk in 256 512; do
    timeout 300 python scripts/prefill_probe.py v6-3b 1 32 256 $chunk 2>&1 | tail -1 >> $O/r4c22_shape192_prefill.log
    timeout 300 python scripts/prefill_probe.py v7-2.9b 2 32 256 $chunk 2>&1 | tail -1 >> $O/r4c22_shape192_prefill.log
    timeout 300 python scripts/prefill_probe.py v6-7b 0 8 1024 $chunk 2>&1 | tail -1 >> $O/r4c22_shape192_prefill.log
    timeout 300 python scripts/prefill_probe.py v6-3b 0 32 256 $chunk 2>&1 | tail -1 >> $O/r4c22_shape192_prefill.log
  done
done
cat $O/r4c22_shape192_prefill.log
