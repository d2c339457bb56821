#!/bin/bash
# round 4, session 23: pipelined-tile stage with hand-issued fragment reads (inline-asm ds_read_b128 + counted lgkmcnt, -DRWKV_EXP_T3_PIPE, librwkv_hip_t3pipe.so)
# against the product build: parity of the tile shapes, isolated launches by rows, prefill A/B
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
O=$R/gpurun_out
mkdir -p $O
V=$R/ai00_server_amd/librwkv_hip_t3pipe.so
RWKV_HIP_LIB=$V timeout 900 python -m pytest tests/test_gpu_bench_paths.py -x -q -k "tile_shape or bit_identical" 2>&1 | tail -4
: > $O/r4c23_t3pipe.log
for lib in product t3pipe; do
  if [ $lib = product ]; then unset RWKV_HIP_LIB; else export RWKV_HIP_LIB=$V; fi
  echo "== $lib" >> $O/r4c23_t3pipe.log
  SHAPES=10,11 TS=256,512,1024,2048 timeout 300 python scripts/tile_by_rows.py 2>&1 | grep -v "^#" >> $O/r4c23_t3pipe.log
  for chunk in 256 2048; do
    timeout 300 python scripts/prefill_probe.py v6-3b 1 32 256 $chunk 2>&1 | tail -1 >> $O/r4c23_t3pipe.log
    timeout 300 python scripts/prefill_probe.py v7-2.9b 2 32 256 $chunk 2>&1 | tail -1 >> $O/r4c23_t3pipe.log
  done
  timeout 300 python scripts/prefill_probe.py v6-7b 0 8 1024 2048 2>&1 | tail -1 >> $O/r4c23_t3pipe.log
  timeout 300 python scripts/prefill_probe.py v6-3b 0 32 256 2048 2>&1 | tail -1 >> $O/r4c23_t3pipe.log
done
cat $O/r4c23_t3pipe.log
