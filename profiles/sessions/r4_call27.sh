#!/bin/bash
# round 4, session 27: the 64-token pipelined tile (shape 11) with an eight-stage X ring and four weight register sets (-DRWKV_EXP_T3_DEEP):
# a weight group gets six stages to arrive instead of three.  Product build (generalised code, same constants as before) and variant: parity, then A/B.
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
O=$R/gpurun_out
mkdir -p $O
V=$R/ai00_server_amd/librwkv_hip_t3deep.so
timeout 900 python -m pytest tests/test_gpu_bench_paths.py -x -q -k "tile_shape or bit_identical" 2>&1 | tail -3
RWKV_HIP_LIB=$V timeout 900 python -m pytest tests/test_gpu_bench_paths.py -x -q -k "tile_shape or bit_identical" 2>&1 | tail -3
: > $O/r4c27_t3deep.log
for lib in product t3deep; do
  if [ $lib = product ]; then unset RWKV_HIP_LIB; else export RWKV_HIP_LIB=$V; fi
  echo "== $lib" >> $O/r4c27_t3deep.log
  SHAPES=11 TS=256,384,512,1024 timeout 300 python scripts/tile_by_rows.py 2>&1 | grep -v "^#" >> $O/r4c27_t3deep.log
  timeout 300 python scripts/prefill_probe.py v6-3b 1 32 256 256 2>&1 | tail -1 >> $O/r4c27_t3deep.log
  timeout 300 python scripts/prefill_probe.py v7-2.9b 2 32 256 256 2>&1 | tail -1 >> $O/r4c27_t3deep.log
  timeout 300 python scripts/prefill_probe.py v6-3b 0 32 256 512 2>&1 | tail -1 >> $O/r4c27_t3deep.log
done
cat $O/r4c27_t3deep.log
