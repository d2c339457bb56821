#!/bin/bash
# round 4, session 10: the pipelined tile kernel on 128 x 64 tiles (shape 11): parity (oracle, bit-identity with the 64x64 shape), prefill A/B at chunk 256 / 512
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
O=$R/gpurun_out
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_bench_paths.py -x -q -k "tile_shape or bit_identical" 2>&1 | tail -4
: > $O/r4c10_tile3_64.log
for spec in "default:" "shape11:RWKV_TILE_SHAPE=11" "t3_64<=256:RWKV_TILE3_64_MAX_T=256" "t3_64<=512:RWKV_TILE3_64_MAX_T=512" "t3_64<=1024:RWKV_TILE3_64_MAX_T=1024"; do
  IFS=: read label envs <<< "$spec"
  ( [ -n "$envs" ] && export $envs
    echo "== $label" >> $O/r4c10_tile3_64.log
    for chunk in 256 512 1024; do
      timeout 300 python scripts/prefill_probe.py v6-3b 1 32 256 $chunk 2>&1 | tail -1 >> $O/r4c10_tile3_64.log
    done
    timeout 300 python scripts/prefill_probe.py v7-2.9b 2 32 256 256 2>&1 | tail -1 >> $O/r4c10_tile3_64.log
    timeout 300 python scripts/prefill_probe.py v6-3b 0 32 256 256 2>&1 | tail -1 >> $O/r4c10_tile3_64.log )
done
cat $O/r4c10_tile3_64.log
