#!/bin/bash
# round 4, session 26: shape 12 (256 x 128, eight waves) forced on whole prefill steps of the fp16 models against the planner's own choice
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
O=$R/gpurun_out
mkdir -p $O
: > $O/r4c26_shape12_prefill.log
for f in default 12; do
  if [ $f = default ]; then unset RWKV_TILE_SHAPE; else export RWKV_TILE_SHAPE=$f; fi
  echo "== RWKV_TILE_SHAPE=$f" >> $O/r4c26_shape12_prefill.log
  timeout 300 python scripts/prefill_probe.py v6-7b 0 8 1024 2048 2>&1 | tail -1 >> $O/r4c26_shape12_prefill.log
  timeout 300 python scripts/prefill_probe.py v6-7b 0 8 1024 1024 2>&1 | tail -1 >> $O/r4c26_shape12_prefill.log
  timeout 300 python scripts/prefill_probe.py v6-3b 0 32 256 2048 2>&1 | tail -1 >> $O/r4c26_shape12_prefill.log
  timeout 300 python scripts/prefill_probe.py v6-3b 1 32 256 2048 2>&1 | tail -1 >> $O/r4c26_shape12_prefill.log
done
cat $O/r4c26_shape12_prefill.log
