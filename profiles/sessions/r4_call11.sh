#!/bin/bash
# round 4, session 11: the pipelined kernel on 128x64 tiles by the row count up to which it is used (RWKV_TILE3_64_MAX_T), chunk 1024 / 2048, three models;
# per-kernel times of a 256-row step with the planner's shapes and with shape 11
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O
cd $R
: > $O/r4c11_tile3_64_by_t.log
for mt in 0 1024 2048; do
  export RWKV_TILE3_64_MAX_T=$mt
  echo "== RWKV_TILE3_64_MAX_T=$mt" >> $O/r4c11_tile3_64_by_t.log
  for chunk in 1024 2048; do
    timeout 300 python scripts/prefill_probe.py v6-3b 1 32 256 $chunk 2>&1 | tail -1 >> $O/r4c11_tile3_64_by_t.log
    timeout 300 python scripts/prefill_probe.py v7-2.9b 2 32 256 $chunk 2>&1 | tail -1 >> $O/r4c11_tile3_64_by_t.log
    timeout 300 python scripts/prefill_probe.py v6-7b 0 8 2048 $chunk 2>&1 | tail -1 >> $O/r4c11_tile3_64_by_t.log
  done
done
unset RWKV_TILE3_64_MAX_T
cat $O/r4c11_tile3_64_by_t.log
export TMPDIR=/tmp
cd /tmp
for spec in "default:" "shape11:RWKV_TILE_SHAPE=11"; do
  IFS=: read label envs <<< "$spec"
  ( [ -n "$envs" ] && export $envs
    rm -rf $O/prof_c256_$label
    DEBUG_CLR_GRAPH_PACKET_CAPTURE=0 timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_c256_$label -o p -- python $R/scripts/prefill_probe.py v6-3b 1 32 256 256 > $O/prof_c256_$label.log 2>&1
    tr=$(find $O/prof_c256_$label -name "*kernel_trace.csv" | head -1)
    python $R/scripts/summarize_trace.py $tr $O/r4c11_kernel_stats_c256_$label.csv --skip-load
    echo "== $label"; head -12 $O/r4c11_kernel_stats_c256_$label.csv | cut -c1-200 )
done
