#!/bin/bash
# round 3, session 22: wide v6_mix as two launches on >= 1024-row steps: parity (three forms bit-equal), prefill A/B, per-kernel times
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_knobs.py -m gpu -q -x -k "wide_mix" > $O/r3_t22.log 2>&1; echo "tests rc=$?"; tail -5 $O/r3_t22.log
: > $O/r3_v6_split.log
for rep in 1 2; do
  for ns in 0 1; do
  for cfg in "v6-3b 1 32 256 2048" "v6-3b 1 32 256 1024" "v6-7b 0 8 2048 1024" "v6-7b 0 8 2048 2048"; do
    RWKV_NO_V6_SPLIT=$ns timeout 300 python scripts/prefill_probe.py $cfg 2>&1 | tail -1 | sed "s/^/NO_V6_SPLIT=$ns /" >> $O/r3_v6_split.log
  done
  done
done
cat $O/r3_v6_split.log
export TMPDIR=/tmp
cd /tmp
rm -rf $O/prof_v6split
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_v6split -o p -- python $R/scripts/prefill_probe.py v6-3b 1 32 256 2048 > $O/prof_v6split.log 2>&1
tr=$(find $O/prof_v6split -name "*kernel_trace.csv" | head -1)
python $R/scripts/summarize_trace.py $tr $O/r3_kernel_stats_prefill_v6split_2048.csv --skip-load > /dev/null
cut -c1-150 $O/r3_kernel_stats_prefill_v6split_2048.csv | head -14
