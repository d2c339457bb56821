#!/bin/bash
# round 3, session 29: threads per row of ln_shift at 512 / 1024 rows; V7-2.9B NF4 prefill kernel statistics at chunk 2048
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O
cd $R
: > $O/r3_ln_threads_512_1024.log
for rep in 1 2; do
  for thr in 0 1024; do
    for cfg in "v6-3b 1 32 256 512" "v6-3b 1 32 256 1024"; do
      RWKV_LN_THREADS=$thr timeout 300 python scripts/prefill_probe.py $cfg 2>&1 | tail -1 | sed "s/^/LN_THREADS=$thr /" >> $O/r3_ln_threads_512_1024.log
    done
  done
done
cat $O/r3_ln_threads_512_1024.log
export TMPDIR=/tmp
cd /tmp
rm -rf $O/prof_v7_2048
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_v7_2048 -o p -- python $R/scripts/prefill_probe.py v7-2.9b 2 32 256 2048 > $O/prof_v7_2048.log 2>&1
tr=$(find $O/prof_v7_2048 -name "*kernel_trace.csv" | head -1)
python $R/scripts/summarize_trace.py $tr $O/r3_kernel_stats_prefill_v7-2.9b_nf4_32x256.csv --skip-load > /dev/null
cut -c1-140 $O/r3_kernel_stats_prefill_v7-2.9b_nf4_32x256.csv | head -14
