#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O
export TMPDIR=/tmp
cd /tmp
rm -rf $O/prof_v7pf
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_v7pf -o p -- python $R/scripts/prefill_probe.py v7-2.9b 2 32 256 2048 > $O/prof_v7pf.log 2>&1
tr=$(find $O/prof_v7pf -name "*kernel_trace.csv" | head -1)
[ -n "$tr" ] && python $R/scripts/summarize_trace.py $tr $O/r2_kernel_stats_prefill_v7-2.9b_nf4_32x256.csv --skip-load
tail -2 $O/prof_v7pf.log
head -16 $O/r2_kernel_stats_prefill_v7-2.9b_nf4_32x256.csv
