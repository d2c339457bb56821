#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O
export TMPDIR=/tmp
cd /tmp
for C in 256 128; do
rm -rf $O/prof_c$C
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_c$C -o p -- python $R/scripts/prefill_probe.py v6-3b 1 32 256 $C > $O/prof_c$C.log 2>&1
tr=$(find $O/prof_c$C -name "*kernel_trace.csv" | head -1)
[ -n "$tr" ] && python $R/scripts/summarize_trace.py $tr $O/r2_kernel_stats_prefill_v6-3b_int8_chunk$C.csv --skip-load
grep "prefill tok" $O/prof_c$C.log | tail -1
head -14 $O/r2_kernel_stats_prefill_v6-3b_int8_chunk$C.csv
done
