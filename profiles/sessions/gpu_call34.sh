#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O
export TMPDIR=/tmp
cd /tmp
rm -rf $O/prof_b
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_b -o p -- python $R/bench.py --steps 5 --warmup 2 > $O/prof_b.log 2>&1; echo "rc=$?"
grep -c '"metric"' $O/prof_b.log; grep "SIGSEGV" $O/prof_b.log | head -2
grep -B2 -A26 "SIGSEGV" $O/prof_b.log | grep "rwkv\|Py" | head
