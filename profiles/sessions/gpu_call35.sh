#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O
export TMPDIR=/tmp
cd /tmp
rm -rf $O/prof_x
PROBE_MAPS=1 timeout 300 rocprofv3 --kernel-trace --output-format csv -d $O/prof_x -o p -- python $R/scripts/prefill_probe.py v6-3b 1 32 256 256 > $O/prof_x.log 2>&1; echo "rc=$?"
grep "^MAP" $O/prof_x.log | awk '{print $1, $2, $7}' 
grep -A30 SIGSEGV $O/prof_x.log | cut -c1-120 | head -34
