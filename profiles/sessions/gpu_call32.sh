#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O
export TMPDIR=/tmp
cd /tmp
echo "--- plain chunk 256"; timeout 300 python $R/scripts/prefill_probe.py v6-3b 1 32 256 256 2>&1 | tail -1
for C in 512 256; do
  for F in 65 0; do
    rm -rf $O/prof_x
    echo "--- rocprofv3 chunk $C RWKV_TILE3_FILL=$F"
    RWKV_TILE3_FILL=$F timeout 400 rocprofv3 --kernel-trace --output-format csv -d $O/prof_x -o p -- python $R/scripts/prefill_probe.py v6-3b 1 32 256 $C > $O/prof_x.log 2>&1; echo "rc=$?"
    grep "prefill tok\|SIGSEGV" $O/prof_x.log | tail -2
  done
done
