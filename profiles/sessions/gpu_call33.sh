#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O
export TMPDIR=/tmp
cd /tmp
run () { rm -rf $O/prof_x; echo "--- $*"; env "$@" timeout 300 rocprofv3 --kernel-trace --output-format csv -d $O/prof_x -o p -- python $R/scripts/prefill_probe.py v6-3b 1 32 256 256 > $O/prof_x.log 2>&1; echo "rc=$?"; grep -c "prefill tok" $O/prof_x.log; }
run RWKV_HIP_LIB=$R/ai00_server_amd/librwkv_hip_trace.so
run RWKV_NO_V6_WIDE=1
run RWKV_NO_DENSE=1
run RWKV_NO_TILE=1
run X=1
grep -A12 SIGSEGV $O/prof_x.log | cut -c1-160 | head -30
