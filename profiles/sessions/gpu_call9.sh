#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O
cd $R
timeout 900 python scripts/ab_bench.py "base::" "ksw8:RWKV_KSW8=1:" "spb2:RWKV_SPB=2:" "spb4:RWKV_SPB=4:" > $O/ab9.log 2>&1; echo "ab rc=$?"
