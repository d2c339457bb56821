#!/bin/bash
# round 3, session 11: GEMM epilogue (K-partials four LDS reads at a time) and Wo with two K blocks, whole-step A/B
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O
cd $R
timeout 900 python scripts/ab_bench.py "old:::ai00_server_amd/librwkv_hip_old.so" "epi4::" "old2:::ai00_server_amd/librwkv_hip_old.so" "epi4b::" "epi4+ksb2:RWKV_KSB=2:" "epi4+ksb10:RWKV_KSB=10:" > $O/r3_ab11.log 2>&1; cat $O/r3_ab11.log
