#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O
cd $R
TS=32,1 timeout 300 python scripts/trace_gemm.py run > $O/trace_gemm.log 2>&1; echo "tg rc=$?"
B=32 timeout 300 python scripts/trace_step.py > $O/trace_step_b32.log 2>&1; echo "ts32 rc=$?"
B=1 timeout 300 python scripts/trace_step.py > $O/trace_step_b1.log 2>&1; echo "ts1 rc=$?"
