#!/bin/bash
# round 3, session 4: what limits the T = 1 GEMM? waves x loads-in-flight matrix, in-kernel timeline with 5 and 10 waves
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O
cd $R
timeout 200 scripts/stream_bench.bin w > $O/r3_stream_wxr.log 2>&1; cat $O/r3_stream_wxr.log
for k in 0 1; do echo "RWKV_KSW8=$k"; RWKV_KSW8=$k TS=1,16 timeout 200 python scripts/trace_gemm.py run; done > $O/r3_trace_gemm_t1.log 2>&1; cat $O/r3_trace_gemm_t1.log
for k in 0 1; do echo "RWKV_KSW8=$k"; RWKV_KSW8=$k FMTS=1,0 TS=1,8,16 SPBS=0 timeout 200 python scripts/gemm_bench.py 2>&1 | grep -v "^w1\|^head"; done > $O/r3_gemm4.log; cat $O/r3_gemm4.log
nproc; cat /sys/fs/cgroup/cpu.max 2>/dev/null; lscpu | grep -E "Model name|Socket|Core|Thread|NUMA node\(s\)"; python -c "import os; print(len(os.sched_getaffinity(0)))"
