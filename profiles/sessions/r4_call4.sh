#!/bin/bash
# round 4, session 4: decode GEMM with the reworked epilogue (parameters pinned before the barrier, bias / POST operands prefetched, grouped
# LDS reads), host-computed Kb / nslice, slice rotation — microbench and whole-step A/B against the previous build; parity of the decode paths
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
O=$R/gpurun_out
mkdir -p $O
: > $O/r4c4_micro.log
RWKV_HIP_LIB=$R/ai00_server_amd/librwkv_hip_base.so timeout 300 python scripts/gemm_micro.py base 2>&1 | tail -8 >> $O/r4c4_micro.log
timeout 300 python scripts/gemm_micro.py new 2>&1 | tail -8 >> $O/r4c4_micro.log
RWKV_GEMM_ROTATE=0 timeout 300 python scripts/gemm_micro.py new-norot 2>&1 | tail -8 >> $O/r4c4_micro.log
cat $O/r4c4_micro.log
timeout 600 python scripts/ab_bench.py "base::ai00_server_amd/librwkv_hip_base.so" "new::" "new-norot:RWKV_GEMM_ROTATE=0:" 2>&1 | tee $O/r4c4_ab.log
AB_QUANT=none timeout 400 python scripts/ab_bench.py "base-f16::ai00_server_amd/librwkv_hip_base.so" "new-f16::" 2>&1 | tee -a $O/r4c4_ab.log
AB_WORKLOAD=v7-2.9b AB_QUANT=nf4 timeout 400 python scripts/ab_bench.py "base-v7::ai00_server_amd/librwkv_hip_base.so" "new-v7::" 2>&1 | tee -a $O/r4c4_ab.log
timeout 900 python -m pytest tests/test_gpu_bench_paths.py tests/test_gpu_knobs.py -x -q 2>&1 | tail -5
