#!/bin/bash
# round 3, session 3: parity of the new tests + v6_mix changes, A/B of decode steps, GEMM microbench by shape
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O
cd $R
timeout 1500 python -m pytest tests/test_gpu_embeddings.py tests/test_converter.py tests/test_gpu_bench_paths.py -m gpu -x -q > $O/r3_t3.log 2>&1; echo "tests rc=$?"; tail -5 $O/r3_t3.log
timeout 600 python scripts/ab_bench.py "new::" "mix_nt2:RWKV_V6MIX_NT2=1:" "ksw8:RWKV_KSW8=1:" > $O/r3_ab3.log 2>&1; cat $O/r3_ab3.log
FMTS=1 TS=1,32 SPBS=0,2,3,4 timeout 300 python scripts/gemm_bench.py 2>&1 | grep -v "^w1\|^head" > $O/r3_gemm3.log; cat $O/r3_gemm3.log
