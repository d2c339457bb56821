#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O
cd $R
timeout 600 python scripts/ab_bench.py "nodense:RWKV_NO_DENSE=1:" "dense::" "dense_ksb1:RWKV_KSB=1:" "dense_ksb2:RWKV_KSB=2:" > $O/ab4.log 2>&1; echo "ab rc=$?"
TS=32,1 timeout 300 python scripts/trace_gemm.py run > $O/trace_gemm_l1.log 2>&1; echo "tg rc=$?"
timeout 900 python -m pytest tests/test_gpu_bench_paths.py tests/test_gpu_parity.py -q -x -k "32_slots or small_batches or 256 or sampling or pageable or mirostat or typical" > $O/t4.log 2>&1; echo "tests rc=$?"
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench4.json 2> $O/bench4.err; echo "bench rc=$?"
