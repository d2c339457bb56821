#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O
cd $R
timeout 600 python scripts/ab_bench.py "ln256:RWKV_LN_256=1:" "ln1024::" > $O/ab12.log 2>&1; echo "ab rc=$?"
AB_WORKLOAD=v7-2.9b AB_QUANT=nf4 timeout 600 python scripts/ab_bench.py "v7_ln256:RWKV_LN_256=1:" "v7_ln1024::" >> $O/ab12.log 2>&1; echo "ab2 rc=$?"
timeout 900 python -m pytest tests/test_gpu_bench_paths.py tests/test_gpu_parity.py -q -k "32_slots or small_batches or 256 or greedy or prefill_logits or interleave" > $O/t12.log 2>&1; echo "tests rc=$?"
