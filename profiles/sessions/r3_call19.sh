#!/bin/bash
# round 3, session 19: wkv_chunk with the V6 decay LoRA on MFMA and packed fp32 in the recurrence — parity, then prefill A/B
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O
cd $R
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_embeddings.py tests/test_gpu_bench_paths.py -m gpu -q -x > $O/r3_t19.log 2>&1; echo "tests rc=$?"; tail -5 $O/r3_t19.log
: > $O/r3_wkv_chunk_pk.log
for rep in 1 2; do
  for cfg in "v6-3b 1 32 256 2048" "v6-3b 1 32 256 256" "v7-2.9b 2 32 256 256" "v7-2.9b 2 32 256 2048" "v6-7b 0 8 2048 1024"; do
    timeout 300 python scripts/prefill_probe.py $cfg 2>&1 | tail -1 >> $O/r3_wkv_chunk_pk.log
  done
done
cat $O/r3_wkv_chunk_pk.log
