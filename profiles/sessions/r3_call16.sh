#!/bin/bash
# round 3, session 16: router_loop device list (GPU test), threads per row of ln_shift on prefill-shaped steps (A/B)
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -k "router or adapter or lora" > $O/r3_t16.log 2>&1; echo "tests rc=$?"; tail -4 $O/r3_t16.log
: > $O/r3_ln_threads_ab.log
for rep in 1 2; do
for thr in 0 512 1024; do
  for cfg in "v6-3b 1 32 256 2048" "v6-3b 1 32 256 256" "v7-2.9b 2 32 256 256"; do
    RWKV_LN_THREADS=$thr timeout 300 python scripts/prefill_probe.py $cfg 2>&1 | tail -1 | sed "s/^/LN_THREADS=$thr /" >> $O/r3_ln_threads_ab.log
  done
done
done
cat $O/r3_ln_threads_ab.log
