#!/bin/bash
# round 3, session 7: K split on both prefill kernels (parity + A/B), Infinity-Cache prefetch by spare row-kernel workgroups (A/B)
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O
cd $R
timeout 600 python scripts/ab_bench.py "prefetch1::" "prefetch0:RWKV_PREFETCH=0:" > $O/r3_ab7.log 2>&1; cat $O/r3_ab7.log
AB_WORKLOAD=v7-2.9b AB_QUANT=nf4 timeout 600 python scripts/ab_bench.py "v7-pf1::" "v7-pf0:RWKV_PREFETCH=0:" >> $O/r3_ab7.log 2>&1; tail -2 $O/r3_ab7.log
timeout 1500 python -m pytest tests/test_gpu_embeddings.py tests/test_gpu_knobs.py tests/test_gpu_bench_paths.py -m gpu -q -k "state_only or KSPLIT or PREFETCH or tile_shape or config5 or mixes" > $O/r3_t7.log 2>&1; echo "tests rc=$?"; tail -5 $O/r3_t7.log
for ks in 0 1; do for C in 2048 1024 512 256; do echo -n "KSPLIT=$ks "; RWKV_TILE_KSPLIT=$ks timeout 200 python scripts/prefill_probe.py v6-3b 1 32 256 $C 2>&1 | tail -1; done; done > $O/r3_ksplit_ab2.log 2>&1
for ks in 0 1; do for C in 256; do echo -n "KSPLIT=$ks "; RWKV_TILE_KSPLIT=$ks timeout 300 python scripts/prefill_probe.py v7-2.9b 2 32 256 $C 2>&1 | tail -1; done; done >> $O/r3_ksplit_ab2.log 2>&1
cat $O/r3_ksplit_ab2.log
