#!/bin/bash
# round 3, session 10: four rows per ln_shift block on prefill-shaped steps — parity and A/B
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O
cd $R
timeout 1500 python -m pytest tests/test_gpu_embeddings.py tests/test_gpu_knobs.py tests/test_gpu_bench_paths.py tests/test_gpu_parity.py -m gpu -q -k "state_only or LN_ROWS or tile_shape or config5 or mixes or prefill or chunk or full_option or scheduler or perplexity" > $O/r3_t10.log 2>&1; echo "tests rc=$?"; tail -5 $O/r3_t10.log
for lr in 1 4; do for C in 2048 512 256; do echo -n "LN_ROWS=$lr "; RWKV_LN_ROWS=$lr timeout 200 python scripts/prefill_probe.py v6-3b 1 32 256 $C 2>&1 | tail -1; done; done > $O/r3_lnrows_ab.log 2>&1
for lr in 1 4; do echo -n "LN_ROWS=$lr "; RWKV_LN_ROWS=$lr timeout 300 python scripts/prefill_probe.py v6-7b 0 8 2048 2048 2>&1 | tail -1; echo -n "LN_ROWS=$lr "; RWKV_LN_ROWS=$lr timeout 300 python scripts/prefill_probe.py v7-2.9b 2 32 256 256 2>&1 | tail -1; done >> $O/r3_lnrows_ab.log 2>&1
cat $O/r3_lnrows_ab.log
