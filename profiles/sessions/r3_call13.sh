#!/bin/bash
# round 3, session 13: wkv_chunk with three blocks per CU (V5, V6 Dd = 64) — parity + prefill A/B against the previous library
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O
cd $R
for lib in ai00_server_amd/librwkv_hip_old.so ai00_server_amd/librwkv_hip.so; do for C in 2048 512 256; do echo -n "$lib "; RWKV_HIP_LIB=$R/$lib timeout 200 python scripts/prefill_probe.py v6-3b 1 32 256 $C 2>&1 | tail -1; done; done > $O/r3_wkv3_ab.log 2>&1
cat $O/r3_wkv3_ab.log
timeout 1500 python -m pytest tests/test_gpu_embeddings.py tests/test_gpu_bench_paths.py tests/test_gpu_parity.py -m gpu -q -k "state_only or tile_shape or mixes or prefill or chunk_size or full_option or golden" > $O/r3_t13.log 2>&1; echo "tests rc=$?"; tail -4 $O/r3_t13.log
