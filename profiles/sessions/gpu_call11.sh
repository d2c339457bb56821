#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_bench_paths.py -q -k "tile or config5 or prefill or chunk_size or full_option or golden" > $O/t11.log 2>&1; echo "tests rc=$?"
for V in 1 0; do
  echo "== RWKV_NO_V6_WIDE=$V" >> $O/wide11.log
  RWKV_NO_V6_WIDE=$V timeout 200 python scripts/prefill_probe.py v6-3b 1 32 256 512 >> $O/wide11.log 2>&1
  RWKV_NO_V6_WIDE=$V timeout 300 python scripts/prefill_probe.py v6-7b 0 8 1024 1024 >> $O/wide11.log 2>&1
done
