#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_bench_paths.py tests/test_gpu_parity.py -q -x -k "tile_shape or config5 or prefill or chunk_size or full_width" > $O/t37.log 2>&1; echo "tests rc=$?"; tail -3 $O/t37.log
for C in 256 512 2048; do timeout 200 python scripts/prefill_probe.py v6-3b 1 32 256 $C 2>&1 | tail -1; done
timeout 300 python scripts/prefill_probe.py v6-7b 0 8 1024 1024 2>&1 | tail -1
