#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O
cd $R
timeout 300 scripts/chain_bench.bin > $O/chain_bench.log 2>&1; echo "chain rc=$?"
timeout 600 python scripts/debug_7b.py > $O/debug_7b.log 2>&1; echo "dbg rc=$?"
timeout 900 python -m pytest tests/test_gpu_bench_paths.py -q -k "not tile_shape" > $O/t_bench_paths2.log 2>&1; echo "tests rc=$?"
timeout 300 python bench.py --steps 20 --warmup 5 --decode-only --no-cpu-baseline > $O/bench2.json 2> $O/bench2.err; echo "bench rc=$?"
