#!/bin/bash
# 64x64 tile kernel with the problem record's hot fields pinned in SGPRs (no scalar reloads in the K loop): parity, then the model (A/B against r5_final_b.sh)
cd $GRAFT_REPO_ROOT
timeout 45 python -m pytest -x -q "tests/test_gpu_bench_paths.py::test_every_prefill_tile_shape_at_3b_width[4-1]" "tests/test_gpu_bench_paths.py::test_every_prefill_tile_shape_at_3b_width[3-0]" "tests/test_gpu_bench_paths.py::test_every_prefill_tile_shape_at_3b_width[7-2]" 2>&1 | tail -2
timeout 25 python scripts/prefill_probe.py v6-3b 1 32 256 256
