#!/bin/bash
# after the tile kernels' address change (scalar base + lane offset): parity of the pipelined shapes, then the reduced bench line
cd $GRAFT_REPO_ROOT
timeout 200 python -m pytest -x -q tests/test_gpu_bench_paths.py -k "bit_identical or (tile_shape_at_3b_width and (10- or 11-))" 2>&1 | tail -3
timeout 120 python -m pytest -x -q tests/test_gpu_knobs.py -k "TILE" 2>&1 | tail -2
timeout 150 python bench.py --steps 20 --warmup 5 --no-configs --no-cpu-baseline > gpurun_out/r5_bench_reduced_after_addr.json 2> gpurun_out/r5_bench_reduced_after_addr.err; tail -c 600 gpurun_out/r5_bench_reduced_after_addr.json
timeout 60 python scripts/prefill_probe.py v6-7b 0 8 1024 1024
timeout 60 python scripts/prefill_probe.py v6-3b 1 32 256 2048
