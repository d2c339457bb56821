#!/bin/bash
# round 4, session 13: 256-row GEMM launches with the matrix resident in the Infinity Cache / L2 (one matrix, repeated) against cold matrices: is the small-step
# tile GEMM bound by the latency of cold weights?
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
mkdir -p gpurun_out
timeout 600 python scripts/tile_hot_cold.py 2>&1 | tee gpurun_out/r4c13_hot_cold.log
