#!/bin/bash
# round 4, session 15: one r/k/v/g tile-GEMM launch by row count and tile shape (is a 256-row launch bound by waves per SIMD?)
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
mkdir -p gpurun_out
timeout 600 python scripts/tile_by_rows.py 2>&1 | tee gpurun_out/r4c15_tile_by_rows.log
