#!/bin/bash
# round 4, session 21: time of ONE block of the tile kernels at 256 rows with L2-resident weights against cold ones (a launch of 32 / 128 blocks: one per CU)
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
mkdir -p gpurun_out
timeout 600 python scripts/tile_block_latency.py 2>&1 | tee gpurun_out/r4c21_block_latency.log
