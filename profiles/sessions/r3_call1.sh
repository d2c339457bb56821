#!/bin/bash
# round 3, session 1: ingest microbenchmark (scripts/stream_bench.hip) + baseline decode numbers of the round-2 build
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O
cd $R
timeout 300 scripts/stream_bench.bin > $O/r3_stream_bench.log 2>&1; echo "stream rc=$?"
cat $O/r3_stream_bench.log
timeout 400 python scripts/ab_bench.py "base::" > $O/r3_ab_base.log 2>&1; cat $O/r3_ab_base.log
