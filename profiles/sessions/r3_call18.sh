#!/bin/bash
# round 3, session 18: in-kernel timeline of wkv_chunk_kernel (trace build) at 8 x 64 and 32 x 64 rows per step
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O
cd $R
( NSEQ=8 LEN=512 CHUNK=512 timeout 300 python scripts/trace_prefill.py; NSEQ=32 LEN=256 CHUNK=2048 timeout 300 python scripts/trace_prefill.py; WORKLOAD=v7-2.9b NSEQ=32 LEN=256 CHUNK=2048 timeout 300 python scripts/trace_prefill.py ) > $O/r3_trace_wkv_chunk.log 2>&1
cat $O/r3_trace_wkv_chunk.log
