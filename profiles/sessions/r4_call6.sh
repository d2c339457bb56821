#!/bin/bash
# round 4, session 6: decode GEMM with a ring of two rounds in flight per wave instead of every load issued up-front (single shot), T = 32 / 16
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
O=$R/gpurun_out
mkdir -p $O
: > $O/r4c6_ring.log
TS=32,16,1 timeout 300 python scripts/gemm_micro.py shot 2>&1 | tail -8 >> $O/r4c6_ring.log
RWKV_EXP_NO_SHOT=1 TS=32,16,1 timeout 300 python scripts/gemm_micro.py ring 2>&1 | tail -8 >> $O/r4c6_ring.log
RWKV_SPB=2 TS=32,16,1 timeout 300 python scripts/gemm_micro.py spb2 2>&1 | tail -8 >> $O/r4c6_ring.log
RWKV_SPB=1 TS=32,16,1 timeout 300 python scripts/gemm_micro.py spb1 2>&1 | tail -8 >> $O/r4c6_ring.log
RWKV_SPB=4 TS=32,16,1 timeout 300 python scripts/gemm_micro.py spb4 2>&1 | tail -8 >> $O/r4c6_ring.log
RWKV_SPB=6 TS=32,16,1 timeout 300 python scripts/gemm_micro.py spb6 2>&1 | tail -8 >> $O/r4c6_ring.log
cat $O/r4c6_ring.log
