#!/bin/bash
# round 4, session 5: two dependent chains over the same matrices on two streams of one graph (what two half-batches of a decode step would
# do) against one chain: us per matrix for 2 x T=16 (dual) vs 1 x T=32 (single), non-temporal and default-policy weight loads
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
O=$R/gpurun_out
mkdir -p $O
: > $O/r4c5_dual.log
for lib in "" exp_WDEFAULT; do
  if [ -z "$lib" ]; then unset RWKV_HIP_LIB; tag=nt; else export RWKV_HIP_LIB=$R/ai00_server_amd/librwkv_hip_$lib.so; tag=dflt; fi
  TS=32,16,8 FMTS=1,0 timeout 300 python scripts/gemm_micro.py single-$tag 2>&1 | tail -8 >> $O/r4c5_dual.log
  RWKV_BENCH_DUAL=1 TS=16,8,4 FMTS=1,0 timeout 300 python scripts/gemm_micro.py dual-$tag 2>&1 | tail -8 >> $O/r4c5_dual.log
done
cat $O/r4c5_dual.log
