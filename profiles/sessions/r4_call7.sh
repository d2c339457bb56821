#!/bin/bash
# round 4, session 7: geometry of the linear (K-split) decode launches: Fv (2560 x 8960) by K split x strips per block
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
O=$R/gpurun_out
mkdir -p $O
: > $O/r4c7_fv_geometry.log
for ksb in 5 7; do for spb in 1 2 3 4 5 6; do
  RWKV_KSB=$ksb RWKV_SPB=$spb SHAPES=fv TS=32,16,1 FMTS=1,0 timeout 120 python scripts/gemm_micro.py ksb$ksb-spb$spb 2>&1 | tail -2 >> $O/r4c7_fv_geometry.log
done; done
SHAPES=fv TS=32,16,1 FMTS=1,0 timeout 120 python scripts/gemm_micro.py planner 2>&1 | tail -2 >> $O/r4c7_fv_geometry.log
cat $O/r4c7_fv_geometry.log
