#!/bin/bash
# round 4, session 9: 256-row steps on the multi-pass K-stationary kernel (RWKV_NO_TILE=1), non-temporal vs default-policy weight loads, against the tile path
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
O=$R/gpurun_out
mkdir -p $O
: > $O/r4c9_notile256.log
for spec in "tile::" "notile-nt:RWKV_NO_TILE=1:" "notile-dflt:RWKV_NO_TILE=1:exp_WDEFAULT"; do
  IFS=: read label envs lib <<< "$spec"
  ( [ -n "$envs" ] && export $envs; [ -n "$lib" ] && export RWKV_HIP_LIB=$R/ai00_server_amd/librwkv_hip_$lib.so
    echo "== $label" >> $O/r4c9_notile256.log
    timeout 300 python scripts/prefill_probe.py v6-3b 1 32 256 256 2>&1 | tail -1 >> $O/r4c9_notile256.log
    timeout 300 python scripts/prefill_probe.py v7-2.9b 2 32 256 256 2>&1 | tail -1 >> $O/r4c9_notile256.log )
done
cat $O/r4c9_notile256.log
