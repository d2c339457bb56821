#!/bin/bash
# round 4, session 28: the tree as it stands at the end of the round — the whole GPU suite, the default bench line, the profile set with launch logs
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
O=$R/gpurun_out
mkdir -p $O
( time timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | tail -6 ) 2>&1 | tail -10
cp $O/full_depth_errors.jsonl $O/r4c28_full_depth_errors.jsonl 2>/dev/null
( time timeout 900 python bench.py --steps 20 --warmup 5 > $O/r4c28_bench_default.json 2> $O/r4c28_bench_default.err ) 2>&1 | tail -3
tail -2 $O/r4c28_bench_default.err
bash scripts/gpu_profiles.sh r4 2>&1 | tail -8
