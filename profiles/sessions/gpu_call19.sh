#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O
cd $R
timeout 1500 python -m pytest tests -m gpu -x -q > $O/t19.log 2>&1; echo "tests rc=$?"
tail -5 $O/t19.log
timeout 600 python bench.py --steps 20 --warmup 5 > $O/bench19.json 2> $O/bench19.err; echo "bench rc=$?"
tail -c 3000 $O/bench19.json
