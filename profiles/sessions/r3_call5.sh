#!/bin/bash
# round 3, session 5: whole GPU suite after the Knobs / KSW8 / LoRA / ABI changes, then the default bench line
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O
cd $R
timeout 2400 python -m pytest tests -m gpu -q > $O/r3_t5.log 2>&1; echo "tests rc=$?"; tail -25 $O/r3_t5.log
( time timeout 900 python bench.py --steps 20 --warmup 5 ) > $O/r3_bench5.json 2> $O/r3_bench5.err; echo "bench rc=$?"; tail -3 $O/r3_bench5.err; head -c 6000 $O/r3_bench5.json
