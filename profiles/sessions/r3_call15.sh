#!/bin/bash
# round 3, session 15: final state — whole GPU suite, smoke, the default bench line (driver's command), launcher test on one device
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O
cd $R
python -c "import __graft_entry__ as g; g.smoke()" > $O/r3_smoke.log 2>&1; echo "smoke rc=$?"; tail -1 $O/r3_smoke.log
timeout 2700 python -m pytest tests -m gpu -q > $O/r3_t15.log 2>&1; echo "tests rc=$?"; tail -6 $O/r3_t15.log
( time timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 ) > $O/r3_bench_default.json 2> $O/r3_bench_default.err; echo "bench rc=$?"; tail -3 $O/r3_bench_default.err; head -c 1500 $O/r3_bench_default.json
