#!/bin/bash
# GPU call: new parity tests, default bench line, decode-only kernel stats for B=32 and B=1
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_bench_paths.py -x -q > $O/t_bench_paths.log 2>&1; echo "tests rc=$?" >> $O/t_bench_paths.log
timeout 600 python bench.py --steps 20 --warmup 5 > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc=$?"
export TMPDIR=/tmp
cd /tmp
for B in 32 1; do
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_b$B -o b$B -- python $R/bench.py --decode-only --no-cpu-baseline --sweep "" --verify-steps 0 --steps 50 --warmup 5 --batch $B > $O/prof_b$B.log 2>&1
  echo "prof b$B rc=$?"
done
find $O -name "*kernel_stats.csv" | head
