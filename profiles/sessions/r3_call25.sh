#!/bin/bash
# round 3, session 25: state after the prefill work — smoke, whole GPU suite, the default bench line, prefill kernel statistics
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
P=$O/profiles_new
mkdir -p $P
cd $R
python -c "import __graft_entry__ as g; g.smoke()" > $O/r3_smoke.log 2>&1; echo "smoke rc=$?"; tail -1 $O/r3_smoke.log
timeout 2700 python -m pytest tests -m gpu -q > $O/r3_t25.log 2>&1; echo "tests rc=$?"; tail -6 $O/r3_t25.log
( time timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 ) > $O/r3_bench_default.json 2> $O/r3_bench_default.err; echo "bench rc=$?"; tail -3 $O/r3_bench_default.err; head -c 600 $O/r3_bench_default.json
export TMPDIR=/tmp
cd /tmp
run_stats () {
  local name=$1; shift
  rm -rf $O/prof_$name
  timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$name -o p -- "$@" > $O/prof_$name.log 2>&1
  local tr=$(find $O/prof_$name -name "*kernel_trace.csv" | head -1)
  [ -n "$tr" ] && python $R/scripts/summarize_trace.py $tr $P/r3_kernel_stats_$name.csv --skip-load > /dev/null
  tail -1 $O/prof_$name.log
}
run_stats prefill_v6-3b_int8_32x256 python $R/scripts/prefill_probe.py v6-3b 1 32 256 2048
DEBUG_CLR_GRAPH_PACKET_CAPTURE=0 run_stats prefill_v6-3b_int8_chunk256 python $R/scripts/prefill_probe.py v6-3b 1 32 256 256
DEBUG_CLR_GRAPH_PACKET_CAPTURE=0 run_stats prefill_v7-2.9b_nf4_chunk256 python $R/scripts/prefill_probe.py v7-2.9b 2 32 256 256
ls $P
