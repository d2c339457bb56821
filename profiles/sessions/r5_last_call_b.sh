set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; P=$O/profiles_new; TAG=r5
mkdir -p $P; export TMPDIR=/tmp; cd /tmp
run_stats () {
  local name=$1; shift
  rm -rf $O/prof_$name; rm -f $P/${TAG}_launch_log_$name.jsonl
  RWKV_LAUNCH_LOG=$P/${TAG}_launch_log_$name.jsonl timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$name -o p -- "$@" > $O/prof_$name.log 2>&1
  [ -f $P/${TAG}_launch_log_$name.jsonl ] && sort -u $P/${TAG}_launch_log_$name.jsonl -o $P/${TAG}_launch_log_$name.jsonl
  local tr=$(find $O/prof_$name -name "*kernel_trace.csv" | head -1)
  [ -n "$tr" ] && python $R/scripts/summarize_trace.py $tr $P/${TAG}_kernel_stats_$name.csv --skip-load
  grep -h "prefill tok/s" $O/prof_$name.log > $P/${TAG}_probe_line_$name.txt
  echo "stats $name"; grep wkv_chunk $P/${TAG}_kernel_stats_$name.csv
}
DEBUG_CLR_GRAPH_PACKET_CAPTURE=0 run_stats prefill_v6-3b_int8_chunk256 python $R/scripts/prefill_probe.py v6-3b 1 32 256 256
DEBUG_CLR_GRAPH_PACKET_CAPTURE=0 run_stats prefill_v7-2.9b_nf4_chunk256 python $R/scripts/prefill_probe.py v7-2.9b 2 32 256 256
