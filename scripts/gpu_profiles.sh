#!/bin/bash
# Round profiles (GPU box): decode-only kernel statistics for the BASELINE engines, a prefill run, MFMA-utilisation and HBM-traffic
# counter passes.  Summaries land in gpurun_out/profiles_new/ (copied to profiles/ by hand after review).
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
P=$O/profiles_new
TAG=${1:-r3}
mkdir -p $P
export TMPDIR=/tmp
cd /tmp
run_stats () {   # name, then the command
  local name=$1; shift
  rm -rf $O/prof_$name
  rm -f $P/${TAG}_launch_log_$name.jsonl
  RWKV_LAUNCH_LOG=$P/${TAG}_launch_log_$name.jsonl timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$name -o p -- "$@" > $O/prof_$name.log 2>&1
  [ -f $P/${TAG}_launch_log_$name.jsonl ] && sort -u $P/${TAG}_launch_log_$name.jsonl -o $P/${TAG}_launch_log_$name.jsonl
  local tr=$(find $O/prof_$name -name "*kernel_trace.csv" | head -1)
  [ -n "$tr" ] && python $R/scripts/summarize_trace.py $tr $P/${TAG}_kernel_stats_$name.csv --skip-load
  # what the profiled command itself reported: the bench's JSON line, or (prefill_probe.py) its "... prefill tok/s" lines — never an empty file
  if grep -q '"metric"' $O/prof_$name.log; then grep -h '"metric"' $O/prof_$name.log | tail -1 > $P/${TAG}_bench_line_$name.json
  else grep -h "prefill tok/s" $O/prof_$name.log > $P/${TAG}_probe_line_$name.txt; fi
  echo "stats $name rc=$?"
}
BENCH="python $R/bench.py --decode-only --no-cpu-baseline --sweep= --verify-steps 0 --steps 50 --warmup 5"
# the modes that hold north_star's 1e-3: Precision::Fp32, and Precision::Fp16 with the sensitive launches promoted
# (ONLY_MODES=1: just these three runs)
run_stats v6-3b_int8_b32_fp32 $BENCH --workload v6-3b --quant int8 --batch 32 --precision fp32
RWKV_PROMOTE=1 run_stats v6-3b_int8_b32_promote1 $BENCH --workload v6-3b --quant int8 --batch 32
RWKV_PROMOTE=7 run_stats v7-2.9b_nf4_b32_promote7 $BENCH --workload v7-2.9b --quant nf4 --batch 32
if [ -n "${ONLY_MODES:-}" ]; then ls $P; exit 0; fi
run_stats v6-3b_int8_b32 $BENCH --workload v6-3b --quant int8 --batch 32
run_stats v6-3b_int8_b1 $BENCH --workload v6-3b --quant int8 --batch 1
run_stats v6-7b_fp16_b8 $BENCH --workload v6-7b --quant none --batch 8
run_stats v7-2.9b_nf4_b32 $BENCH --workload v7-2.9b --quant nf4 --batch 32
run_stats v6-3b_fp16_b32 $BENCH --workload v6-3b --quant none --batch 32
run_stats prefill_v6-3b_int8_32x256 python $R/scripts/prefill_probe.py v6-3b 1 32 256 2048
DEBUG_CLR_GRAPH_PACKET_CAPTURE=0 run_stats prefill_v6-3b_int8_chunk256 python $R/scripts/prefill_probe.py v6-3b 1 32 256 256
DEBUG_CLR_GRAPH_PACKET_CAPTURE=0 run_stats prefill_v7-2.9b_nf4_chunk256 python $R/scripts/prefill_probe.py v7-2.9b 2 32 256 256
# MFMA utilisation (north_star): matrix-pipe busy cycles against shader busy cycles, decode B=32 and the prefill run
for name in decode_v6-3b_int8_b32 prefill_v6-3b_int8; do
  rm -rf $O/pmc_$name
  if [ $name = decode_v6-3b_int8_b32 ]; then CMD="$BENCH --workload v6-3b --quant int8 --batch 32 --steps 12"; else CMD="python $R/scripts/prefill_probe.py v6-3b 1 32 256 2048"; fi
  timeout 400 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $O/pmc_$name -o p -- $CMD > $O/pmc_$name.log 2>&1
  python - $O/pmc_$name $P/${TAG}_pmc_mfma_$name.txt <<'PY'
import csv, glob, collections, sys
src, dst = sys.argv[1], sys.argv[2]
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for f in glob.glob(src + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].replace("void rwkv::", "").replace("rwkv::", "")[:64]
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
        if r["Counter_Name"] == "SQ_BUSY_CYCLES": n[k] += 1
with open(dst, "w") as out:
    out.write("kernel | launches | MFMA busy / SQ busy (matrix-pipe utilisation; SQ_BUSY is per-SE, so the ratio is relative across kernels) | wave cycles parked (SQ_WAIT_ANY / SQ_WAVE_CYCLES)\n")
    for k, c in sorted(acc.items(), key=lambda kv: -kv[1].get("SQ_BUSY_CYCLES", 0))[:14]:
        b = c.get("SQ_BUSY_CYCLES", 0) or 1; w = c.get("SQ_WAVE_CYCLES", 0) or 1
        out.write(f"{k} | {n[k]} | mfma_busy={c.get('SQ_VALU_MFMA_BUSY_CYCLES', 0):.4g} sq_busy={b:.4g} ratio={c.get('SQ_VALU_MFMA_BUSY_CYCLES', 0) / b:.3f} | parked={c.get('SQ_WAIT_ANY', 0) / w:.3f} active={c.get('SQ_ACTIVE_INST_ANY', 0) / w:.3f}\n")
PY
  echo "pmc $name done"
done
cd $R
timeout 400 python scripts/collect_pmc.py --round $TAG --batch 32 > $O/collect_pmc.log 2>&1; echo "traffic rc=$?"
cp $O/${TAG}_pmc_traffic_v6-3b_int8_b32.json $P/ 2>/dev/null
ls $P
