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
  echo "stats $name rc=$? $(wc -c < $P/${TAG}_kernel_stats_$name.csv 2>/dev/null) bytes of summary"
  rm -rf $O/prof_$name                                     # the raw traces are hundreds of MB; gpurun_out/ is copied back only below 64 MiB
}
BENCH="python $R/bench.py --decode-only --no-cpu-baseline --sweep= --verify-steps 0 --steps 50 --warmup 5"
# every engine in the library's default precision (ABI 7: Precision::Fp16 = f16 operands, the error-carrying launches hi + lo); the other two modes of the
# headline configuration and the raw mode of config #4 beside them
run_stats v6-3b_int8_b32 $BENCH --workload v6-3b --quant int8 --batch 32
run_stats v6-3b_int8_b32_fp16raw $BENCH --workload v6-3b --quant int8 --batch 32 --precision fp16raw
run_stats v6-3b_int8_b32_fp32 $BENCH --workload v6-3b --quant int8 --batch 32 --precision fp32
if [ -n "${ONLY_MODES:-}" ]; then ls $P; exit 0; fi
run_stats v6-3b_int8_b1 $BENCH --workload v6-3b --quant int8 --batch 1
run_stats v6-7b_fp16_b8 $BENCH --workload v6-7b --quant none --batch 8
run_stats v7-2.9b_nf4_b32 $BENCH --workload v7-2.9b --quant nf4 --batch 32
run_stats v7-2.9b_nf4_b32_fp16raw $BENCH --workload v7-2.9b --quant nf4 --batch 32 --precision fp16raw
run_stats v6-3b_fp16_b32 $BENCH --workload v6-3b --quant none --batch 32
run_stats prefill_v6-3b_int8_32x256 python $R/scripts/prefill_probe.py v6-3b 1 32 256 2048
DEBUG_CLR_GRAPH_PACKET_CAPTURE=0 run_stats prefill_v6-3b_int8_chunk256 python $R/scripts/prefill_probe.py v6-3b 1 32 256 256
DEBUG_CLR_GRAPH_PACKET_CAPTURE=0 run_stats prefill_v7-2.9b_nf4_chunk256 python $R/scripts/prefill_probe.py v7-2.9b 2 32 256 256
# MFMA utilisation (north_star): matrix-pipe busy cycles against shader busy cycles, decode B=32 and the prefill run
for name in decode_v6-3b_int8_b32 prefill_v6-3b_int8; do
  rm -rf $O/pmc_$name
  if [ $name = decode_v6-3b_int8_b32 ]; then CMD="$BENCH --workload v6-3b --quant int8 --batch 32 --steps 12"; else CMD="python $R/scripts/prefill_probe.py v6-3b 1 32 256 2048"; fi
  timeout 400 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d $O/pmc_$name -o p -- $CMD > $O/pmc_$name.log 2>&1
  python - $O/pmc_$name $P/${TAG}_pmc_mfma_$name.txt <<'PY'
import csv, glob, collections, sys
src, dst = sys.argv[1], sys.argv[2]
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for f in glob.glob(src + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].replace("void rwkv::", "").replace("rwkv::", "")[:64]
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
        if r["Counter_Name"] == "SQ_WAVE_CYCLES": n[k] += 1
SIMDS, XCDS = 256 * 4, 8
with open(dst, "w") as out:
    out.write("# MFMA utilisation AGAINST THE CHIP'S PEAK (round 6): SQ_VALU_MFMA_BUSY_CYCLES (matrix-pipe busy cycles, summed over the SIMDs) / (cycles the launch kept the GPU busy x 1024 SIMDs).\n"
              "# 1.0 = every SIMD's matrix pipe busy for the whole launch = the dense f16 peak at whatever clock the launch ran.  Busy cycles of the launch = GRBM_GUI_ACTIVE / 8: this rocprofv3\n"
              "# reports the counter once per XCD and the CSV sums the eight (a 150 us launch reads 2.1e6 = 8 x 150 us x ~1.8 GHz: the counter window of a profiled launch is a little longer than the kernel; scripts/clock_probe.hip measures 2.1 GHz during these launches).  Cross-check: the pipelined prefill kernel comes out at 0.29 here\n"
              "# and at 0.26-0.31 as FLOP / time / 2.5 PFLOP/s in the roofline table.\n")
    out.write("kernel | launches | mfma_busy_cycles (all launches) | GPU-busy cycles per launch | MFMA utilisation of the chip peak | wave cycles parked (SQ_WAIT_ANY / SQ_WAVE_CYCLES) | issuing (SQ_ACTIVE_INST_ANY / SQ_WAVE_CYCLES)\n")
    for k, c in sorted(acc.items(), key=lambda kv: -kv[1].get("GRBM_GUI_ACTIVE", 0))[:16]:
        w = c.get("SQ_WAVE_CYCLES", 0) or 1
        gui = c.get("GRBM_GUI_ACTIVE", 0) / XCDS
        util = c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / (gui * SIMDS) if gui else float("nan")
        out.write(f"{k} | {n[k]} | {c.get('SQ_VALU_MFMA_BUSY_CYCLES', 0):.4g} | {gui / max(1, n[k]):.4g} | {util:.4f} | parked={c.get('SQ_WAIT_ANY', 0) / w:.3f} | active={c.get('SQ_ACTIVE_INST_ANY', 0) / w:.3f}\n")
PY
  echo "pmc $name done"
  rm -rf $O/pmc_$name
done
cd $R
timeout 400 python scripts/collect_pmc.py --round $TAG --batch 32 > $O/collect_pmc.log 2>&1; echo "traffic rc=$?"
cp $O/${TAG}_pmc_traffic_v6-3b_int8_b32.json $P/ 2>/dev/null
rm -rf $O/pmc_fetch $O/pmc_write
ls -la $P; du -sh $O
