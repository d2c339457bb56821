#!/bin/bash
# round 4, session 1: full-depth parity tests (config #3 / #4 at 32 layers), ablation of the decode GEMM launch (what X delivery, dequantisation
# and the MFMAs each cost), first-strip-first issue order, counters of the Int8 launch against the fp16 one.
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
O=$R/gpurun_out
mkdir -p $O
export TMPDIR=/tmp
: > $O/full_depth_errors.jsonl
timeout 900 python -m pytest tests/test_gpu_full_depth.py -x -q -s 2>&1 | grep -v "^$" | tail -40 > $O/r4c1_full_depth.log
tail -25 $O/r4c1_full_depth.log
echo "=== microbench"
: > $O/r4c1_micro.log
for v in base exp_W0FIRST exp_NOX exp_NODQ exp_NOMFMA; do
  if [ $v = base ]; then unset RWKV_HIP_LIB; else export RWKV_HIP_LIB=$R/ai00_server_amd/librwkv_hip_$v.so; fi
  timeout 300 python scripts/gemm_micro.py $v 2>&1 | tail -8 >> $O/r4c1_micro.log
done
unset RWKV_HIP_LIB
cat $O/r4c1_micro.log
echo "=== whole step A/B"
timeout 400 python scripts/ab_bench.py "base::" "w0first::ai00_server_amd/librwkv_hip_exp_W0FIRST.so" 2>&1 | tee $O/r4c1_ab.log
AB_QUANT=none timeout 300 python scripts/ab_bench.py "base-f16::" "w0first-f16::ai00_server_amd/librwkv_hip_exp_W0FIRST.so" 2>&1 | tee -a $O/r4c1_ab.log
echo "=== counters: Int8 vs fp16 launch (rkvg shape, T = 32)"
cat > /tmp/one.py <<'PY'
import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from ai00_server_amd import runtime as rt
for fmt in (int(os.environ["FMT"]),):
    for T in (int(os.environ["TT"]),):
        nmat = 24 if fmt == 1 else 12
        us, blk = rt.bench_gemm(10240, 2560, fmt, T, False, 0, nmat, 100)
        print(f"fmt {fmt} T {T}: {us:.2f} us {blk:.0f} blocks", flush=True)
PY
cd /tmp
: > $O/r4c1_pmc.txt
for SET in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_VALU_MFMA_BUSY_CYCLES" "SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_PENDING_STALL_CYCLES_sum TCP_GATE_EN1_sum" "TA_BUSY_sum TA_TA_BUSY_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_TA_TCP_STATE_READ_sum"; do
  for FT in "1 1" "1 32" "0 1" "0 32"; do
  set -- $FT; export FMT=$1 TT=$2
  echo "fmt=$FMT T=$TT" >> $O/r4c1_pmc.txt
  rm -rf $O/pmc_c1
  timeout 200 rocprofv3 --kernel-trace --pmc $SET --output-format csv -d $O/pmc_c1 -o p -- python /tmp/one.py > $O/pmc_c1.log 2>&1; echo "rc=$? ($SET)"
  python - $O/pmc_c1 >> $O/r4c1_pmc.txt <<'PY'
import csv, glob, collections, sys
# launches arrive in the order of /tmp/one.py: (int8,T1) (int8,T32) (f16,T1) (f16,T32): key on kernel name + VGPR count is not enough, so group by dispatch order
rows = []
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    rows += list(csv.DictReader(open(f)))
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(collections.Counter)
for r in rows:
    k = r["Kernel_Name"].replace("void rwkv::", "").replace("rwkv::", "")[:48]
    if "gemm_kernel" not in k: continue
    key = (k, r.get("Grid_Size", ""), r.get("LDS_Block_Size", ""), r.get("VGPR_Count", r.get("Arch_VGPR_Count", "")))
    acc[key][r["Counter_Name"]] += float(r["Counter_Value"]); n[key][r["Counter_Name"]] += 1
for k, c in sorted(acc.items()):
    print(k, "|", " ".join(f"{cn}={v / max(1, n[k][cn]):.4g}" for cn, v in sorted(c.items())), "| launches", max(n[k].values()))
PY
  grep "fmt" $O/pmc_c1.log >> $O/r4c1_pmc.txt
  done
done
cat $O/r4c1_pmc.txt
