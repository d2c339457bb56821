#!/bin/bash
# Dev tool (GPU box): counters of the pipelined tile kernel against the 64x64 shape on one whole-round GEMM (8192 x 2560, T = 1024, Int8).
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O
export TMPDIR=/tmp
cd /tmp
cat > /tmp/one_gemm.py <<'PY'
import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from ai00_server_amd import runtime as rt
for shape in (4, 10):
    us, blk = rt.bench_gemm(8192, 2560, 1, 1024, False, shape, 4, 20)
    print(f"shape {shape}: {us:.1f} us, {2.0 * 8192 * 2560 * 1024 / us / 1e6:.0f} TFLOP/s, {blk:.0f} blocks", flush=True)
PY
: > $O/pmc_tile3.txt
for SET in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_PENDING_STALL_CYCLES_sum TCP_GATE_EN1_sum"; do
  rm -rf $O/pmc_t3
  timeout 200 rocprofv3 --kernel-trace --pmc $SET --output-format csv -d $O/pmc_t3 -o p -- python /tmp/one_gemm.py > $O/pmc_t3.log 2>&1; echo "rc=$? ($SET)"
  python - $O/pmc_t3 >> $O/pmc_tile3.txt <<'PY'
import csv, glob, collections, sys
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(collections.Counter)
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].replace("void rwkv::", "").replace("rwkv::", "")[:60]
        if "gemm_tile" not in k: continue
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[k][r["Counter_Name"]] += 1
for k, c in acc.items():
    print(k, "|", " ".join(f"{cn}={v / max(1, n[k][cn]):.4g}/launch" for cn, v in sorted(c.items())))
PY
done
grep "shape" $O/pmc_t3.log >> $O/pmc_tile3.txt
cat $O/pmc_tile3.txt
