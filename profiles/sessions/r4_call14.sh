#!/bin/bash
# round 4, session 14: counters of the 256-row r/k/v/g launch (Int8): pipelined 128x64 tiles (shape 11) against 64x64 tiles (shape 4)
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O
export TMPDIR=/tmp
cd /tmp
cat > /tmp/one256.py <<'PY'
import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from ai00_server_amd import runtime as rt
sh = int(os.environ["SHAPE"])
us, blk = rt.bench_gemm(10240, 2560, 1, 256, False, sh, 24, 60)
print(f"shape {sh}: {us:.1f} us {blk:.0f} blocks", flush=True)
PY
: > $O/r4c14_pmc_t256.txt
for SHAPE in 11 4; do
export SHAPE
for SET in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_VALU_MFMA_BUSY_CYCLES" "SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_PENDING_STALL_CYCLES_sum TCP_GATE_EN1_sum" "SQ_INSTS_MFMA SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_IFETCH SQ_IFETCH_LEVEL SQ_ACTIVE_INST_FLAT"; do
  rm -rf $O/pmc_c14
  timeout 200 rocprofv3 --kernel-trace --pmc $SET --output-format csv -d $O/pmc_c14 -o p -- python /tmp/one256.py > $O/pmc_c14.log 2>&1; echo "rc=$? shape $SHAPE ($SET)"
  python - $O/pmc_c14 $SHAPE >> $O/r4c14_pmc_t256.txt <<'PY'
import csv, glob, collections, sys
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(collections.Counter)
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].replace("void rwkv::", "").replace("rwkv::", "")[:40]
        if "gemm_tile" not in k: continue
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[k][r["Counter_Name"]] += 1
for k, c in sorted(acc.items()):
    print("shape", sys.argv[2], k, "|", " ".join(f"{cn}={v / max(1, n[k][cn]):.4g}" for cn, v in sorted(c.items())))
PY
done
done
cat $O/r4c14_pmc_t256.txt
