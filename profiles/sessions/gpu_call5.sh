#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O
cd $R
timeout 900 python -m pytest tests -m gpu -x -q > $O/t5_full.log 2>&1; echo "tests rc=$?"
timeout 400 python bench.py --steps 20 --warmup 5 > $O/bench5.json 2> $O/bench5.err; echo "bench rc=$?"
TS=512,1024 SHAPES=0,3,4,7,9 timeout 300 python scripts/tile_bench.py > $O/tile5.log 2>&1; echo "tile rc=$?"
export TMPDIR=/tmp
cd /tmp
timeout 300 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_ANY --output-format csv -d $O/pmc_prefill -o p -- python $R/scripts/prefill_probe.py v6-3b 1 8 512 512 > $O/pmc_prefill.log 2>&1; echo "pmc rc=$?"
python - <<'PY'
import csv, glob, collections, os
O = os.environ.get("GRAFT_REPO_ROOT", "/root/repo") + "/gpurun_out"
for f in glob.glob(O + "/pmc_prefill/**/*counter_collection.csv", recursive=True):
    acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"][:70]
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); 
        if r["Counter_Name"] == "SQ_WAVE_CYCLES": n[k] += 1
    with open(O + "/pmc_prefill_summary.txt", "w") as out:
        for k, c in sorted(acc.items(), key=lambda kv: -kv[1].get("SQ_BUSY_CYCLES", 0))[:12]:
            out.write(f"{k} launches={n[k]} " + " ".join(f"{a}={v:.3g}" for a, v in sorted(c.items())) + "\n")
PY
cat $O/pmc_prefill_summary.txt 2>/dev/null | head -20
