#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O
cd $R
bash scripts/gpu_profiles.sh r2 > $O/profiles20.log 2>&1; echo "profiles rc=$?"
tail -5 $O/profiles20.log
cd $R
timeout 300 python scripts/prefill_probe.py v6-7b 0 8 1024 2048 > $O/pf20.log 2>&1; tail -1 $O/pf20.log
timeout 900 python bench.py --steps 20 --warmup 5 --config5 > $O/bench20.json 2> $O/bench20.err; echo "bench rc=$?"
python - <<'PY'
import json
d = json.loads(open("/root/repo/gpurun_out/bench20.json").read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"], d["embeddings"]["value"], d["config5"])
PY
