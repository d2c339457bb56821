#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O
cd $R
for W in "v6-1.6b none 1" "v5-0.4b none 1" "v6-7b none 1" "v7-2.9b nf4 1" "v6-3b int8 16" "v6-3b int8 2" "v6-3b int8 4"; do
  set -- $W
  timeout 300 python bench.py --workload $1 --quant $2 --batch $3 --decode-only --no-cpu-baseline --sweep= --steps 40 --warmup 5 > $O/b39.json 2> $O/b39.err
  python - "$W" <<'PY'
import json, sys
d = json.loads(open("/root/repo/gpurun_out/b39.json").read().strip().splitlines()[-1])
print(sys.argv[1], "|", round(d["value"], 1), "tok/s", round(d["ms_per_step"], 4), "ms", "step frac", round(d["roofline"]["step"]["frac_of_peak"], 3), "verified", d["tokens_verified"])
PY
done
