#!/bin/bash
# host-side wait policy on the serving loops: does the stream sync wake up faster when it spins?
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O
cd $R
: > $O/wait25.log
for V in "X=1" "ROC_ACTIVE_WAIT_TIMEOUT=5000" "HSA_ENABLE_INTERRUPT=0" "ROC_ACTIVE_WAIT_TIMEOUT=5000 HSA_ENABLE_INTERRUPT=0"; do
  echo "== $V" >> $O/wait25.log
  env $V timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --sweep= > $O/b25.json 2> $O/b25.err
  python - >> $O/wait25.log <<'PY'
import json
d = json.loads(open("/root/repo/gpurun_out/b25.json").read().strip().splitlines()[-1])
print(round(d["value"]), round(d["ms_per_step"], 4), "pcie", round(d["pcie_inclusive_tokens_per_s"]), round(d["pcie_inclusive_tokens_per_s"]/d["value"], 4), "sample", round(d["on_device_sampling_tokens_per_s"]), round(d["on_device_sampling_tokens_per_s"]/d["value"], 4), "emb", round(d["embeddings"]["value"], 1))
PY
done
cat $O/wait25.log
