#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O
cd $R
timeout 600 python -m pytest tests/test_gpu_parity.py -q -x -k "sampl or mirostat or typical or formatter" > $O/t23.log 2>&1; echo "tests rc=$?"; tail -3 $O/t23.log
timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --sweep= > $O/bench23.json 2> $O/bench23.err; echo "bench rc=$?"
python - <<'PY'
import json
d = json.loads(open("/root/repo/gpurun_out/bench23.json").read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"], d["embeddings"]["value"], d["pcie_inclusive_tokens_per_s"], d["on_device_sampling_tokens_per_s"], d["on_device_sampling_tokens_per_s"]/d["value"])
PY
