#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O
cd $R
timeout 600 python -m pytest tests/test_gpu_bench_paths.py tests/test_gpu_parity.py -q -x -k "greedy or small_batches or 32_slots" > $O/t22.log 2>&1; echo "tests rc=$?"; tail -3 $O/t22.log
timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench22.json 2> $O/bench22.err; echo "bench rc=$?"
python - <<'PY'
import json
d = json.loads(open("/root/repo/gpurun_out/bench22.json").read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"], d["device_ms_per_step"], d["embeddings"]["value"], d["pcie_inclusive_tokens_per_s"], d["on_device_sampling_tokens_per_s"], d["sweep"])
PY
