#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O
cd $R
timeout 1500 python -m pytest tests -m gpu -x -q > $O/t38.log 2>&1; echo "tests rc=$?"; tail -3 $O/t38.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
timeout 600 python bench.py --steps 20 --warmup 5 > $O/bench38.json 2> $O/bench38.err; echo "bench rc=$?"
python - <<'PY'
import json
d = json.loads(open("/root/repo/gpurun_out/bench38.json").read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"], d["embeddings"]["value"], d["embeddings"]["at_token_chunk_size_256"], d["pcie_inclusive_tokens_per_s"], d["on_device_sampling_tokens_per_s"], d["sweep"], d["roofline"]["frac"], d["cpu_baseline"]["value"])
PY
