#!/bin/bash
# round 4, session 8: StateJob + asynchronous layer read-back against the oracle; the bench as two ranks on one device; the default bench line
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
O=$R/gpurun_out
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_embeddings.py -x -q -k "state_job" 2>&1 | tail -5
timeout 900 python -m pytest tests/test_bench_launcher.py -x -q -m gpu 2>&1 | tail -5
( time timeout 900 python bench.py --steps 20 --warmup 5 > $O/r4c8_bench_default.json 2> $O/r4c8_bench_default.err ) 2>&1 | tail -3
tail -3 $O/r4c8_bench_default.err
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r4c8_bench_default.json").read().strip().splitlines()[-1])
print({k: d[k] for k in ("value", "ms_per_step", "pcie_inclusive_tokens_per_s", "on_device_sampling_tokens_per_s", "sweep", "tokens_verified")})
print("embeddings", json.dumps(d["embeddings"])[:900])
print("roofline", {k: d["roofline"][k] for k in ("frac", "achieved", "avg_launch_us", "traffic")}, d["roofline"]["step"]["frac_of_peak"])
for k, v in (d.get("configs") or {}).items():
    print(k, json.dumps(v.get("decode")), json.dumps(v.get("prefill"))[:300] if v.get("prefill") else "", json.dumps(v.get("embeddings"))[:300] if v.get("embeddings") else "")
print("cpu", d["cpu_baseline"]["value"], d["cpu_baseline"].get("config1_v5_0.4b_b1_tokens_per_s"))
PY
