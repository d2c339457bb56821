#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_bench_paths.py -q -x -k "not tile_shape and not pipelined" > $O/t27.log 2>&1; echo "tests rc=$?"; tail -3 $O/t27.log
: > $O/samp27.log
for V in "RWKV_SAMP_H2D=1" "RWKV_SAMP_H2D=0" "RWKV_SAMP_H2D=1" "RWKV_SAMP_H2D=0"; do
  echo "== $V" >> $O/samp27.log
  env $V timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --sweep= > $O/b27.json 2> $O/b27.err
  python - >> $O/samp27.log <<'PY'
import json
d = json.loads(open("/root/repo/gpurun_out/b27.json").read().strip().splitlines()[-1])
print(round(d["value"]), round(d["ms_per_step"], 4), "pcie", round(d["pcie_inclusive_tokens_per_s"]), round(d["pcie_inclusive_tokens_per_s"]/d["value"], 4), "sample", round(d["on_device_sampling_tokens_per_s"]), round(d["on_device_sampling_tokens_per_s"]/d["value"], 4), "emb", round(d["embeddings"]["value"], 1))
PY
done
cat $O/samp27.log
