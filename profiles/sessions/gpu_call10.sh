#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O
cd $R
timeout 1200 python -m pytest tests -m gpu -q > $O/t10_full.log 2>&1; echo "tests rc=$?"
timeout 600 python bench.py --steps 20 --warmup 5 --config5 > $O/bench10.json 2> $O/bench10.err; echo "bench rc=$?"
BENCH_SHARE_GPU=1 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --steps 20 --warmup 5 --decode-only --no-cpu-baseline --sweep= > $O/bench10_n2.json 2> $O/bench10_n2.err; echo "n2 rc=$?"
bash scripts/gpu_profiles.sh r2 > $O/profiles.log 2>&1; echo "profiles rc=$?"
