#!/bin/bash
# decode GEMM with branch-free single-shot weight loads (counted vmcnt waits: round r multiplied while rounds > r stream): parity, then the bench
cd $GRAFT_REPO_ROOT
timeout 170 python -m pytest -x -q tests/test_gpu_parity.py -k "quantised_layers or greedy_ids_identical or full_width_3b_shapes or full_width_v7 or chunk_size_batch" 2>&1 | tail -3
timeout 100 python bench.py --decode-only --no-cpu-baseline --sweep=1,8 --steps 50 --warmup 5 > gpurun_out/r5_gemm_counted_v6.json 2>/dev/null; python - <<'PY'
import json
d=json.load(open('gpurun_out/r5_gemm_counted_v6.json'))
print('v6-3b int8 b32', d['ms_per_step'], d['value'], d['roofline']['frac'], {k:v['ms_per_step'] for k,v in d['sweep'].items()}, 'verified', d.get('tokens_verified'))
PY
timeout 60 python bench.py --decode-only --no-cpu-baseline --sweep= --verify-steps 0 --steps 50 --warmup 5 --workload v7-2.9b --quant nf4 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('v7 nf4 b32', d['ms_per_step'])"
timeout 60 python bench.py --decode-only --no-cpu-baseline --sweep= --verify-steps 0 --steps 50 --warmup 5 --workload v6-7b --quant none --batch 8 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('7b fp16 b8', d['ms_per_step'])"
