mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_knobs.py -x -q 2>&1 | tail -3
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q 2>&1 | tail -3
timeout 900 python -m pytest tests/test_gpu_bench_paths.py -x -q 2>&1 | tail -3
timeout 600 python scripts/ab_bench.py "v6 pruned::" 2>&1
AB_WORKLOAD=v7-2.9b AB_QUANT=nf4 timeout 600 python scripts/ab_bench.py "v7 pruned::" 2>&1
AB_WORKLOAD=v6-7b AB_QUANT=none timeout 600 python scripts/ab_bench.py "7b fp16::" 2>&1
