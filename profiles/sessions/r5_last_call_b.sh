timeout 900 python -m pytest tests/test_gpu_parity.py -x -q 2>&1 | tail -2
timeout 900 python -m pytest tests/test_gpu_bench_paths.py -x -q -k "tile_shape or 7b or decode" 2>&1 | tail -2
timeout 600 python scripts/ab_bench.py "v6 prob_of::" 2>&1 | grep -v amdgpu.ids
AB_WORKLOAD=v7-2.9b AB_QUANT=nf4 timeout 600 python scripts/ab_bench.py "v7 prob_of::" 2>&1 | grep -v amdgpu.ids
FMTS=1 TS=1,32 SHAPES=rkvg,fkfr python scripts/gemm_micro.py prob_of 2>&1 | grep -v amdgpu.ids
