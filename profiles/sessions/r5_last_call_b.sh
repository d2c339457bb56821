mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_full_depth.py -x -q -s -k "Fp32 or promoted" 2>&1 | grep -v '^$' | tail -30
( time timeout 1200 python bench.py --steps 20 --warmup 5 > gpurun_out/r5_bench_default.json 2> gpurun_out/r5_bench_default.err ) 2>&1 | grep real
tail -2 gpurun_out/r5_bench_default.err
ONLY_MODES=1 bash scripts/gpu_profiles.sh r5 2>&1 | tail -4
