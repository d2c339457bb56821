mkdir -p gpurun_out; rm -f gpurun_out/full_depth_errors.jsonl
timeout 1300 python -m pytest tests -x -q -m gpu 2>&1 | tail -5
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
( time timeout 900 python bench.py --steps 20 --warmup 5 > gpurun_out/r5_bench_default.json 2> gpurun_out/r5_bench_default.err ) 2>&1 | grep real
tail -2 gpurun_out/r5_bench_default.err
