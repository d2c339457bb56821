set -x
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_knobs.py -x -q -k "SMALLK or KSW8" 2>&1 | tail -5 > gpurun_out/r5_t1.log
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "v7 or V7" 2>&1 | tail -5 >> gpurun_out/r5_t1.log
timeout 900 python -m pytest tests/test_gpu_bench_paths.py -x -q -k "v7 or V7 or nf4" 2>&1 | tail -5 >> gpurun_out/r5_t1.log
cat gpurun_out/r5_t1.log
AB_WORKLOAD=v7-2.9b AB_QUANT=nf4 timeout 600 python scripts/ab_bench.py "v7 smallk::" "v7 generic:RWKV_NO_SMALLK=1:" 2>&1 | tee gpurun_out/r5_ab_smallk.log
for e in 0 1; do echo "RWKV_NO_SMALLK=$e" | tee -a gpurun_out/r5_ab_smallk.log; RWKV_NO_SMALLK=$e timeout 300 python scripts/prefill_probe.py v7-2.9b 2 32 256 256 2>&1 | grep "tok/s" | tee -a gpurun_out/r5_ab_smallk.log; RWKV_NO_SMALLK=$e timeout 300 python scripts/prefill_probe.py v7-2.9b 2 32 256 2048 2>&1 | grep "tok/s" | tee -a gpurun_out/r5_ab_smallk.log; done
