set -x
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_knobs.py -x -q -k "ROWJOB or NO_DENSE" 2>&1 | tail -5 > gpurun_out/r5_t1.log
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "decode or greedy or v7 or 256" 2>&1 | tail -5 >> gpurun_out/r5_t1.log
cat gpurun_out/r5_t1.log
timeout 600 python scripts/ab_bench.py "v6 rowjob-wt::" "v6 norowjob:RWKV_ROWJOB=0:" 2>&1 | tee gpurun_out/r5_ab_rowjob.log
AB_WORKLOAD=v7-2.9b AB_QUANT=nf4 timeout 600 python scripts/ab_bench.py "v7 rowjob-wt::" "v7 norowjob:RWKV_ROWJOB=0:" 2>&1 | tee -a gpurun_out/r5_ab_rowjob.log
