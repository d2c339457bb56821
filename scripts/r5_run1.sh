set -x
mkdir -p gpurun_out
rm -f gpurun_out/full_depth_errors.jsonl
timeout 1500 python -m pytest tests/test_gpu_full_depth.py -x -q -s -k "promoted or (embeddings and v7)" 2>&1 | grep -v '^$' | tail -25 > gpurun_out/r5_t1.log
timeout 600 python -m pytest tests/test_gpu_embeddings.py -x -q 2>&1 | tail -5 >> gpurun_out/r5_t1.log
cat gpurun_out/r5_t1.log
AB_WORKLOAD=v7-2.9b AB_QUANT=nf4 timeout 600 python scripts/ab_bench.py "v7 fp16::" "v7 promote7:RWKV_PROMOTE=7:" 2>&1 | tee gpurun_out/r5_ab_promote.log
timeout 600 python scripts/ab_bench.py "v6 fp16::" "v6 promote5:RWKV_PROMOTE=5:" 2>&1 | tee -a gpurun_out/r5_ab_promote.log
