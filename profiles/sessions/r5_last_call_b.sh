mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q 2>&1 | tail -3
timeout 900 python -m pytest tests/test_gpu_embeddings.py -x -q 2>&1 | tail -3
( time timeout 1200 python bench.py --steps 20 --warmup 5 > gpurun_out/r5_bench_default.json 2> gpurun_out/r5_bench_default.err ) 2>&1 | grep real
tail -3 gpurun_out/r5_bench_default.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r5_bench_default.json').readline())
print('value',d['value'],'ms',d['ms_per_step'])
print('emb',d['embeddings']['value'],d['embeddings']['prefill_tokens_per_s'],d['embeddings']['at_token_chunk_size_2048'])
f=d['precision_fp32']; print('fp32 emb', f['embeddings']['value'])
for k,c in d['configs'].items():
    print(k, c.get('embeddings',{}).get('value'), c.get('embeddings',{}).get('prefill_tokens_per_s'), {ch:round(v['tokens_per_s']) for ch,v in c.get('prefill',{}).items()}, c.get('fp16_promoted',{}).get('embeddings',{}).get('value'))
PY
