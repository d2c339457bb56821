import sys, os, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import rwkv_ref as R
from ai00_server_amd import runtime as rt
name, quant, B, T, chunk = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5])
st, tens = R.synth_st(name)
info = R.model_info(tens)
eng = rt.ModelBuilder(st).quant(info.num_layer if quant else 0, rt.Quant(quant)).build(max_batch=B, token_chunk_size=chunk)
V = info.num_vocab
prompts = [[t % V for t in R.synth_prompt(s, T)] for s in range(B)]
if os.environ.get('PROBE_MAPS'):
    for ln in open('/proc/self/maps'):
        if 'r-xp' in ln and any(k in ln for k in ('hip', 'hsa', 'rocprof', 'rwkv', 'libc.so', 'roctx', 'amd_comgr')):
            print('MAP', ln.strip(), flush=True)
for rep in range(2):
    inp = rt.RnnInput([rt.RnnInputBatch(list(p), rt.RnnOption.Last) for p in prompts])
    t0 = time.perf_counter(); calls = 0
    while inp.num_token() > 0:
        inp, outs = eng.infer(inp); calls += 1
    dt = time.perf_counter() - t0
    print(f"{name} q{quant} B={B} T={T} chunk={chunk}: {B*T/dt:.0f} prefill tok/s ({dt*1e3:.1f} ms, {calls} infer calls)", flush=True)
