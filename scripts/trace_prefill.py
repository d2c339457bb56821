"""Dev tool: in-kernel timeline of wkv_chunk_kernel inside a real prefill step (-DRWKV_TRACE build, scripts/trace_gemm.py build 1)."""
import ctypes, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from ai00_server_amd import runtime as rt
rt.LIB_PATH = os.path.join(ROOT, "ai00_server_amd", "librwkv_hip_trace.so")
from oracle import rwkv_ref as R
name = os.environ.get("WORKLOAD", "v6-3b")
st, tensors = R.synth_st(name, fast=True)
info = R.model_info(tensors)
NSEQ, LEN, CHUNK = int(os.environ.get("NSEQ", 8)), int(os.environ.get("LEN", 512)), int(os.environ.get("CHUNK", 512))
eng = rt.ModelBuilder(st).quant(info.num_layer, rt.Quant(1)).build(max_batch=NSEQ, token_chunk_size=CHUNK, precision=rt.Precision.Fp16)
V = info.num_vocab
prompts = [[t % V for t in R.synth_prompt(s, LEN)] for s in range(NSEQ)]
inp = rt.RnnInput([rt.RnnInputBatch(list(p), rt.RnnOption.Last) for p in prompts])
while inp.num_token() > 0:
    inp, outs = eng.infer(inp)
buf = np.zeros(4 * 2048 * 8, dtype=np.uint64)
rt.lib().rwkv_debug_trace2(buf.ctypes.data_as(ctypes.c_void_p))
tr = buf.reshape(4, 2048, 8).astype(np.int64)[3]
act = tr[:, 0] > 0
newest = tr[:, 0][act].max()
act &= tr[:, 0] > newest - 20000
t0 = tr[:, 0][act].min()
print(f"wkv_chunk_kernel: {act.sum()} blocks traced (last launch), {NSEQ} sequences x {CHUNK // NSEQ} rows per step")
print("  since the launch's first block entered:")
for i, lab in enumerate(["entry", "A1 loads issued (chunk 0)", "A2 done", "after barrier", "B done (32 tokens)", "C done", "all chunks done", "state stored"]):
    v = (tr[:, i][act] - t0) / 100.0
    print(f"    {lab:28s} min {v.min():6.2f}  med {np.median(v):6.2f}  max {v.max():6.2f} us")
print("  since the block's own entry:")
for i, lab in enumerate(["entry", "A1 loads issued (chunk 0)", "A2 done", "after barrier", "B done (32 tokens)", "C done", "all chunks done", "state stored"]):
    v = (tr[:, i][act] - tr[:, 0][act]) / 100.0
    print(f"    {lab:28s} min {v.min():6.2f}  med {np.median(v):6.2f}  max {v.max():6.2f} us")
