"""Dev tool: in-kernel timelines (wall_clock64 probes, -DRWKV_TRACE) of the small kernels inside a real decode step.

    python scripts/trace_gemm.py build 1      # build host: ai00_server_amd/librwkv_hip_trace.so
    B=32 python scripts/trace_step.py         # GPU box: last layer of the last step
"""
import ctypes, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from ai00_server_amd import runtime as rt
rt.LIB_PATH = os.path.join(ROOT, "ai00_server_amd", "librwkv_hip_trace.so")
from oracle import rwkv_ref as R

B = int(os.environ.get("B", "32"))
st, tensors = R.synth_st(os.environ.get("WORKLOAD", "v6-3b"), fast=True)
info = R.model_info(tensors)
eng = rt.ModelBuilder(st).quant(info.num_layer, rt.Quant(1)).build(max_batch=B, token_chunk_size=512, precision=rt.Precision.Fp16)
first = np.arange(B, dtype=np.uint32) + 100
eng.decode_greedy(first, 20)
toks, ms = eng.decode_greedy(first, 50)
print(f"B={B}: {ms / 50:.3f} ms/step")
lib = rt.lib()
buf = np.zeros(4 * 2048 * 8, dtype=np.uint64)
lib.rwkv_debug_trace2(buf.ctypes.data_as(ctypes.c_void_p))
tr = buf.reshape(4, 2048, 8).astype(np.int64)
names = {0: ("v6_mix", ["entry", "phase1 mfma done", "after barrier", "m_c ready", "exit"]),
         1: ("wkv", ["entry", "loads issued", "after barrier", "recurrence done", "before state store", "exit"]),
         2: ("ln_shift (ffn)", ["entry", "x + partials landed", "layernorm done", "exit"])}
for kid, (name, labs) in names.items():
    act = tr[kid, :, 0] > 0
    if not act.any():
        continue
    # keep only blocks of the most recent launch (entries within 100 us of the newest)
    newest = tr[kid, :, 0][act].max()
    act &= tr[kid, :, 0] > newest - 10000
    t0 = tr[kid, :, 0][act].min()
    print(f"{name}: {act.sum()} blocks traced")
    for i, lab in enumerate(labs):
        v = (tr[kid, :, i][act] - t0) / 100.0
        print(f"    {lab:22s} min {v.min():6.2f}  med {np.median(v):6.2f}  max {v.max():6.2f} us")
