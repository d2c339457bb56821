"""Dev tool: load time from safetensors vs from a prefab image (v6-3b int8)."""
import os, sys, time, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import rwkv_ref as R
from ai00_server_amd import runtime as rt
st, tens = R.synth_st("v6-3b", fast=True)
info = R.model_info(tens)
t0 = time.perf_counter(); eng = rt.ModelBuilder(st).quant(info.num_layer, rt.Quant(1)).build(max_batch=8); t1 = time.perf_counter()
path = os.path.join(tempfile.gettempdir(), "m.prefab")
eng.save_prefab(path); t2 = time.perf_counter()
eng.close()
img = open(path, "rb").read(); t3 = time.perf_counter()
eng2 = rt.ModelBuilder(img).build(max_batch=8); t4 = time.perf_counter()
print(f"safetensors ({len(st)/1e9:.2f} GB) -> engine: {t1-t0:.2f} s; save prefab ({len(img)/1e9:.2f} GB): {t2-t1:.2f} s; prefab -> engine: {t4-t3:.2f} s")
