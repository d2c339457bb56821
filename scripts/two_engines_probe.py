"""Round 6 probe (GPU box): is a decode step latency-bound enough that TWO independent half-batch engines on ONE device beat one full-batch engine?
Two engines of 16 slots each, driven from two host threads (rwkv_decode_greedy releases the GIL inside the C call), against one engine of 32 / 16
slots.  Each engine streams the weights itself (its own copy): the aggregate moves twice the bytes per generated token.

    python scripts/two_engines_probe.py [steps]
"""
import os, sys, threading, time
import numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from ai00_server_amd import runtime as rt
from oracle import rwkv_ref as R      # checkpoint synthesis only

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 60
import bench
st, tensors = R.synth_st("v6-3b", fast=True)
info = R.model_info(tensors)
V = info.num_vocab

def engine(B):
    return bench.build_engine(rt, st, 0, info.num_layer, 1, B, 2048, "fp16")

def run(engs, Bs, label):
    firsts = [bench.first_tokens(R, V, B) for B in Bs]
    for e, f in zip(engs, firsts): e.decode_greedy(f, 5)
    out = [None] * len(engs)
    def work(i): out[i] = engs[i].decode_greedy(firsts[i], steps)
    th = [threading.Thread(target=work, args=(i,)) for i in range(len(engs))]
    t = time.perf_counter()
    for x in th: x.start()
    for x in th: x.join()
    dt = time.perf_counter() - t
    tok = sum(Bs) * steps
    print(f"{label}: {dt * 1e3 / steps:7.3f} ms per (concurrent) step, {tok / dt:8.0f} tokens/s aggregate; device ms/step per engine: "
          + ", ".join(f"{o[1] / steps:.3f}" for o in out), flush=True)

e32 = engine(32); run([e32], [32], "one engine, 32 slots       "); e32.close()
a = engine(16); run([a], [16], "one engine, 16 slots       ")
b = engine(16); run([a, b], [16, 16], "two engines, 16 + 16 slots ")
c = engine(16); d = engine(16); run([a, b, c, d], [16] * 4, "four engines, 4 x 16 slots ")
