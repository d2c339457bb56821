"""Bisect helper (GPU box): logits error of a 2-layer 7B-width V6 model under different step shapes / tile shapes."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ai00_server_amd import runtime as rt
from oracle import rwkv_ref as R

def feed(eng, prompts):
    B = eng.max_batch
    inp = rt.RnnInput([rt.RnnInputBatch(list(prompts[b]) if b < len(prompts) else [], rt.RnnOption.Last) for b in range(B)])
    rows = [[] for _ in range(B)]
    while inp.num_token() > 0:
        inp, outs = eng.infer(inp)
        for b, o in enumerate(outs):
            rows[b].extend(list(o))
    return rows

C, F = int(os.environ.get("DBG_C", 4096)), int(os.environ.get("DBG_F", 14336))
tens = R.synth_checkpoint(6, 2, C, F, 2048, seed=29)
st = R.st_serialize(tens)
rb = R.RwkvRefBatch(tens)
B = 8
ps = [[t % 2048 for t in R.synth_prompt(500 + b, 128)] for b in range(B)]
states = rb.init_states(B)
want = rb.prefill(ps, states)
scale = max(1.0, float(np.abs(want).max()))
for chunk, shape, nslot in [(1, None, 1), (16, None, 1), (16, None, 8), (128, None, 8), (256, None, 8), (1024, 3, 8), (1024, 4, 8), (1024, 7, 8), (1024, None, 8)]:
    if shape is not None:
        os.environ["RWKV_TILE_SHAPE"] = str(shape)
    else:
        os.environ.pop("RWKV_TILE_SHAPE", None)
    eng = rt.ModelBuilder(st).build(max_batch=B, token_chunk_size=chunk, precision=rt.Precision.Fp16)
    rows = feed(eng, ps[:nslot])
    err = max(float(np.abs(rows[b][-1] - want[b]).max()) for b in range(nslot))
    serr = max(float(np.abs(eng.state.back(b) - states[b]).max()) for b in range(nslot))
    print(f"C={C} chunk={chunk} shape={shape} slots={nslot}: logits err {err:.3e} (tol {1e-3 * scale:.3e}) state err {serr:.3e}", flush=True)
    eng.close()
