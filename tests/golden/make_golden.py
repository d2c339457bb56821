"""Regenerate tests/golden/*.npz from the oracle (run from the repo root: python tests/golden/make_golden.py).

The reference (ai00_server) ships no golden vectors for this path, and its implementation
(web-rwkv, Rust + wgpu) cannot run here, so these fixtures pin the ORACLE against regressions; they are not
outputs of the reference.  Each file: model config name, seed, prompt, logits of the last prompt token (fp32),
16 greedy token ids, and a checksum of the final state slab."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import rwkv_ref as R  # noqa: E402

CASES = [("v5-tiny", 0, 0), ("v6-tiny", 0, 0), ("v7-tiny", 0, 0), ("v6-small", 2, R.QUANT_INT8), ("v6-small", 2, R.QUANT_NF4)]


def make(name, ql, qt):
    t = R.synth_named(name)
    ref = R.RwkvRef(t, quant_layers=ql, quant_type=qt)
    V = ref.info.num_vocab
    prompt = [tk % V for tk in R.synth_prompt(3, 24)]
    st = ref.init_state()
    logits = ref.forward(prompt, st)[-1]
    toks, st = ref.greedy(prompt, 16)
    return dict(name=name, quant_layers=ql, quant_type=qt, prompt=np.array(prompt), logits=logits.astype(np.float32),
                greedy=np.array(toks), state_sum=np.float64(st.astype(np.float64).sum()),
                state_abs=np.float64(np.abs(st).astype(np.float64).sum()))


if __name__ == "__main__":
    out = os.path.dirname(os.path.abspath(__file__))
    for name, ql, qt in CASES:
        d = make(name, ql, qt)
        fn = os.path.join(out, f"{name}-q{qt}.npz")
        np.savez_compressed(fn, **d)
        print("wrote", fn)
