"""Quick GPU probe: HIP path vs oracle on tiny models, printed (not asserted). Dev tool."""
import sys, os, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import rwkv_ref as R
from ai00_server_amd import runtime as rt

def run(name, precision, quant=(0, 0), T=9, B=2, chunk=16):
    t = R.synth_named(name)
    st = R.st_serialize(t)
    ref = R.RwkvRef(t, quant_layers=quant[0], quant_type=quant[1])
    V = ref.info.num_vocab
    eng = rt.ModelBuilder(st).quant(quant[0], rt.Quant(quant[1])).build(max_batch=B, token_chunk_size=chunk, precision=precision)
    prompts = [[tk % V for tk in R.synth_prompt(s, T + s)] for s in range(B)]
    # prefill (chunked) + Last
    inp = rt.RnnInput([rt.RnnInputBatch(list(p), rt.RnnOption.Last) for p in prompts])
    got = [None] * B
    while inp.num_token() > 0:
        inp, outs = eng.infer(inp)
        for b, o in enumerate(outs):
            if len(o): got[b] = o[-1]
    for b in range(B):
        s = ref.init_state()
        want = ref.forward(prompts[b], s)[-1]
        back = eng.state.back(b)
        print(f"{name} prec={precision.name} q={quant} slot{b}: logits maxabs={np.abs(got[b]-want).max():.3e} (|l|max {np.abs(want).max():.2f}) "
              f"state maxabs={np.abs(back-s).max():.3e} argmax {int(got[b].argmax())}/{int(want.argmax())}")
    eng.close()

if __name__ == "__main__":
    print(rt.list_adapters())
    for name in ["v6-tiny", "v5-tiny", "v7-tiny"]:
        for prec in [rt.Precision.Fp32, rt.Precision.Fp16]:
            try:
                run(name, prec)
            except Exception as e:
                print(name, prec, "FAILED:", repr(e))
    for q in [(2, 1), (2, 2)]:
        try:
            run("v6-small", rt.Precision.Fp32, quant=q)
        except Exception as e:
            print("quant", q, "FAILED:", repr(e))
