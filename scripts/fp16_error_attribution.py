"""Which operand class carries a model's `Precision::Fp16` error at full depth?  (Next #1b.)

CPU only: the compiled restatement (oracle/cpu_backend.c) with its operand-rounding switch — bit i set rounds the GEMM operands of class i
to fp16 on the way in, which is exactly what the GPU's Fp16 mode does (f16 operands, fp32 accumulate, reload.rs:89-94); everything else
stays fp32.  For each model: the exact run (mask 0), all classes rounded (= the GPU's Fp16 mode), each class alone, and all but each
class (= what promoting THAT class to hi + lo operands would leave).  Same workload as tests/test_gpu_full_depth.py: 32 slots, ragged
prefill, 12 teacher-forced decode steps; logits of every step and the final state slab.

  python scripts/fp16_error_attribution.py v7-2.9b nf4 > profiles/r5_fp16_error_attribution_sim_v7-2.9b_nf4.jsonl
"""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import rwkv_ref as R  # noqa: E402
from oracle.cpu_backend import CpuBackend  # noqa: E402


def run(cpu, prompts, ids, mask, B, n_steps):
    cpu.set_operand_rounding(mask)
    states = cpu.init_states(B)
    out = []
    for s in range(max(len(p) for p in prompts)):
        act = [b for b in range(B) if len(prompts[b]) > s]
        sub = np.ascontiguousarray(states[act])
        need = any(len(prompts[b]) == s + 1 for b in act)
        lg = cpu.step([prompts[b][s] for b in act], sub, want_logits=need)
        states[act] = sub
        if need:
            out.append((act, lg))
    last = np.zeros((B, cpu.info.num_vocab), np.float32)
    for s, (act, lg) in enumerate(out):
        for i, b in enumerate(act):
            last[b] = lg[i]
    lgs = [last]
    cur = ids[0] if ids is not None else [int(t) for t in np.argmax(last, axis=1)]
    new_ids = [list(cur)]
    for s in range(n_steps):
        lg = cpu.step(cur, states)
        lgs.append(lg)
        cur = ids[s + 1] if ids is not None else [int(t) for t in np.argmax(lg, axis=1)]
        new_ids.append(list(cur))
    cpu.set_operand_rounding(0)
    return np.stack(lgs), states, new_ids


def main():
    name = sys.argv[1] if len(sys.argv) > 1 else "v7-2.9b"
    qt = {"none": R.QUANT_NONE, "int8": R.QUANT_INT8, "nf4": R.QUANT_NF4}[sys.argv[2] if len(sys.argv) > 2 else "nf4"]
    B, n_steps = 32, 12
    st, tens = R.synth_st(name, fast=True)
    info = R.model_info(tens)
    cpu = CpuBackend(tens, info.num_layer, qt)
    V = info.num_vocab
    prompts = [[t % V for t in R.synth_prompt(900 + b, [5, 3, 6, 2, 4][b % 5])] for b in range(B)]
    t0 = time.time()
    ref_lg, ref_st, ids = run(cpu, prompts, None, 0, B, n_steps)
    print(f"# exact run {time.time() - t0:.1f} s", file=sys.stderr)
    classes = list(CpuBackend.OPERAND_CLASSES)
    used = [i for i, c in enumerate(classes) if (info.version == 7 and c in ("att", "lora1", "lora2", "wo", "ffn1", "fv", "head")) or
            (info.version == 6 and c not in ("lora1", "lora2")) or (info.version == 5 and c in ("att", "wo", "ffn1", "fv", "head"))]
    full = sum(1 << i for i in used)
    cases = [("all (= Precision::Fp16)", full)] + [(f"only {classes[i]}", 1 << i) for i in used] + \
            [(f"all but {classes[i]}", full & ~(1 << i)) for i in used]
    if len(sys.argv) > 3 and sys.argv[3] == "att-split" and info.version == 6:
        # round 6: which of V6's five time-mix projections (bits 10..14: r, k, v, g, decay LoRA stage 1) must read hi + lo operands?  `rest` = every
        # other class rounded (what Precision::Fp16 leaves f16 anyway); a listed projection is ROUNDED (not promoted), the others are exact
        rest = full & ~1
        names = {"r": 10, "k": 11, "v": 12, "g": 13, "w": 14}
        import itertools
        cases = [("all (= raw f16)", full), ("att promoted (= Precision::Fp16 today)", rest)]
        for n in (1, 2, 3):
            for sub in itertools.combinations("rkvgw", n):
                cases.append(("rounded: " + "+".join(sub) + " (promoted: " + "+".join(c for c in "rkvgw" if c not in sub) + ")", rest | sum(1 << names[c] for c in sub)))
    if info.version == 7:
        cases += [("all but lora1+lora2", full & ~0b110), ("all but lora2+wo", full & ~0b1100), ("all but att+lora1", full & ~0b11),
                  ("all but att+lora1+lora2", full & ~0b111)]
    for label, mask in cases:
        lg, stt, _ = run(cpu, prompts, ids, mask, B, n_steps)
        e_lg = float(np.abs(lg - ref_lg).max())
        e_st = float(np.abs(stt - ref_st).max())
        e_emb = float(np.abs(stt[:, -1, 1:-1] - ref_st[:, -1, 1:-1]).max())
        rec = {"model": name, "quant": sys.argv[2] if len(sys.argv) > 2 else "nf4", "case": label, "mask": mask,
               "logits_max_abs": e_lg, "logits_ref_inf": float(np.abs(ref_lg).max()), "logits_rel": e_lg / max(1.0, float(np.abs(ref_lg).max())),
               "state_max_abs": e_st, "state_ref_inf": float(np.abs(ref_st).max()), "state_rel": e_st / max(1.0, float(np.abs(ref_st).max())),
               "emb_last_layer_max_abs": e_emb, "emb_ref_inf": float(np.abs(ref_st[:, -1, 1:-1]).max())}
        print(json.dumps(rec), flush=True)


if __name__ == "__main__":
    main()
