"""GPU (-m gpu): parity at the FULL DEPTH of the configurations BASELINE's metric is quoted on.

lib.rs:465 quantises layers `0..quant`; configs #3 / #4 set quant = num_layer = 32, so the headline engines are 32 fake-quantised
layers deep.  The two-layer tests of test_gpu_bench_paths.py / test_gpu_embeddings.py cover every kernel at the right widths but say
nothing about error growth (and arg-max stability) over 32 quantised layers.  Here the real shapes run end to end:

  * V6-World-3B shapes, Int8 on all 32 layers, 32 slots (config #3): a ragged prefill, then 12 decode steps through `rwkv_infer`
    (logits of every slot, arg-max) and through `rwkv_decode_greedy` (ids).  The synthetic checkpoints have nearly flat logits over
    65 536 tokens, so top-2 gaps below the logits error occur: an arg-max that differs is accepted only when the REFERENCE's gap
    between the two candidates is within twice the measured error of that row (what |got - want| <= err permits), and counted;
  * V7-World-2.9B shapes, NF4 on all 32 layers, 32 slots (config #4's engine): the same;
  * 32 x 256-token documents prefilled with `RWKV_OPTION_NONE` at `token_chunk_size` 256 (config #4's job), the layer-31 slice
    (`rwkv_state_back_layer`, docs/doc-api/openai.md:376-437) and the whole slab compared; in `Precision::Fp32` the slice is held to
    north_star's ABSOLUTE 1e-3.

Reference: `oracle.cpu_backend.CpuBackend` — the compiled restatement, pinned to `RwkvRefBatch` by tests/test_oracle.py (a 32-slot
32-layer step is 0.25 s on the GPU box's 16 cores).  Every comparison prints the measured max-abs AND relative error (run with -s),
and appends them to gpurun_out/full_depth_errors.jsonl when that directory exists, so DESIGN.md quotes measured numbers.

Bounds.  `Precision::Fp16` — the engine's DEFAULT mode and the one bench.py quotes (ABI 7: f16 operands, the launches that carry
a model's operand-rounding error read hi + lo) — logits and state max-abs <= 1e-3 * max(1, |ref|_inf) at scale 1.0 for every model;
`Precision::Fp32` — absolute 1e-3 on logits, embedding slice and state; `Precision::Fp16Raw` (f16 operands on every launch, an explicitly
named opt-in) — the same relative bound for V6, 3e-3 for V7-2.9B NF4 (measured 1.9e-3: the reason the raw mode is not the default)."""
import json
import os

import numpy as np
import pytest

from ai00_server_amd import runtime as rt
from oracle import rwkv_ref as R

pytestmark = pytest.mark.gpu
FP16_TOL = 1e-3
ABS_TOL = 1e-3
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def report(what, got, want, bound):
    err = float(np.abs(np.asarray(got, np.float64) - np.asarray(want, np.float64)).max())
    mag = float(np.abs(want).max())
    rec = {"case": what, "max_abs_err": err, "ref_inf_norm": mag, "rel_err": err / max(mag, 1e-30), "bound": bound}
    print(f"[full-depth] {what}: max-abs {err:.3e}, |ref|inf {mag:.3f}, relative {rec['rel_err']:.3e}, bound {bound:.3e}")
    out = os.path.join(ROOT, "gpurun_out")
    if os.path.isdir(out):
        with open(os.path.join(out, "full_depth_errors.jsonl"), "a") as f:
            f.write(json.dumps(rec) + "\n")
    return err


def log_record(rec):
    out = os.path.join(ROOT, "gpurun_out")
    if os.path.isdir(out):
        with open(os.path.join(out, "full_depth_errors.jsonl"), "a") as f:
            f.write(json.dumps(rec) + "\n")


def rel_bound(want):
    return FP16_TOL * max(1.0, float(np.abs(want).max()))


class Model:
    def __init__(self, name, qt):
        from oracle.cpu_backend import CpuBackend
        self.st, self.tens = R.synth_st(name, fast=True)
        self.info = R.model_info(self.tens)
        self.qt = qt
        self.cpu = CpuBackend(self.tens, self.info.num_layer, qt)

    def engine(self, B, chunk, prec=rt.Precision.Fp16):
        return rt.ModelBuilder(self.st).quant(self.info.num_layer, rt.Quant(self.qt)).build(max_batch=B, token_chunk_size=chunk, precision=prec)

    def cpu_prefill(self, prompts, states, want_logits=True):
        """Ragged prompts through the lock-step CPU step: at step s the slots that still have a token advance together.
        Returns the logits after each slot's last token."""
        B = len(prompts)
        last = [None] * B
        for s in range(max(len(p) for p in prompts)):
            act = [b for b in range(B) if len(prompts[b]) > s]
            sub = np.ascontiguousarray(states[act])
            need = want_logits and any(len(prompts[b]) == s + 1 for b in act)
            lg = self.cpu.step([prompts[b][s] for b in act], sub, want_logits=need)
            states[act] = sub
            if need:
                for i, b in enumerate(act):
                    if len(prompts[b]) == s + 1:
                        last[b] = lg[i].copy()
        return last


@pytest.fixture(scope="module")
def v6_int8():
    return Model("v6-3b", R.QUANT_INT8)


@pytest.fixture(scope="module")
def v7_nf4():
    return Model("v7-2.9b", R.QUANT_NF4)


def feed(eng, prompts, option=rt.RnnOption.Last):
    B = eng.max_batch
    inp = rt.RnnInput([rt.RnnInputBatch(list(prompts[b]) if b < len(prompts) else [], option) for b in range(B)])
    rows = [[] for _ in range(B)]
    while inp.num_token() > 0:
        inp, outs = eng.infer(inp)
        for b, o in enumerate(outs):
            rows[b].extend(list(o))
    return rows


def decode_case(m, tag, n_steps=12, B=32, prec=rt.Precision.Fp16, tol_scale=1.0, abs_bound=None):
    """`tol_scale` is 1.0 in the default mode and in Fp32; only the explicitly named raw-f16 case of V7 states a wider bound.
    Precision::Fp32 is held to north_star's absolute 1e-3 whatever the magnitude."""
    V = m.info.num_vocab
    eng = m.engine(B, 256, prec)

    def rel_bound(want):                      # shadows the module-level rule for this case
        if abs_bound is not None:
            return abs_bound
        return ABS_TOL if prec == rt.Precision.Fp32 else tol_scale * FP16_TOL * max(1.0, float(np.abs(want).max()))

    prompts = [[t % V for t in R.synth_prompt(900 + b, [5, 3, 6, 2, 4][b % 5])] for b in range(B)]
    states = m.cpu.init_states(B)
    want = np.stack(m.cpu_prefill(prompts, states))
    rows = feed(eng, prompts)
    got = np.stack([rows[b][-1] for b in range(B)])
    assert report(f"{tag} prefill logits (32 slots, ragged)", got, want, rel_bound(want)) <= rel_bound(want)
    for b in np.nonzero(np.argmax(got, axis=1) != np.argmax(want, axis=1))[0]:      # near-tie rule, see below
        gap = float(want[b].max() - want[b, int(np.argmax(got[b]))])
        assert gap <= 2.0 * float(np.abs(got[b] - want[b]).max()), f"arg-max after the prefill, slot {b}: reference gap {gap:.3e}"
    # ---- rwkv_infer, teacher-forced with the reference's ids: logits of every slot at every step
    snaps = [eng.state.back(b) for b in range(B)]
    cur = [int(t) for t in np.argmax(want, axis=1)]
    first = list(cur)
    want_ids = np.zeros((n_steps, B), np.int64)
    ref_lg, step_err = [], []
    worst, flips, min_margin = 0.0, [], np.inf
    for s in range(n_steps):
        lg = m.cpu.step(cur, states)
        inp = rt.RnnInput([rt.RnnInputBatch([cur[b]], rt.RnnOption.Last) for b in range(B)])
        _, outs = eng.infer(inp)
        g = np.stack([outs[b][-1] for b in range(B)])
        err_b = np.abs(g - lg).max(axis=1)
        e = float(err_b.max())
        worst = max(worst, e / rel_bound(lg))
        assert e <= rel_bound(lg), f"step {s}: {e}"
        gi, wi = np.argmax(g, axis=1), np.argmax(lg, axis=1)
        top2 = np.partition(lg, -2, axis=1)[:, -2:]
        min_margin = min(min_margin, float((top2[:, 1] - top2[:, 0]).min()))
        for b in np.nonzero(gi != wi)[0]:
            # |got - want| <= err on both candidates, so a flip is only possible when the reference's own gap between the two is
            # within 2 * err: a near-tie of the synthetic model, not an engine error.  Anything else fails.
            gap = float(lg[b, wi[b]] - lg[b, gi[b]])
            assert gap <= 2.0 * float(err_b[b]), f"arg-max differs at step {s} slot {b} with a reference gap of {gap:.3e} (error {err_b[b]:.3e})"
            flips.append((s, int(b), gap, float(err_b[b])))
        cur = [int(t) for t in wi]
        want_ids[s] = cur
        ref_lg.append(lg)
        step_err.append(err_b)
    print(f"[full-depth] {tag} decode: worst logits error over {n_steps} steps = {worst:.3f} of the bound; smallest top-2 gap of the reference "
          f"{min_margin:.3e}; arg-max flips on near-ties: {[(s, b, f'gap {g:.1e} <= 2 x err {e:.1e}') for s, b, g, e in flips]}")
    assert len(flips) <= max(1, B * n_steps // 100), "too many near-tie flips to call the arg-max stable"
    log_record({"case": f"{tag} arg-max", "pairs": B * n_steps, "near_tie_flips": len(flips), "smallest_reference_top2_gap": min_margin,
                "flips": [{"step": s_, "slot": b_, "reference_gap": g_, "row_error": e_} for s_, b_, g_, e_ in flips]})
    back = np.stack([eng.state.back(b) for b in range(B)])
    assert report(f"{tag} state after prefill + {n_steps} decode steps", back, states, rel_bound(states)) <= rel_bound(states)
    # ---- rwkv_decode_greedy (the bench's timed call): ids stay on the device.  A slot is compared until its first near-tie flip (same
    # rule; after it the slot legitimately follows another trajectory); at most one slot in ten may leave this way.
    for b in range(B):
        eng.state.load(snaps[b], b)
    toks, _ = eng.decode_greedy(first, n_steps)
    toks = np.asarray(toks, dtype=np.int64)[:, :B]
    alive, left = set(range(B)), []
    for s in range(n_steps):
        for b in sorted(alive):
            if toks[s, b] != want_ids[s, b]:
                gap = float(ref_lg[s][b, want_ids[s, b]] - ref_lg[s][b, toks[s, b]])
                assert gap <= 2.0 * float(step_err[s][b]) + 1e-6, f"device greedy id differs at step {s} slot {b}, reference gap {gap:.3e}"
                alive.discard(b)
                left.append((s, b, gap))
    print(f"[full-depth] {tag} rwkv_decode_greedy: {len(alive)}/{B} slots identical over {n_steps} steps; left on a near-tie: {left}")
    log_record({"case": f"{tag} rwkv_decode_greedy ids", "slots_identical": len(alive), "slots": B, "steps": n_steps,
                "left_on_near_tie": [{"step": s_, "slot": b_, "reference_gap": g_} for s_, b_, g_ in left]})
    assert len(alive) >= B - max(1, B // 10)
    eng.close()


MODES = [rt.Precision.Fp16, rt.Precision.Fp32]


@pytest.mark.parametrize("prec", MODES, ids=[p.name for p in MODES])
def test_config3_v6_3b_int8_32_layers_32_slots(v6_int8, prec):
    """BASELINE config #3 as quoted: 32 Int8 layers x 32 slots (lib.rs:465), in both of the reference's `Precision`s (reload.rs:89-94,
    lib.rs:503-515).  Fp16 is the engine's default and the bench headline's mode: bound at scale 1.0."""
    decode_case(v6_int8, f"v6-3b int8 x32 layers Precision::{prec.name}", prec=prec)


@pytest.mark.parametrize("prec", MODES, ids=[p.name for p in MODES])
def test_config4_engine_v7_2p9b_nf4_32_layers_32_slots(v7_nf4, prec):
    """Config #4's engine: V7-2.9B shapes, NF4 on all 32 layers, 32 slots, in both `Precision`s at scale 1.0.  (Which operand class carries
    V7's f16 error was measured on the CPU restatement with its per-class rounding switch — scripts/fp16_error_attribution.py,
    profiles/r5_fp16_error_attribution_sim_v7-2.9b_nf4.jsonl: the r / k / v projections' inputs alone carry 4.7e-3 of 4.9e-3; with the
    time-mix launch, the second-stage LoRAs and the output projection reading hi + lo operands 9.7e-4 absolute is left.  That set IS
    Precision::Fp16 for V7 since ABI 7.)"""
    decode_case(v7_nf4, f"v7-2.9b nf4 x32 layers Precision::{prec.name}", prec=prec)


def test_raw_f16_mode_v6_3b_int8_32_layers(v6_int8):
    """RWKV_PRECISION_FP16_RAW (explicit opt-in, not the default): f16 operands on every launch.  V6-3B Int8 stays inside the relative bound
    (measured 1.35e-3 absolute = 5.4e-4 of |ref|inf)."""
    decode_case(v6_int8, "v6-3b int8 x32 layers raw f16 (Precision::Fp16Raw)", prec=rt.Precision.Fp16Raw)


def test_raw_f16_mode_v7_2p9b_nf4_32_layers_states_its_wider_bound(v7_nf4):
    """The raw mode on V7-2.9B NF4 measures 1.9e-3 of |ref|inf (4.7e-3 absolute) at 32 layers — outside north_star's 1e-3, which is why it is
    NOT the default.  Its bound here is 3e-3 relative, stated for this named mode only."""
    decode_case(v7_nf4, "v7-2.9b nf4 x32 layers raw f16 (Precision::Fp16Raw)", prec=rt.Precision.Fp16Raw, tol_scale=3.0)


@pytest.mark.parametrize("which", ["v7_nf4", "v6_int8"])
def test_embeddings_job_at_full_depth(which, request):
    """32 documents x 256 tokens, state-only (`RWKV_OPTION_NONE`), `token_chunk_size` 256 as SURVEY 8(d) names for config #4: the
    layer-31 embedding slice and the whole slab against the CPU restatement, in both precisions (+ V7's raw-f16 mode under its own name)."""
    m = request.getfixturevalue(which)
    V, L, B, T = m.info.num_vocab, m.info.num_layer, 32, 256
    docs = [[t % V for t in R.synth_prompt(1200 + b, T)] for b in range(B)]
    states = m.cpu.init_states(B)
    m.cpu_prefill(docs, states, want_logits=False)
    # public slab [L][N+2][C]: rows 1..N of a layer are its WKV matrix = the embedding of docs/doc-api/openai.md:376-437
    want_emb = states[:, L - 1, 1:-1, :]
    for prec in (rt.Precision.Fp16, rt.Precision.Fp32) + ((rt.Precision.Fp16Raw,) if m.info.version == 7 else ()):
        name = prec.name
        eng = m.engine(B, 256, prec)
        feed(eng, docs, rt.RnnOption.NoOutput)
        emb = np.stack([eng.state.embed(L - 1, b).reshape(want_emb.shape[1:]) for b in range(B)])
        back = np.stack([eng.state.back(b) for b in range(B)])
        eng.close()
        if prec == rt.Precision.Fp32:
            assert report(f"{which} embeddings (layer {L - 1}) Precision::{name}", emb, want_emb, ABS_TOL) <= ABS_TOL
            assert report(f"{which} state slab Precision::{name}", back, states, ABS_TOL) <= ABS_TOL
        else:
            # scale 1.0 in the default mode; only V7's explicitly named raw mode (measured 1.3e-3 of |ref|inf on the embedding) states 3e-3
            k = 3.0 if prec == rt.Precision.Fp16Raw else 1.0
            assert report(f"{which} embeddings (layer {L - 1}) Precision::{name}", emb, want_emb, k * rel_bound(want_emb)) <= k * rel_bound(want_emb)
            assert report(f"{which} state slab Precision::{name}", back, states, k * rel_bound(states)) <= k * rel_bound(states)
