"""CPU: the oracle against its golden fixtures, against an independent literal transcription of the published
BlinkDL functions, and its own invariants (chunking, state round trip, quantisation bounds)."""
import glob
import os

import numpy as np
import pytest

from oracle import rwkv_ref as R
from tests.blinkdl_literal import Literal

GOLD = sorted(glob.glob(os.path.join(os.path.dirname(__file__), "golden", "*.npz")))


def _prompt(ref, slot, n):
    return [t % ref.info.num_vocab for t in R.synth_prompt(slot, n)]


@pytest.mark.parametrize("path", GOLD, ids=[os.path.basename(p) for p in GOLD])
def test_oracle_matches_golden(path):
    g = np.load(path)
    ref = R.RwkvRef(R.synth_named(str(g["name"])), int(g["quant_layers"]), int(g["quant_type"]))
    st = ref.init_state()
    logits = ref.forward(list(g["prompt"]), st)[-1]
    np.testing.assert_allclose(logits, g["logits"], rtol=0, atol=2e-5)
    toks, st = ref.greedy(list(g["prompt"]), 16)
    assert toks == list(g["greedy"])
    assert abs(st.astype(np.float64).sum() - float(g["state_sum"])) <= 1e-3 * max(1.0, float(g["state_abs"])) * 1e-2


@pytest.mark.parametrize("name", ["v5-tiny", "v6-tiny", "v7-tiny"])
def test_oracle_matches_literal_blinkdl(name):
    t = R.synth_named(name)
    ref, lit = R.RwkvRef(t), Literal(t)
    st, ls = ref.init_state(), lit.new_state()
    for tok in _prompt(ref, 1, 12):
        a = ref.forward([tok], st)[-1]
        b = lit.forward(tok, ls)
        np.testing.assert_allclose(a, b, rtol=0, atol=5e-5)
    # state conventions: slab[l][1+i][h*N+j] = S_h[i][j] in BlinkDL's own [H,N,N] indexing
    N, H = ref.info.head_size, ref.info.num_head
    for l in range(ref.info.num_layer):
        S = st[l, 1:1 + N].reshape(N, H, N).transpose(1, 0, 2)
        np.testing.assert_allclose(S, ls[l][1].numpy(), rtol=0, atol=5e-5)
        np.testing.assert_allclose(st[l, 0], ls[l][0].numpy(), atol=5e-5)
        np.testing.assert_allclose(st[l, N + 1], ls[l][2].numpy(), atol=5e-5)


@pytest.mark.parametrize("name", ["v5-tiny", "v6-tiny", "v7-tiny"])
def test_chunking_is_exact_and_state_roundtrips(name):
    ref = R.RwkvRef(R.synth_named(name))
    p = _prompt(ref, 0, 30)
    s1 = ref.init_state()
    full = ref.forward(p, s1, full=True)
    s2 = ref.init_state()
    a = ref.forward(p[:7], s2, full=True)
    saved = s2.copy()                                  # "back" ...
    s3 = saved.copy()                                  # ... "load" into another slot
    b = ref.forward(p[7:], s3, full=True)
    np.testing.assert_array_equal(np.concatenate([a, b]), full)
    np.testing.assert_array_equal(s1, s3)
    assert full.shape == (30, ref.info.num_vocab) and np.isfinite(full).all()


def test_long_run_stays_finite():
    ref = R.RwkvRef(R.synth_named("v6-tiny"))
    toks, st = ref.greedy(_prompt(ref, 5, 4), 300)
    assert np.isfinite(st).all() and np.abs(st).max() < 1e4 and len(toks) == 300


def test_quant_reference_bounds_and_layout():
    rng = np.random.default_rng(0)
    w = (rng.standard_normal((32, 512)) * 0.05).astype(np.float16)
    q, a, b = R.quant_int8(w)
    d = R.dequant_int8(q, a, b).astype(np.float32)
    step = a.astype(np.float32).repeat(128, axis=1)
    assert np.all(np.abs(d - w.astype(np.float32)) <= 0.51 * step + 2e-3 * np.abs(w.astype(np.float32)) + 1e-6)
    idx, am = R.quant_nf4(w)
    d4 = R.dequant_nf4(idx, am).astype(np.float32)
    assert idx.max() <= 15 and np.all(np.abs(d4) <= am.astype(np.float32).repeat(64, axis=1) * 1.001)
    # the block absmax element decodes exactly to +-absmax
    blk = w.reshape(32, 8, 64).astype(np.float32)
    pos = np.abs(blk).argmax(axis=2)
    dd = d4.reshape(32, 8, 64)
    np.testing.assert_array_equal(np.abs(np.take_along_axis(dd, pos[..., None], 2))[..., 0], am.astype(np.float32))
    # only the big projection matrices of layers < quant are quantised (lib.rs:465)
    t = R.synth_named("v6-small")
    r0, r1 = R.RwkvRef(t), R.RwkvRef(t, 1, R.QUANT_INT8)
    assert not np.array_equal(r0.w["blocks.0.att.key.weight"], r1.w["blocks.0.att.key.weight"])
    np.testing.assert_array_equal(r0.w["blocks.1.att.key.weight"], r1.w["blocks.1.att.key.weight"])
    np.testing.assert_array_equal(r0.w["head.weight"], r1.w["head.weight"])
    np.testing.assert_array_equal(r0.w["blocks.0.att.time_mix_w1"], r1.w["blocks.0.att.time_mix_w1"])


def test_mix_convention_trap_v5_vs_v6():
    """Both versions store `time_mix_k`, with opposite meaning: mu_v6 == 1 - mu_v5."""
    t5 = R.synth_named("v5-tiny")
    ref = R.RwkvRef(t5)
    xx, sx = np.ones(128, np.float32), np.zeros(128, np.float32)
    mu = ref.w["blocks.0.ffn.time_mix_k"].reshape(-1)
    v5 = xx * mu + sx * (1 - mu)
    v6 = xx + (sx - xx) * mu
    np.testing.assert_allclose(v5, 1 - v6, atol=1e-6)


def test_safetensors_roundtrip_and_info():
    t = R.synth_named("v7-tiny")
    data = R.st_serialize(t, metadata={"format": "pt"})
    back = R.st_deserialize(data)
    assert set(back) == set(t) and all(np.array_equal(back[k], t[k]) for k in t)
    i = R.model_info(back)
    assert (i.version, i.num_layer, i.num_emb, i.num_hidden, i.num_vocab, i.num_head) == (7, 2, 128, 512, 512, 2)
    buf, views = R.synth_st("v6-tiny", fast=False)
    assert all(np.array_equal(views[k], R.synth_named("v6-tiny")[k]) for k in views)


def test_init_state_perplexity_softmax_helpers():
    ref = R.RwkvRef(R.synth_named("v6-tiny"))
    st = R.synth_init_state(ref.info)
    slab = ref.read_init_state(st)
    N, H = ref.info.head_size, ref.info.num_head
    ts = np.asarray(st["blocks.1.att.time_state"], np.float32).transpose(0, 2, 1)     # [H, i, j]
    assert slab[1, 1 + 5, 1 * N + 9] == ts[1, 5, 9] and not slab[:, 0].any() and not slab[:, N + 1].any()
    p = _prompt(ref, 2, 6)
    s = ref.init_state()
    rows = ref.forward([0] + p, s, full=True)
    ppl = R.perplexity_ref(rows, p)
    sm = R.softmax_ref(rows)
    want = -np.mean([np.log(sm[i, p[i]]) for i in range(len(p))] ) * len(p) / (len(p) + 1)
    assert abs(ppl - want) < 1e-4
    np.testing.assert_allclose(sm.sum(axis=1), 1.0, atol=1e-5)


def test_algorithmic_bytes_match_survey_table():
    for name, wq_fp16 in [("v6-3b", 5.865e9), ("v6-1.6b", 2.932e9), ("v5-0.4b", 0.790e9), ("v6-7b", 14.736e9)]:
        cfg = R.CONFIGS[name]
        shapes = R.synth_checkpoint(*cfg, shapes_only=True)
        info = R.ModelInfo(cfg[0], cfg[1], cfg[2], cfg[3], cfg[4], cfg[2] // 64)
        ab = R.algorithmic_bytes(info, shapes, 0, R.QUANT_NONE, 1)
        assert abs(ab["W_q"] - wq_fp16) / wq_fp16 < 0.01, (name, ab["W_q"])
    cfg = R.CONFIGS["v6-3b"]
    shapes = R.synth_checkpoint(*cfg, shapes_only=True)
    info = R.ModelInfo(6, 32, 2560, 8960, 65536, 40)
    ab = R.algorithmic_bytes(info, shapes, 32, R.QUANT_INT8, 32)
    assert abs(ab["W_q"] - 3.224e9) / 3.224e9 < 0.01 and ab["S"] == 21626880


def test_nucleus_reference_semantics():
    """Restatement of sampler/nucleus.rs: first element always kept, top_p compared BEFORE adding, find_or_first."""
    p = np.array([0.5, 0.3, 0.1, 0.06, 0.04], np.float32)
    assert R.nucleus_ref(p, 0.0, 128, 1.0, 0.99)[0] == 0                 # top_p = 0 keeps exactly the first element
    assert R.nucleus_ref(p, 0.5, 128, 1.0, 0.99)[0] == 1                 # cum(before 2nd) = 0.5 <= 0.5 -> 2 kept
    assert R.nucleus_ref(p, 0.5, 128, 1.0, 0.50)[0] == 0                 # u <= 0.5/0.8
    assert R.nucleus_ref(p, 1.0, 2, 1.0, 0.99)[0] == 1                   # top_k cut
    assert R.nucleus_ref(p, 1.0, 128, 1.0, 1.5)[0] == 0                  # nothing found -> FIRST (find_or_first)
    hot = R.nucleus_ref(p, 1.0, 128, 0.1, 0.93)[0]                       # low temperature sharpens
    assert hot == 0
    s = R.NucleusRef()
    s.init([7, 8, 7])
    assert abs(s.penalties[7] - (0.3 + 0.3 + 0.3 * 0.99654026 ** 2)) < 1e-6 and abs(s.penalties[8] - (0.3 + 0.3 * 0.99654026)) < 1e-6
    lg = np.zeros(16, np.float32)
    assert s.transform(lg)[7] == -s.penalties[7]
    s.update(8)
    assert abs(s.penalties[8] - ((0.3 + 0.3 * 0.99654026) * 0.99654026 + 0.3)) < 1e-6
    s.update(3)
    assert abs(s.penalties[3] - 0.3) < 1e-7


def test_typical_ref_matches_a_literal_restatement():
    """typical_ref against a from-the-text restatement of sampler/typical.rs:70-120 (sort of (|y - H|, id) pairs)."""
    rng = np.random.default_rng(9)
    for trial in range(20):
        x = rng.standard_normal(300).astype(np.float32) * 3
        p = R.softmax_ref(x[None])[0]
        tau, top_k, temp, u = float(rng.uniform(0.2, 0.95)), int(rng.integers(1, 64)), float(rng.uniform(0.5, 1.5)), float(rng.random())
        items = [(i, float(v), -float(np.log(np.float32(v)))) for i, v in enumerate(p) if v > 0]
        h = np.float32(0)
        for _, a, b in items:
            h = np.float32(h + np.float32(np.float32(a) * np.float32(b)))
        srt = sorted(((np.float32(abs(np.float32(b) - h)), i, a) for i, a, b in items))[:top_k]
        kept, cum = [], np.float32(0)
        for _, i, a in srt:
            if cum > np.float32(tau):
                break
            cum = np.float32(cum + np.float32(a))
            kept.append((i, np.float32(a) ** np.float32(1.0 / temp)))
        tot = np.float32(0)
        for _, q in kept:
            tot = np.float32(tot + q)
        c, want = np.float32(0), kept[0][0]
        for i, q in kept:
            c = np.float32(c + np.float32(q / tot))
            if np.float32(u) <= c:
                want = i
                break
        got, _ = R.typical_ref(p, tau, top_k, temp, u)
        assert got == want


def test_mirostat_ref_matches_a_literal_restatement():
    """mirostat_ref against a from-the-text restatement of sampler/mirostat.rs:44-84 (sort, running sum, truncate at the
    first token whose surprise exceeds max_surprise, draw u * sum), and the surprise it reports."""
    rng = np.random.default_rng(10)
    for trial in range(20):
        x = rng.standard_normal(400).astype(np.float32) * 3
        p = R.softmax_ref(x[None])[0]
        ms, u = float(rng.uniform(2.0, 9.0)), float(rng.random())
        srt = sorted(((-float(v), i) for i, v in enumerate(p)))                  # descending by p, ties by id
        cum, c = [], np.float32(0)
        for nv, i in srt:
            c = np.float32(c + np.float32(-nv))
            cum.append((i, c, np.float32(-nv)))
        k = len(cum)
        for pos, (_, _, v) in enumerate(cum):
            if -np.log2(v) > np.float32(ms):
                k = pos + 1
                break
        cum = cum[:k]
        total = cum[-1][1]
        r = np.float32(np.float32(u) * total)
        tok, prob = cum[0][0], cum[0][2]
        for i, cc, v in cum:
            if r <= cc:
                tok, prob = i, v
                break
        got_tok, got_surprise, _ = R.mirostat_ref(p, ms, u)
        assert got_tok == tok
        assert abs(got_surprise - float(np.log2(total) - np.log2(prob))) < 1e-6


@pytest.mark.parametrize("name,quant", [("v5-small", (0, 0)), ("v6-small", (2, 1)), ("v7-small", (2, 2)), ("v6-tiny", (0, 0))])
def test_lockstep_batch_form_agrees_with_the_per_token_restatement(name, quant):
    """RwkvRefBatch (the multi-slot form the 32-slot GPU tests check against) == RwkvRef slot by slot: logits and state to
    fp32 round-off, greedy ids identical."""
    t = R.synth_named(name)
    ref, rb = R.RwkvRef(t, *quant), R.RwkvRefBatch(t, *quant)
    V = ref.info.num_vocab
    ps = [[x % V for x in R.synth_prompt(s, 4 + 3 * s)] for s in range(5)]
    states = rb.init_states(5)
    lg = rb.prefill(ps, states)
    ids, _ = rb.greedy_batch(np.argmax(lg, axis=1), 12, states)
    for b in range(5):
        s = ref.init_state()
        want = ref.forward(ps[b], s)[-1]
        assert np.abs(want - lg[b]).max() <= 5e-6 * max(1.0, float(np.abs(want).max()))
        g, s2 = ref.greedy(ps[b], 13)
        assert g[0] == int(np.argmax(lg[b])) and g[1:] == [int(x) for x in ids[:, b]]


@pytest.mark.parametrize("name,quant", [("v5-tiny", (0, 0)), ("v6-tiny", (0, 0)), ("v7-tiny", (0, 0)), ("v5-small", (2, 1)), ("v6-small", (3, 1)),
                                        ("v6-small", (2, 2)), ("v7-small", (3, 2)), ("v7-small", (2, 1))])
def test_compiled_restatement_agrees_with_the_numpy_one(name, quant):
    """oracle/cpu_backend.c (C + OpenMP, the CPU baseline of bench.py) against RwkvRefBatch: two independent codes of the same
    formulas, the same fp16 / fake-quantised weights — logits and state slabs to fp32 round-off, arg-max identical, over 16 lock-step
    steps of five slots."""
    from oracle.cpu_backend import CpuBackend
    t = R.synth_named(name)
    rb, cb = R.RwkvRefBatch(t, *quant), CpuBackend(t, *quant)
    B, V = 5, rb.info.num_vocab
    s1, s2 = rb.init_states(B), cb.init_states(B)
    rng = np.random.default_rng(3)
    for step in range(16):
        toks = [int(x) for x in rng.integers(1, V, B)]
        a, b = rb.step(toks, s1), cb.step(toks, s2)
        assert np.abs(a - b).max() <= 1e-5 * max(1.0, float(np.abs(a).max())), step
        assert (np.argmax(a, axis=1) == np.argmax(b, axis=1)).all()
        assert np.abs(s1 - s2).max() <= 2e-5 * max(1.0, float(np.abs(s1).max()))
    assert cb.step([1] * B, s2, want_logits=False) is None


@pytest.mark.parametrize("name", ["v6-small", "v7-small"])
def test_operand_rounding_switch_of_the_compiled_restatement(name):
    """`CpuBackend.set_operand_rounding` (the instrument behind profiles/r5_fp16_error_attribution_sim_*.jsonl and RWKV_PROMOTE): mask 0 IS
    the oracle (bit for bit what every parity test uses); a set bit rounds that class's GEMM operands to fp16 and nothing else — the result
    moves by f16 operand noise (well above fp32 round-off, well below 1e-2), a class the model version does not have moves nothing, and
    switching the mask off again restores the exact bits."""
    from oracle.cpu_backend import CpuBackend
    t = R.synth_named(name)
    cb = CpuBackend(t, 0, 0)
    B, V = 4, cb.info.num_vocab
    toks = [[int(x) for x in np.random.default_rng(5 + s).integers(1, V, B)] for s in range(6)]

    def run(mask):
        cb.set_operand_rounding(mask)
        st = cb.init_states(B)
        lg = None
        for tk in toks:
            lg = cb.step(tk, st)
        cb.set_operand_rounding(0)
        return lg, st

    cls = list(CpuBackend.OPERAND_CLASSES)
    base_l, base_s = run(0)
    again_l, again_s = run(0)
    assert np.array_equal(base_l, again_l) and np.array_equal(base_s, again_s)
    all_l, _ = run((1 << len(cls)) - 1)
    err = float(np.abs(all_l - base_l).max())
    assert 1e-6 < err < 1e-2, err
    att_l, _ = run(1 << cls.index("att"))
    assert 1e-7 < float(np.abs(att_l - base_l).max()) <= 3 * err
    absent = "lora2" if cb.info.version != 7 else "mix1"                  # V6 has no second-stage LoRAs, V7 no token-shift LoRA
    off_l, off_s = run(1 << cls.index(absent))
    assert np.array_equal(off_l, base_l) and np.array_equal(off_s, base_s)


def test_compiled_fake_quantisation_is_bit_identical_to_the_numpy_one():
    """`rwkv_cpu_fake_quant_int8` (per 128-block a, b in fp16, one rounding of a*q + b from float64) and `rwkv_cpu_fake_quant_nf4`
    (per 64-block absmax, 15 fp32 midpoints, one rounding of absmax * table[idx]) against rwkv_ref.fake_quant over seven orders of
    magnitude, constant and zero blocks included."""
    import ctypes as C
    from oracle import cpu_backend as cb
    lib = C.CDLL(cb.build())
    u16p, f32p = C.POINTER(C.c_uint16), C.POINTER(C.c_float)
    lib.rwkv_cpu_fake_quant_int8.argtypes = [u16p, C.c_long, C.c_long]
    lib.rwkv_cpu_fake_quant_nf4.argtypes = [u16p, C.c_long, C.c_long, f32p, u16p]
    mid = np.ascontiguousarray(R.NF4_MID, dtype=np.float32)
    tab = np.ascontiguousarray(R.NF4_TABLE_F16, dtype=np.float16)
    rng = np.random.default_rng(0)
    for scale in (1e-6, 1e-3, 0.05, 1.0, 30.0, 3000.0):
        w = (rng.standard_normal((32, 512)) * scale).astype(np.float16)
        w[0, :128] = np.float16(0.37)
        w[1, :128] = 0
        got = w.copy()
        lib.rwkv_cpu_fake_quant_int8(got.view(np.uint16).ctypes.data_as(u16p), 32, 512)
        assert np.array_equal(got.view(np.uint16), R.fake_quant(w.copy(), R.QUANT_INT8).view(np.uint16)), scale
        got = w.copy()
        lib.rwkv_cpu_fake_quant_nf4(got.view(np.uint16).ctypes.data_as(u16p), 32, 512, mid.ctypes.data_as(f32p), tab.view(np.uint16).ctypes.data_as(u16p))
        assert np.array_equal(got.view(np.uint16), R.fake_quant(w.copy(), R.QUANT_NF4).view(np.uint16)), scale


def test_lora_argument_equals_blending_the_tensors_by_hand():
    """`RwkvRef(..., lora=[(file, alpha)])` = the checkpoint with W += alpha B A^T on matrices (fp16 result) and
    v = alpha l + (1 - alpha) v on every other `blocks.N.*` tensor the LoRA file names; tensors outside `blocks.N.*` are out of
    `LoraBlend::full`'s pattern and stay as they are (lib.rs:466-482; web-rwkv's loader restated, unpinned — see RwkvRef.__init__)."""
    t = R.synth_named("v6-tiny")
    rng = np.random.default_rng(9)
    C, r, alpha = 128, 4, 0.75
    lora = {"blocks.0.att.key.lora.0": (rng.standard_normal((C, r)) * 0.05).astype(np.float16),
            "blocks.0.att.key.lora.1": (rng.standard_normal((C, r)) * 0.05).astype(np.float16),
            "blocks.1.att.time_mix_k": (rng.standard_normal((1, 1, C)) * 0.1).astype(np.float16),
            "blocks.0.ln2.weight": (rng.standard_normal(C) * 0.1).astype(np.float16),
            "blocks.1.att.time_decay": (rng.standard_normal((1, 1, C)) * 0.1).astype(np.float16),
            "ln_out.weight": (rng.standard_normal(C) * 0.1).astype(np.float16),                                  # outside the pattern
            "head.lora.0": (rng.standard_normal((C, r)) * 0.05).astype(np.float16),
            "head.lora.1": (rng.standard_normal((t["head.weight"].shape[0], r)) * 0.05).astype(np.float16)}
    a = R.RwkvRef(t, lora=[(lora, alpha)])
    k = "blocks.0.att.key.weight"
    want = (t[k].astype(np.float32) + np.float32(alpha) * (lora["blocks.0.att.key.lora.1"].astype(np.float32) @ lora["blocks.0.att.key.lora.0"].astype(np.float32).T)).astype(np.float16)
    np.testing.assert_array_equal(a.w[k], want.astype(np.float32))
    for name in ("blocks.1.att.time_mix_k", "blocks.0.ln2.weight", "blocks.1.att.time_decay"):
        np.testing.assert_array_equal(a.w[name], np.float32(alpha) * lora[name].astype(np.float32).reshape(t[name].shape)
                                      + (np.float32(1.0) - np.float32(alpha)) * t[name].astype(np.float32))
    for name in ("blocks.1.ln2.weight", "ln_out.weight", "head.weight"):                                # untouched
        assert np.array_equal(a.w[name], t[name].astype(np.float32)), name
    p = _prompt(a, 2, 9)
    base = R.RwkvRef(t)
    s0, s1 = base.init_state(), a.init_state()
    assert np.abs(base.forward(p, s0)[-1] - a.forward(p, s1)[-1]).max() > 1e-4                          # and it changes the answer
