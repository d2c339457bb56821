"""GPU (-m gpu): the HIP path, called through the C ABI (ctypes), against the oracle on the same seeded inputs.

Tolerances (stated per north_star): logits/state max-abs <= 1e-3 * max(1, |ref|_inf) in Precision::Fp16 mode
(f16 GEMM operands), <= 2e-5 in Precision::Fp32 mode (hi+lo f16 split operands, fp32-class); greedy token ids
identical.  Byte/index work (token ids, state load/back round trips, snapshots) is bit-exact."""
import glob
import os

import numpy as np
import pytest

from ai00_server_amd import runtime as rt
from ai00_server_amd.harness import InferLoop, InferRequest, greedy_process, perplexity
from oracle import rwkv_ref as R

pytestmark = pytest.mark.gpu
GOLD = sorted(glob.glob(os.path.join(os.path.dirname(__file__), "golden", "*.npz")))
FP16_TOL, FP32_TOL = 1e-3, 2e-5


def tol(prec, want):
    return (FP32_TOL if prec == rt.Precision.Fp32 else FP16_TOL) * max(1.0, float(np.abs(want).max()))


def prompt(ref, slot, n):
    return [t % ref.info.num_vocab for t in R.synth_prompt(slot, n)]


def build(name, prec=rt.Precision.Fp32, quant=(0, 0), B=2, chunk=16, lora=None):
    t = R.synth_named(name)
    b = rt.ModelBuilder(R.st_serialize(t)).quant(quant[0], rt.Quant(quant[1]))
    if lora:
        b = b.lora(*lora)
    return t, b.build(max_batch=B, token_chunk_size=chunk, precision=prec)


def run_prompts(eng, prompts, option=rt.RnnOption.Last):
    B = eng.max_batch
    inp = rt.RnnInput([rt.RnnInputBatch(list(prompts[b]) if b < len(prompts) else [], option) for b in range(B)])
    rows = [[] for _ in range(B)]
    while inp.num_token() > 0:
        inp, outs = eng.infer(inp)
        for b, o in enumerate(outs):
            rows[b].extend(list(o))
    return [np.stack(r) if r else np.zeros((0, eng.info.num_vocab), np.float32) for r in rows]


@pytest.mark.parametrize("name", ["v5-tiny", "v6-tiny", "v7-tiny", "v5-small", "v6-small", "v7-small"])
@pytest.mark.parametrize("prec", [rt.Precision.Fp32, rt.Precision.Fp16], ids=["fp32", "fp16"])
def test_prefill_logits_and_state_match_oracle(name, prec):
    t, eng = build(name, prec, B=3, chunk=16)
    ref = R.RwkvRef(t)
    ps = [prompt(ref, s, 11 + 7 * s) for s in range(3)]
    got = run_prompts(eng, ps)
    for b in range(3):
        s = ref.init_state()
        want = ref.forward(ps[b], s)[-1]
        assert got[b].shape == (1, ref.info.num_vocab)
        assert np.abs(got[b][0] - want).max() <= tol(prec, want)
        back = eng.state.back(b)
        assert back.shape == s.shape and eng.state.shape == ref.state_shape()
        assert np.abs(back - s).max() <= tol(prec, s)
    eng.close()


@pytest.mark.parametrize("path", GOLD, ids=[os.path.basename(p) for p in GOLD])
def test_golden_fixtures_through_the_abi(path):
    g = np.load(path)
    ql, qt = int(g["quant_layers"]), int(g["quant_type"])
    t, eng = build(str(g["name"]), rt.Precision.Fp32, quant=(ql, qt), B=1, chunk=32)
    got = run_prompts(eng, [list(g["prompt"])])[0][0]
    assert np.abs(got - g["logits"]).max() <= FP32_TOL * max(1.0, float(np.abs(g["logits"]).max()))
    loop = InferLoop(eng)
    eng.state.load(eng.state.init(), 0)
    toks = greedy_process(loop, 0, list(g["prompt"]), 16)
    assert toks == list(g["greedy"])[:len(toks)] and (len(toks) == 16 or g["greedy"][len(toks)] == 0)
    eng.close()


@pytest.mark.parametrize("name,quant", [("v6-small", (3, 1)), ("v6-small", (2, 2)), ("v5-small", (2, 1)), ("v7-small", (3, 2)),
                                         ("v7-small", (1, 1))])
def test_quantised_layers_match_oracle_with_same_quantisation(name, quant):
    t, eng = build(name, rt.Precision.Fp32, quant=quant, B=2, chunk=8)
    ref = R.RwkvRef(t, quant_layers=quant[0], quant_type=quant[1])
    ps = [prompt(ref, 4, 19), prompt(ref, 5, 3)]
    got = run_prompts(eng, ps)
    for b in range(2):
        s = ref.init_state()
        want = ref.forward(ps[b], s)[-1]
        assert np.abs(got[b][0] - want).max() <= tol(rt.Precision.Fp32, want)     # bit-identical dequantisation
    # and quantisation changed the answer at all (guards against silently skipping it)
    ref0 = R.RwkvRef(t)
    s = ref0.init_state()
    assert np.abs(got[0][0] - ref0.forward(ps[0], s)[-1]).max() > 1e-4
    eng.close()


@pytest.mark.parametrize("name", ["v5-tiny", "v6-tiny", "v7-tiny"])
def test_greedy_ids_identical_over_many_steps(name):
    t, eng = build(name, rt.Precision.Fp16, B=2, chunk=8)
    ref = R.RwkvRef(t)
    p = prompt(ref, 6, 10)
    want, sw = ref.greedy(p, 96)
    loop = InferLoop(eng)
    got = greedy_process(loop, 1, p, 96)
    n = len(got)
    assert got == want[:n] and (n == 96 or want[n] == 0)
    # device-resident greedy loop (token ids fed back on the GPU) gives the same ids
    eng.state.load(eng.state.init(), 0)
    first = run_prompts(eng, [p])[0][0]
    t0 = int(np.argmax(first))
    toks, _ = eng.decode_greedy([t0], 40)
    assert [t0] + [int(x) for x in toks[:-1, 0]] == want[:40]
    eng.close()


def test_chunk_size_batch_layout_and_slot_neighbours_do_not_matter():
    t = R.synth_named("v6-small")
    st = R.st_serialize(t)
    ref = R.RwkvRef(t)
    p = prompt(ref, 7, 37)
    outs = []
    for B, chunk, slot in [(1, 64, 0), (4, 8, 2), (3, 5, 1), (8, 128, 7)]:
        eng = rt.ModelBuilder(st).build(max_batch=B, token_chunk_size=chunk, precision=rt.Precision.Fp32)
        others = [prompt(ref, 30 + b, 3 + 5 * b) for b in range(B)]
        others[slot] = p
        got = run_prompts(eng, others)
        outs.append((got[slot][0], eng.state.back(slot)))
        eng.close()
    for lg, stt in outs[1:]:
        assert np.abs(lg - outs[0][0]).max() <= 2e-5 and np.abs(stt - outs[0][1]).max() <= 2e-5
    # same T, different neighbour contents: bit-exact
    eng = rt.ModelBuilder(st).build(max_batch=2, token_chunk_size=64, precision=rt.Precision.Fp16)
    a = run_prompts(eng, [p, prompt(ref, 40, 37)])[0]
    eng.state.load(eng.state.init(), 0)
    eng.state.load(eng.state.init(), 1)
    b = run_prompts(eng, [p, prompt(ref, 41, 37)])[0]
    np.testing.assert_array_equal(a, b)
    eng.close()


def test_full_option_rows_and_perplexity():
    t, eng = build("v6-tiny", rt.Precision.Fp32, B=2, chunk=6)
    ref = R.RwkvRef(t)
    p = prompt(ref, 8, 17)
    got = run_prompts(eng, [p, []], rt.RnnOption.Full)[0]
    s = ref.init_state()
    want = ref.forward(p, s, full=True)
    assert got.shape == want.shape and np.abs(got - want).max() <= 2e-5
    eng.state.load(eng.state.init(), 0)
    loop = InferLoop(eng)
    choice = prompt(ref, 9, 9)
    ppl = perplexity(loop, 0, choice)
    s = ref.init_state()
    rows = ref.forward([0] + choice, s, full=True)
    assert abs(ppl - R.perplexity_ref(rows, choice)) < 1e-4
    eng.close()


def test_state_ops_are_bit_exact_and_snapshots_restore():
    t, eng = build("v7-small", rt.Precision.Fp16, B=3, chunk=16)
    ref = R.RwkvRef(t)
    rng = np.random.default_rng(5)
    slab = rng.standard_normal(eng.state.init().shape).astype(np.float32)
    eng.state.load(slab, 2)
    np.testing.assert_array_equal(eng.state.back(2), slab)                       # load -> back round trip
    assert not eng.state.back(0).any()                                           # other slots untouched (zero init)
    N = ref.info.head_size
    np.testing.assert_array_equal(eng.state.embed(1, 2), slab[1, 1:1 + N])       # back_layer == slice of the slab
    snap = eng.state.read(2)
    p = prompt(ref, 10, 9)
    a = run_prompts(eng, [[], [], p])[2]
    moved = eng.state.back(2)
    assert np.abs(moved - slab).max() > 1e-3
    eng.state.write(snap, 2)                                                     # restore, reuse snapshot twice
    np.testing.assert_array_equal(eng.state.back(2), slab)
    b = run_prompts(eng, [[], [], p])[2]
    np.testing.assert_array_equal(a, b)
    eng.state.write(snap, 1)
    np.testing.assert_array_equal(eng.state.back(1), slab)
    # continuing from a loaded state == oracle continuing from the same slab
    s = slab.copy()
    want = ref.forward(p, s)[-1]
    assert np.abs(b[0] - want).max() <= tol(rt.Precision.Fp16, want)
    eng.close()


@pytest.mark.parametrize("name", ["v5-tiny", "v6-tiny", "v7-tiny"])
def test_read_init_state_matches_oracle(name):
    t, eng = build(name, B=1)
    ref = R.RwkvRef(t)
    st = R.synth_init_state(ref.info)
    np.testing.assert_array_equal(eng.read_state(R.st_serialize(st)), ref.read_init_state(st))
    with pytest.raises(rt.RwkvError) as e:
        eng.read_state(R.st_serialize(t))                                        # a model file has no time_state
    assert e.value.code == -6
    eng.close()


def test_softmax_matches_and_runs_from_second_thread():
    import threading
    t, eng = build("v6-tiny", B=2)
    rng = np.random.default_rng(2)
    rows = [rng.standard_normal(eng.info.num_vocab).astype(np.float32) * 5 for _ in range(3)]
    res = {}
    th = threading.Thread(target=lambda: res.setdefault("p", rt.softmax(eng, rows)))
    th.start()
    ref = R.RwkvRef(t)
    run_prompts(eng, [prompt(ref, 1, 12)])                                      # infer concurrently (run.rs:1164-1190)
    th.join()
    for got, x in zip(res["p"], rows):
        np.testing.assert_allclose(got, R.softmax_ref(x[None])[0], rtol=1e-5, atol=1e-8)
    assert rt.softmax(eng, []) == []
    eng.close()


def test_lora_blend_at_load():
    t = R.synth_named("v6-tiny")
    rng = np.random.default_rng(3)
    C, r, alpha = 128, 8, 0.5
    A = (rng.standard_normal((C, r)) * 0.05).astype(np.float16)                  # `.lora.0`, stored [in, r]
    Bm = (rng.standard_normal((C, r)) * 0.05).astype(np.float16)                 # `.lora.1`, [out, r]
    lora = {"blocks.0.att.key.lora.0": A, "blocks.0.att.key.lora.1": Bm}
    _, eng = build("v6-tiny", rt.Precision.Fp32, B=1, lora=(R.st_serialize(lora), alpha))
    t2 = dict(t)
    w = t["blocks.0.att.key.weight"].astype(np.float32) + np.float32(alpha) * (Bm.astype(np.float32) @ A.astype(np.float32).T)
    t2["blocks.0.att.key.weight"] = w.astype(np.float16)
    ref = R.RwkvRef(t2)
    p = prompt(ref, 11, 9)
    s = ref.init_state()
    want = ref.forward(p, s)[-1]
    got = run_prompts(eng, [p])[0][0]
    assert np.abs(got - want).max() <= 5e-4 * max(1.0, np.abs(want).max())      # fp32 dot order differs before the fp16 rounding
    eng.close()


@pytest.mark.parametrize("name", ["v5-tiny", "v6-tiny", "v7-tiny"])
def test_lora_blends_vectors_as_well_as_matrices(name):
    """`LoraBlend::full(alpha)` (lib.rs:466-482) matches every `blocks.N.*` tensor: the engine blends the projection matrices
    (W += alpha B A^T) and whatever other per-block tensor the LoRA file holds under the model's own name (token-shift mixes, decay,
    LayerNorm weights: v = alpha l + (1 - alpha) v, before the load-time transform of the V5 decay); `ln_out.bias` in the file is outside
    the pattern and must be ignored.  Two LoRA files stack.  Against the oracle loaded the same way (web-rwkv's loader restated, unpinned)."""
    t = R.synth_named(name)
    rng = np.random.default_rng(13)
    C, r = 128, 8
    ver = R.model_info(t).version
    mixk = {5: "att.time_mix_k", 6: "att.time_mix_k", 7: "att.x_k"}[ver]
    l1 = {"blocks.0.att.key.lora.0": (rng.standard_normal((C, r)) * 0.05).astype(np.float16),
          "blocks.0.att.key.lora.1": (rng.standard_normal((C, r)) * 0.05).astype(np.float16),
          f"blocks.0.{mixk}": (rng.standard_normal(t[f"blocks.0.{mixk}"].shape) * 0.2).astype(np.float16),
          "blocks.1.ln1.weight": (rng.standard_normal(C) * 0.2).astype(np.float16),
          "ln_out.bias": (rng.standard_normal(C) * 0.2).astype(np.float16)}
    if ver != 7:
        l1["blocks.1.att.time_decay"] = (rng.standard_normal(t["blocks.1.att.time_decay"].shape) * 0.2).astype(np.float16)   # V5: blended BEFORE exp(-exp(.))
        l1["blocks.0.att.time_first"] = (rng.standard_normal(t["blocks.0.att.time_first"].shape) * 0.2).astype(np.float16)
    else:
        l1["blocks.1.att.w0"] = (rng.standard_normal(t["blocks.1.att.w0"].shape) * 0.2).astype(np.float16)
        l1["blocks.0.att.k_k"] = (rng.standard_normal(t["blocks.0.att.k_k"].shape) * 0.2).astype(np.float16)
    l2 = {"blocks.1.ffn.value.lora.0": (rng.standard_normal((t["blocks.1.ffn.value.weight"].shape[1], r)) * 0.05).astype(np.float16),
          "blocks.1.ffn.value.lora.1": (rng.standard_normal((C, r)) * 0.05).astype(np.float16),
          "blocks.1.ln1.weight": (rng.standard_normal(C) * 0.2).astype(np.float16)}            # the same vector again: blends stack
    b = rt.ModelBuilder(R.st_serialize(t)).lora(R.st_serialize(l1), 0.5).lora(R.st_serialize(l2), -0.75)
    eng = b.build(max_batch=1, token_chunk_size=16, precision=rt.Precision.Fp32)
    ref = R.RwkvRef(t, lora=[(l1, 0.5), (l2, -0.75)])
    p = prompt(ref, 14, 21)
    s = ref.init_state()
    want = ref.forward(p, s)[-1]
    got = run_prompts(eng, [p])[0][0]
    assert np.abs(got - want).max() <= 5e-4 * max(1.0, np.abs(want).max())      # the blend's fp32 dot order differs before the fp16 rounding
    s0 = R.RwkvRef(t).init_state()
    assert np.abs(got - R.RwkvRef(t).forward(p, s0)[-1]).max() > 1e-3           # the adapters did something
    eng.close()
    with pytest.raises(rt.RwkvError) as e:                                      # a vector of the wrong size fails the load, not a kernel
        rt.ModelBuilder(R.st_serialize(t)).lora(R.st_serialize({"blocks.0.ln1.weight": np.zeros(C // 2, np.float16)}), 1.0).build(max_batch=1)
    assert e.value.code == -2


def test_adapter_selection_and_device_names():
    """lib.rs:339-368: `list_adapters` names every device; `AdapterOption::Auto / Economical / Manual(n)` pick one; Manual(n) beyond
    the list fails like `ContextError::RequestAdapterFailed` (RWKV_ERR_DEVICE), it does not fall back."""
    names = rt.list_adapters()
    n = rt.lib().rwkv_device_count()
    assert n >= 1 and len(names) == n
    for nm in names:
        assert "MI3" in nm or "gfx950" in nm or "AMD" in nm or "Instinct" in nm, nm
    import ctypes as C
    small = C.create_string_buffer(4)
    assert rt.lib().rwkv_device_name(0, small, 4) == 0 and len(small.value) <= 3            # truncated, NUL-terminated
    assert rt.lib().rwkv_device_name(n, small, 4) == -4 and rt.lib().rwkv_device_name(-1, small, 4) != 0
    st = R.st_serialize(R.synth_named("v6-tiny"))
    for adapter in (-1, -2, 0, n - 1):                                          # Auto, Economical, Manual(0), Manual(last)
        eng = rt.ModelBuilder(st, adapter=adapter).build(max_batch=1, token_chunk_size=8)
        assert eng.device == (adapter if adapter >= 0 else 0)
        assert eng.max_batch == 1 and eng.token_chunk_size == 8                 # read back from the engine (rwkv_engine_token_chunk_size)
        eng.close()
    for bad in (n, n + 7):
        with pytest.raises(rt.RwkvError) as e:
            rt.ModelBuilder(st, adapter=bad).build(max_batch=1)
        assert e.value.code == -4
    with pytest.raises(rt.RwkvError):
        rt.ModelBuilder(st, adapter=-3).build(max_batch=1)                      # not an AdapterOption


def test_error_codes_never_abort():
    t, eng = build("v6-tiny", B=2, chunk=4)
    with pytest.raises(rt.RwkvError) as e:
        eng.state.back(5)
    assert e.value.code == -1
    with pytest.raises(rt.RwkvError):
        eng.state.load(np.zeros(7, np.float32), 0)
    with pytest.raises(rt.RwkvError):
        eng.infer(rt.RnnInput([rt.RnnInputBatch([1])]))                          # wrong number of slots
    # token ids beyond the vocabulary are clamped, not a crash
    out = run_prompts(eng, [[10 ** 6, 3]])[0]
    assert np.isfinite(out).all()
    with pytest.raises(rt.RwkvError) as e:
        rt.ModelBuilder(R.st_serialize(R.synth_named("v6-tiny"))).quant(2, rt.Quant.Int8).build()   # C=128 not /256
    assert e.value.code == -3
    eng.close()


def test_full_width_3b_shapes_two_layers():
    """BASELINE shapes (C=2560, F=8960, V=65536), 2 layers: fp16, int8 and nf4 against the oracle, plus the
    size-independent properties (chunked == unchunked, state round trip)."""
    tens = R.synth_checkpoint(6, 2, 2560, 8960, 65536, seed=11)
    st = R.st_serialize(tens)
    p = [int(x) for x in R.synth_prompt(12, 21)]
    for qt in (0, 1, 2):
        ref = R.RwkvRef(tens, 2 if qt else 0, qt)
        eng = rt.ModelBuilder(st).quant(2 if qt else 0, rt.Quant(qt)).build(max_batch=2, token_chunk_size=16,
                                                                             precision=rt.Precision.Fp16)
        got = run_prompts(eng, [p, p[:5]])
        s = ref.init_state()
        want = ref.forward(p, s)[-1]
        assert np.abs(got[0][0] - want).max() <= tol(rt.Precision.Fp16, want)
        assert int(np.argmax(got[0][0])) == int(np.argmax(want))
        back = eng.state.back(0)
        assert np.abs(back - s).max() <= tol(rt.Precision.Fp16, s)
        eng.state.load(back, 1)                                                  # continue in another slot
        a = run_prompts(eng, [[5], [5]])
        np.testing.assert_array_equal(a[0], a[1])
        eng.close()


@pytest.mark.parametrize("name,quant,prec", [("v6-small", (0, 0), rt.Precision.Fp32), ("v6-small", (3, 1), rt.Precision.Fp16),
                                             ("v5-small", (2, 1), rt.Precision.Fp32), ("v7-small", (3, 2), rt.Precision.Fp32),
                                             ("v7-tiny", (0, 0), rt.Precision.Fp16), ("v5-tiny", (0, 0), rt.Precision.Fp32)])
def test_prefill_tile_gemm_path(name, quant, prec):
    """Steps with >= 193 rows go through the LDS-tiled MFMA GEMM (gemm_tile_kernel; the 246-row step here), steps with
    33..192 rows through the four-tile decode GEMM (the 70-row Full request): ragged multi-slot prefill, Last and Full
    outputs, against the oracle and against the same prompts fed in small chunks (decode path)."""
    t, eng = build(name, prec, quant=quant, B=3, chunk=256)
    ref = R.RwkvRef(t, quant_layers=quant[0], quant_type=quant[1])
    ps = [prompt(ref, 20, 70), prompt(ref, 21, 131), prompt(ref, 22, 45)]
    got = run_prompts(eng, ps)
    for b in range(3):
        s = ref.init_state()
        want = ref.forward(ps[b], s)[-1]
        assert np.abs(got[b][0] - want).max() <= tol(prec, want)
        assert np.abs(eng.state.back(b) - s).max() <= tol(prec, s)
    for b in range(3):
        eng.state.load(eng.state.init(), b)
    full = run_prompts(eng, [ps[0], [], []], rt.RnnOption.Full)[0]
    s = ref.init_state()
    want = ref.forward(ps[0], s, full=True)
    assert full.shape == want.shape and np.abs(full - want).max() <= tol(prec, want)
    eng.close()
    _, eng2 = build(name, prec, quant=quant, B=3, chunk=16)            # same prompts through the decode-shaped path
    got2 = run_prompts(eng2, ps)
    for b in range(3):
        assert np.abs(got2[b][0] - got[b][0]).max() <= 2 * tol(prec, got[b][0])
    eng2.close()


def test_cpp_host_mirror_decode_loop(tmp_path):
    """harness/decode_loop.cpp (C++ mirror of the infer task + greedy process loop over include/rwkv_runtime.hpp)
    reproduces the oracle's greedy ids for two concurrent slots, Int8 layers included."""
    import subprocess
    from ai00_server_amd import build as B
    exe = B.build_harness(verbose=False) if not os.path.exists(B.HARNESS_BIN) else B.HARNESS_BIN
    t = R.synth_named("v6-small")
    path = tmp_path / "m.st"
    path.write_bytes(R.st_serialize(t))
    ref = R.RwkvRef(t, 2, R.QUANT_INT8)
    p0, p1 = prompt(ref, 50, 9), prompt(ref, 51, 23)
    args = [exe, str(path), "2", "1", "3", "8", "12"] + [str(x) for x in p0] + ["/"] + [str(x) for x in p1]
    out = subprocess.run(args, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr
    lines = [[int(x) for x in ln.split()] for ln in out.stdout.strip().splitlines()]
    for got, p in zip(lines, (p0, p1)):
        want, _ = ref.greedy(p, 12)
        n = len(got)
        assert got == want[:n] and (n == 12 or want[n] == 0)


@pytest.mark.parametrize("mode", ["nucleus", "typical", "mirostat"])
def test_cpp_sampler_state_drives_on_device_sampling(tmp_path, mode):
    """harness/decode_loop.cpp with RWKV_DECODE_SAMPLER: include/rwkv_sampler.hpp (penalty maps / max_surprise on the host) +
    Runtime::infer_sample over the real engine.  The Python mirrors (harness.NucleusSampler & co, themselves checked against the
    reference samplers' restatements elsewhere in this file) replay the same loop with the same uniform draws on a second engine:
    the ids must be identical token for token — same kernel, same adjustments, same feedback."""
    import subprocess
    from ai00_server_amd import build as B
    from ai00_server_amd import harness as H
    exe = B.build_harness(verbose=False) if not os.path.exists(B.HARNESS_BIN) else B.HARNESS_BIN
    t = R.synth_named("v6-small")
    path = tmp_path / "m.st"
    path.write_bytes(R.st_serialize(t))
    ref = R.RwkvRef(t)
    p0, p1 = prompt(ref, 52, 9), prompt(ref, 53, 14)
    n_new = 10
    args = [exe, str(path), "2", "1", "3", "8", str(n_new)] + [str(x) for x in p0] + ["/"] + [str(x) for x in p1]
    out = subprocess.run(args, capture_output=True, text=True, timeout=300, env=dict(os.environ, RWKV_DECODE_SAMPLER=mode))
    assert out.returncode == 0, out.stderr
    got = [[int(x) for x in ln.split()] for ln in out.stdout.strip().splitlines()]
    _, eng = build("v6-small", rt.Precision.Fp16, quant=(2, 1), B=3, chunk=8)
    make = {"nucleus": H.NucleusSampler, "typical": H.TypicalSampler, "mirostat": H.MirostatSampler}[mode]
    smp = [make(), make(), None]
    pend = [list(p0), list(p1), []]
    for b in range(2):
        smp[b].init(pend[b])
    want = [[], []]
    for step in range(n_new):
        us = [float(np.fmod(np.float32(0.137) + np.float32(0.618034) * np.float32(step + 1) + np.float32(0.31) * np.float32(b), np.float32(1.0)))
              for b in range(3)]
        inp = rt.RnnInput([rt.RnnInputBatch(list(pend[b])) for b in range(3)])
        res = [None] * 3
        while inp.num_token() > 0:
            inp, outs = eng.infer_sample(inp, smp, us)
            for b in range(2):
                if outs[b] is not None:
                    res[b] = outs[b]
        for b in range(2):
            tok, prob = res[b]
            smp[b].update(prob if mode == "mirostat" else tok)
            want[b].append(tok)
            pend[b] = [tok]
    eng.close()
    assert got == want


def test_full_v6_3b_greedy_ids_match_oracle():
    """BASELINE headline shape, all 32 layers (fp16 weights): 24-token prompt + 24 greedy tokens, ids identical to
    the fp32 oracle; last-prompt logits within the Fp16 tolerance; then the same ids again from the device-resident
    greedy loop and from an Int8 engine compared with itself across batch layouts (size-independent property)."""
    import psutil
    if psutil.virtual_memory().available < 48 << 30:
        pytest.skip("needs ~40 GB of host RAM for the fp32 oracle of a 3B model")
    st, tens = R.synth_st("v6-3b", fast=True)
    ref = R.RwkvRef(tens)
    p = [int(x) for x in R.synth_prompt(60, 24)]
    want, _ = ref.greedy(p, 24)
    eng = rt.ModelBuilder(st).build(max_batch=2, token_chunk_size=32, precision=rt.Precision.Fp16)
    got_logits = run_prompts(eng, [p])[0][0]
    s = ref.init_state()
    wl = ref.forward(p, s)[-1]
    assert np.abs(got_logits - wl).max() <= tol(rt.Precision.Fp16, wl)
    eng.state.load(eng.state.init(), 0)
    got = greedy_process(InferLoop(eng), 0, p, 24)
    assert got == want[:len(got)] and (len(got) == 24 or want[len(got)] == 0)
    eng.close()
    del ref
    # Int8 engine: slot 0 alone == slot 1 next to a busy neighbour (bit-exact), and chunked == unchunked (<= 1e-3)
    e8 = rt.ModelBuilder(st).quant(32, rt.Quant.Int8).build(max_batch=2, token_chunk_size=64)
    a = run_prompts(e8, [p])[0][0]
    e8.state.load(e8.state.init(), 0)
    b = run_prompts(e8, [[int(x) for x in R.synth_prompt(61, 24)], p])[1][0]
    # (not bit-exact: the two steps have different row counts, hence different GEMM tilings)
    assert np.abs(a - b).max() <= 1e-3 * max(1.0, float(np.abs(a).max()))
    e8.close()
    e8 = rt.ModelBuilder(st).quant(32, rt.Quant.Int8).build(max_batch=1, token_chunk_size=8)
    c = run_prompts(e8, [p])[0][0]
    assert np.abs(a - c).max() <= 1e-3 * max(1.0, float(np.abs(a).max()))
    e8.close()


def test_on_device_nucleus_sampling_matches_reference_sampler():
    """rwkv_infer_sample (penalties/bias -> softmax -> top-k -> top-p -> temperature -> inverse CDF on the device)
    against the restatement of sampler/nucleus.rs on the SAME logits: every step the state is snapshotted, the host
    path produces the logits the oracle sampler sees, the snapshot is restored and the device samples."""
    from ai00_server_amd.harness import NucleusSampler
    t, eng = build("v6-small", rt.Precision.Fp16, B=3, chunk=16)
    ref = R.RwkvRef(t)
    rng = np.random.default_rng(77)
    cfgs = [dict(top_p=0.5, top_k=128, temperature=1.0), dict(top_p=0.9, top_k=40, temperature=0.7),
            dict(top_p=0.3, top_k=256, temperature=1.5, presence_penalty=0.6, frequency_penalty=0.1)]
    dev = [NucleusSampler(bias={5: 1.5, 9: -2.0}, **c) for c in cfgs]
    orc = [R.NucleusRef(**c) for c in cfgs]
    prompts = [prompt(ref, 70 + b, 6 + b) for b in range(3)]
    for b in range(3):
        dev[b].init(prompts[b][-3:])
        orc[b].init(prompts[b][-3:])
    pending = [list(p) for p in prompts]
    checked = skipped = 0
    for step in range(24):
        snaps = [eng.state.read(b) for b in range(3)]
        inp = rt.RnnInput([rt.RnnInputBatch(list(pending[b]), rt.RnnOption.Last) for b in range(3)])
        rows = [None] * 3
        while inp.num_token() > 0:
            inp, outs = eng.infer(inp)
            for b, o in enumerate(outs):
                if len(o):
                    rows[b] = o[-1]
        us = [float(rng.random()) for _ in range(3)]
        want, margin = [], []
        for b in range(3):
            x = orc[b].transform(rows[b])
            for tk, bv in {5: 1.5, 9: -2.0}.items():
                x[tk] += np.float32(bv)
            tok, mg = R.nucleus_ref(R.softmax_ref(x[None])[0], orc[b].top_p, orc[b].top_k, orc[b].temperature, us[b])
            want.append(tok)
            margin.append(mg)
        for b in range(3):
            eng.state.write(snaps[b], b)
        inp = rt.RnnInput([rt.RnnInputBatch(list(pending[b]), rt.RnnOption.Last) for b in range(3)])
        got = [None] * 3
        while inp.num_token() > 0:
            inp, outs = eng.infer_sample(inp, dev, us)
            for b, o in enumerate(outs):
                if o is not None:
                    got[b] = o
        for b in range(3):
            if margin[b] > 1e-4:                                   # away from a CDF boundary: ids must agree
                assert got[b][0] == want[b], (step, b, got[b], want[b], margin[b])
                checked += 1
            else:
                skipped += 1
            tok = want[b]                                          # follow the reference trajectory
            dev[b].update(tok)
            orc[b].update(tok)
            pending[b] = [tok]
            assert 0.0 < got[b][1] <= 1.0
    assert checked >= 60 and skipped <= 6
    # greedy corner: top_k = 1 always returns the arg-max regardless of the draw
    g = NucleusSampler(top_p=0.0, top_k=1, presence_penalty=0.0, frequency_penalty=0.0)
    inp = rt.RnnInput([rt.RnnInputBatch([3, 4, 5]), rt.RnnInputBatch(), rt.RnnInputBatch()])
    snap = eng.state.read(0)
    _, outs = eng.infer(inp)
    eng.state.write(snap, 0)
    inp = rt.RnnInput([rt.RnnInputBatch([3, 4, 5]), rt.RnnInputBatch(), rt.RnnInputBatch()])
    _, sm = eng.infer_sample(inp, [g, None, None], [0.99, 0.0, 0.0])
    assert sm[0][0] == int(np.argmax(outs[0][-1])) and sm[1] is None
    eng.close()


def test_on_device_sampling_full_vocabulary_ties_and_top_k_zero():
    """The V = 65536 instantiation of the sampling kernel (no bound predicates), on a one-layer model whose head holds 8000
    copies of the arg-max row: an 8001-way tie at the top of the distribution, more than the candidate buffer holds, so the
    candidate set has to come from the id-bound search (lowest ids first, as the oracle's restatement orders ties — the
    reference's own tie order is undefined, its sort is unstable, nucleus.rs:76).  Also `top_k = 0` (nothing kept ->
    `unwrap_or_default` -> token 0, nucleus.rs:78-101) and a plain row without ties."""
    from ai00_server_amd.harness import NucleusSampler
    tens = R.synth_checkpoint(6, 1, 128, 448, 65536, seed=5)
    p = [int(x) for x in R.synth_prompt(90, 5)]
    base = R.RwkvRef(tens).forward(p, R.RwkvRef(tens).init_state())[-1]
    a = int(np.argmax(base))
    tens["head.weight"][1000:9000] = tens["head.weight"][a]
    ref = R.RwkvRef(tens)
    logits = ref.forward(p, ref.init_state())[-1]
    assert (logits[1000:9000] == logits[a]).all() and logits.max() == logits[a]
    probs = R.softmax_ref(logits[None])[0]
    eng = rt.ModelBuilder(R.st_serialize(tens)).build(max_batch=4, token_chunk_size=16, precision=rt.Precision.Fp32)
    rows = run_prompts(eng, [p])[0]
    dev_probs = R.softmax_ref(rows[0][None])[0]
    ties = np.nonzero(dev_probs == dev_probs.max())[0]
    assert len(ties) >= 8000                                              # the device logits tie too (identical head rows)
    rng = np.random.default_rng(3)
    checked = 0
    for top_k, top_p, temp in [(256, 1.0, 1.0), (40, 1.0, 0.8), (256, 0.5, 1.3), (1, 0.0, 1.0)]:
        for _ in range(6):
            u = float(rng.random())
            want, margin = R.nucleus_ref(dev_probs, top_p, top_k, temp, u)
            smp = NucleusSampler(top_p=top_p, top_k=top_k, temperature=temp, presence_penalty=0.0, frequency_penalty=0.0)
            for b in range(4):
                eng.state.load(eng.state.init(), b)
            inp = rt.RnnInput([rt.RnnInputBatch(list(p))] + [rt.RnnInputBatch() for _ in range(3)])
            _, out = eng.infer_sample(inp, [smp, None, None, None], [u, 0.0, 0.0, 0.0])
            if margin > 1e-4:
                assert out[0][0] == want, (top_k, top_p, temp, u, out[0], want)
                checked += 1
    assert checked >= 18
    smp = NucleusSampler(top_p=0.5, top_k=0, presence_penalty=0.0, frequency_penalty=0.0)
    eng.state.load(eng.state.init(), 0)
    _, out = eng.infer_sample(rt.RnnInput([rt.RnnInputBatch(list(p))] + [rt.RnnInputBatch() for _ in range(3)]), [smp, None, None, None], [0.3, 0, 0, 0])
    assert out[0][0] == 0
    eng.close()


def test_on_device_sampling_with_a_formatter_mask():
    """`Formatter::transform` on the device path (run.rs:676-683, sampler/bnf.rs:35-38): the grammar's allowed-token set arrives
    as a dense mask; forbidden logits become -inf after the penalties and before the bias, exactly as the host path orders them."""
    from ai00_server_amd.harness import NucleusSampler
    t, eng = build("v6-small", rt.Precision.Fp32, B=2, chunk=16)
    ref = R.RwkvRef(t)
    V = ref.info.num_vocab
    p = prompt(ref, 140, 9)
    logits = ref.forward(p, ref.init_state())[-1]
    rng = np.random.default_rng(9)
    checked = 0
    for trial in range(12):
        allow = (rng.random(V) < 0.08).astype(np.uint8)
        allow[int(np.argmax(logits))] = 0                                 # the unconstrained favourite is forbidden
        allow[rng.integers(0, V)] = 1
        x = logits.copy()
        x[allow == 0] = -np.inf
        x[7] += np.float32(2.0)                                           # bias on a token: stays -inf if forbidden
        u = float(rng.random())
        want, margin = R.nucleus_ref(R.softmax_ref(x[None])[0], 0.9, 50, 1.1, u)
        smp = NucleusSampler(top_p=0.9, top_k=50, temperature=1.1, presence_penalty=0.0, frequency_penalty=0.0, bias={7: 2.0})
        smp.allow = allow
        for b in range(2):
            eng.state.load(eng.state.init(), b)
        _, out = eng.infer_sample(rt.RnnInput([rt.RnnInputBatch(list(p)), rt.RnnInputBatch()]), [smp, None], [u, 0.0])
        assert allow[out[0][0]] == 1
        if margin > 1e-4:
            assert out[0][0] == want, (trial, out[0], want)
            checked += 1
    assert checked >= 8
    eng.close()


def test_infer_into_pageable_and_pinned_destinations():
    """rwkv_infer copies logits straight into pinned destinations (rwkv_host_alloc; the Python mirror hands out pieces of one
    pinned block) and stages them for pageable ones: same bits either way, also for scattered per-slot buffers."""
    import ctypes as C
    t, eng = build("v6-small", rt.Precision.Fp16, B=3, chunk=16)
    ref = R.RwkvRef(t)
    ps = [prompt(ref, 80 + b, 4 + b) for b in range(3)]
    want = run_prompts(eng, ps)                                          # pinned path (runtime arena)
    V = ref.info.num_vocab
    for b in range(3):
        eng.state.load(eng.state.init(), b)
    bufs = [np.full((1, V), np.nan, np.float32) for _ in range(3)]      # pageable, not contiguous
    toks = [np.asarray(x, dtype=np.uint32) for x in ps]
    ins = (rt._SlotInC * 3)(*[rt._SlotInC(toks[b].ctypes.data_as(C.POINTER(C.c_uint32)), toks[b].size, 0, 0) for b in range(3)])
    outs = (rt._SlotOutC * 3)(*[rt._SlotOutC(bufs[b].ctypes.data_as(C.POINTER(C.c_float)), 1, 0, 0) for b in range(3)])
    rt._check(rt.lib().rwkv_infer(eng._h, ins, outs))
    for b in range(3):
        assert outs[b].n_rows == 1 and outs[b].n_consumed == len(ps[b])
        np.testing.assert_array_equal(bufs[b][0], want[b][0])
    eng.close()


@pytest.mark.parametrize("name", ["v5-small", "v6-small", "v7-small"])
def test_single_token_steps_with_ln_prologue_interleave_with_chunks(name):
    """T = 1 steps run LayerNorm + token shift as a prologue of the consuming kernel and commit the shift state in
    the following launch; multi-token steps use the row kernel.  Alternating the two on a non-zero slot must give the
    oracle's logits at every step and its state at the end (Fp16 mode: the fused path is only taken there)."""
    t, eng = build(name, rt.Precision.Fp16, B=3, chunk=8)
    ref = R.RwkvRef(t)
    p = prompt(ref, 11, 9)
    sw = ref.init_state()
    want = ref.forward(p, sw, full=True)
    cuts = [1, 3, 1, 1, 2, 1]                                   # 9 tokens
    pos, slot = 0, 2
    for n in cuts:
        inp = rt.RnnInput([rt.RnnInputBatch(p[pos:pos + n] if b == slot else [], rt.RnnOption.Last) for b in range(3)])
        inp, outs = eng.infer(inp)
        assert inp.num_token() == 0
        pos += n
        got = outs[slot][-1]
        assert np.abs(got - want[pos - 1]).max() <= tol(rt.Precision.Fp16, want[pos - 1])
    st = eng.state.back(slot)
    assert np.abs(st - sw).max() <= tol(rt.Precision.Fp16, sw)
    eng.close()


@pytest.mark.parametrize("name,quant", [("v6-small", (3, 1)), ("v7-small", (2, 2)), ("v5-small", (0, 0))])
def test_prefab_round_trip_is_bit_exact(name, quant, tmp_path):
    """Prefab save/load (`ModelSerialize::serialize` lib.rs:131-154, `LoadType::Prefab` lib.rs:517-553): a model saved after
    quantisation + LoRA blend and loaded back from the image gives bit-identical logits and state, and reports the
    same info and weight bytes; a LoRA on top of a prefab is refused."""
    t = R.synth_named(name)
    st = R.st_serialize(t)
    rng = np.random.default_rng(5)
    C = R.model_info(t).num_emb
    lora = R.st_serialize({"blocks.0.att.output.lora.0": (rng.standard_normal((C, 4)) * 0.05).astype(np.float16),
                           "blocks.0.att.output.lora.1": (rng.standard_normal((C, 4)) * 0.05).astype(np.float16)})
    eng = rt.ModelBuilder(st).quant(quant[0], rt.Quant(quant[1])).lora(lora, 0.5).build(max_batch=2, token_chunk_size=16)
    ref = R.RwkvRef(t)
    ps = [prompt(ref, 3, 13), prompt(ref, 4, 7)]
    a = run_prompts(eng, ps)
    sa = [eng.state.back(b) for b in range(2)]
    path = str(tmp_path / "model.prefab")
    eng.save_prefab(path)
    info, wb = eng.info, eng.weight_bytes
    eng.close()
    image = open(path, "rb").read()
    assert image[:7] == b"RWKVHIP" and rt.Loader.info(image) == info
    eng2 = rt.ModelBuilder(image).build(max_batch=2, token_chunk_size=16)          # quant settings come from the image
    assert eng2.info == info and eng2.weight_bytes == wb
    b = run_prompts(eng2, ps)
    for x, y in zip(a, b):
        np.testing.assert_array_equal(x, y)
    for bi in range(2):
        np.testing.assert_array_equal(sa[bi], eng2.state.back(bi))
    eng2.close()
    with pytest.raises(rt.RwkvError) as e:
        rt.ModelBuilder(image).lora(lora, 0.5).build(max_batch=1)
    assert e.value.code == -3


def test_cpp_scheduler_serves_real_engine(tmp_path):
    """harness/serve_loop.cpp: include/rwkv_scheduler.hpp (slot choice, prefix cache, continuous batching — run.rs:289-331,
    441-662, 1113-1157) over the real engine.  A runs alone, B joins mid-flight, both decode greedily; C = A's whole
    history + a tail continues from the cached state.  Token ids must equal the oracle's for every stage."""
    import subprocess
    from ai00_server_amd import build as B
    B.build(verbose=False)
    t = R.synth_named("v6-small")
    path = tmp_path / "m.st"
    path.write_bytes(R.st_serialize(t))
    ref = R.RwkvRef(t, 2, R.QUANT_INT8)
    p0, p1, tail = prompt(ref, 60, 21), prompt(ref, 61, 5), prompt(ref, 62, 4)
    n_new = 6
    args = [B.SERVE_BIN, str(path), "2", "1", "3", "8", str(n_new)] + [str(x) for x in p0] + ["/"] + [str(x) for x in p1] + ["/"] + [str(x) for x in tail]
    out = subprocess.run(args, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr
    lines = out.stdout.strip().splitlines()
    gen_a, gen_b, gen_c = ([int(x) for x in ln.split()] for ln in lines[:3])
    meta = [int(x) for x in lines[3].split()[1:]]
    want_a, _ = ref.greedy(p0, n_new)
    want_b, _ = ref.greedy(p1, n_new)
    assert gen_a == want_a[:n_new] and gen_b == want_b[:n_new]
    want_c, _ = ref.greedy(p0 + gen_a + tail, n_new)
    assert gen_c == want_c[:n_new]
    riders_first, riders_second, slot_a, slot_c, rc, c_prefix = meta
    assert riders_first == 1 and riders_second == 2          # 21 prompt tokens at chunk 8: A is mid-flight when B joins
    assert slot_c == slot_a and rc == 0                       # Continue on A's slot, SlotResult::Success
    assert c_prefix == len(p0) + n_new                        # checked out at A's full history (prompt + every fed token)
    # stage 5: GenerateKind::Choose (run.rs:936-979) on C's slot, calibrated, against the oracle's own evaluation
    ch = lines[4].split()
    assert ch[0] == "choose" and ch[3] == "inf" and ch[4:] == ["restored", "1"]
    ctx = p0 + gen_a + tail + gen_c
    st = ref.init_state()
    lg = ref.forward(ctx, st)[-1].astype(np.float32)
    probs = np.exp(lg - lg.max())
    probs /= probs.sum(dtype=np.float32)

    def ref_ppl(state, toks, head):                           # run.rs:699-755
        all_ = list(toks) if head is not None else [0] + list(toks)
        p = [head] if head is not None else []
        rows = ref.forward(all_, state.copy(), full=True)
        for j in range(1, len(all_)):
            e = np.exp(rows[j - 1].astype(np.float32))
            p.append(float(e[all_[j]] / e.sum(dtype=np.float32)))
        return -sum(np.log(x) for x in p) / len(all_)

    for got, choice in zip(ch[1:3], (p1[:3], tail)):
        want = -ref_ppl(ref.init_state(), choice, None) + ref_ppl(st, choice, float(probs[choice[0]]))
        assert abs(float(got) - want) <= 5e-3 * max(1.0, abs(want)), (got, want)


def test_cpp_router_over_two_real_engines_on_one_device(tmp_path):
    """SURVEY 8(e) without a second GPU: harness/router_loop.cpp builds TWO engines of one model on device 0, drives each from
    its own thread through include/rwkv_router.hpp and routes six requests (more than one replica holds) by least-busy, then
    a follow-up by prefix affinity.  Every request's greedy ids equal the oracle's; both replicas served requests; the
    follow-up went back to the replica that cached its prefix."""
    import subprocess
    from ai00_server_amd import build as B
    B.build_harness(verbose=False)
    t = R.synth_named("v6-small")
    path = tmp_path / "m.st"
    path.write_bytes(R.st_serialize(t))
    ref = R.RwkvRef(t)
    ps = [prompt(ref, 120 + i, 5 + 2 * i) for i in range(6)]
    n_new = 6
    args = [B.ROUTER_BIN, str(path), "2", "0,0", "3", "16", str(n_new)]       # device list: replica r on devices[r % len]
    for i, p in enumerate(ps):
        args += ([] if i == 0 else ["/"]) + [str(x) for x in p]
    out = subprocess.run(args, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr
    assert "over 2 replicas on 2 device(s)" in out.stderr
    bad = subprocess.run([B.ROUTER_BIN, str(path), "1", str(rt.lib().rwkv_device_count()), "1", "16", "1", "5"], capture_output=True, text=True, timeout=120)
    assert bad.returncode == 1 and "adapter index out of range" in bad.stderr             # a replica on a device that does not exist
    lines = out.stdout.strip().splitlines()
    rows = [[int(x) for x in ln.split()] for ln in lines[:7]]
    for i, p in enumerate(ps):
        want, _ = ref.greedy(p, n_new)
        assert rows[i][1:] == want, (i, rows[i], want)
    assert {r[0] for r in rows[:6]} == {0, 1}                            # least-busy placement used both replicas
    follow = ps[0] + rows[0][1:] + [rows[0][-1]]
    want, _ = ref.greedy(follow, n_new)
    assert rows[6][1:] == want
    meta = [int(x) for x in lines[7].split()[1:]]
    assert meta[0] == meta[2] == rows[6][0] and meta[1] == len(follow) - 1   # routed to the replica that cached prompt + output
    # the prompts again as an `/embeddings` batch job over both replicas (ReplicaRouter::embed_documents: State-kind requests, asynchronous
    # read-back): the last layer's WKV rows of each document against the oracle's state of that document prefilled alone
    emb = [[float(x) for x in ln.split()[1:]] for ln in lines if ln.startswith("emb ")]
    assert len(emb) == 6
    for i, p in enumerate(ps):
        st = ref.init_state()
        ref.forward(p, st)
        rows_ = st[-1, 1:-1].reshape(-1)
        tol = 1e-3 * max(1.0, float(np.abs(rows_).max()))
        n = rows_.size
        assert abs(emb[i][1] - rows_[0]) <= tol and abs(emb[i][2] - rows_[n // 3]) <= tol and abs(emb[i][3] - rows_[n - 1]) <= tol, i
        assert abs(emb[i][0] - float(rows_.sum(dtype=np.float64))) <= tol * np.sqrt(n) , i
    esteps = [int(x) for x in [ln for ln in lines if ln.startswith("embsteps")][0].split()[1:]]
    assert min(esteps) > 0                                                # both replicas took documents


def test_on_device_typical_sampling_matches_reference_sampler():
    """rwkv_infer_sample with kind = Typical (keys |(-ln p) - H| ascending -> top_k -> tau -> temperature -> inverse CDF on
    the device) against the restatement of sampler/typical.rs on the SAME logits, penalties and bias included."""
    from ai00_server_amd.harness import TypicalSampler
    t, eng = build("v6-small", rt.Precision.Fp16, B=3, chunk=16)
    ref = R.RwkvRef(t)
    rng = np.random.default_rng(78)
    cfgs = [dict(tau=0.5, top_k=128, temperature=1.0), dict(tau=0.9, top_k=40, temperature=0.7),
            dict(tau=0.3, top_k=256, temperature=1.5, presence_penalty=0.6, frequency_penalty=0.1)]
    dev = [TypicalSampler(bias={5: 1.5, 9: -2.0}, **c) for c in cfgs]
    prompts = [prompt(ref, 80 + b, 6 + b) for b in range(3)]
    for b in range(3):
        dev[b].init(prompts[b][-3:])
    pending = [list(p) for p in prompts]
    checked = skipped = 0
    for step in range(24):
        snaps = [eng.state.read(b) for b in range(3)]
        inp = rt.RnnInput([rt.RnnInputBatch(list(pending[b]), rt.RnnOption.Last) for b in range(3)])
        rows = [None] * 3
        while inp.num_token() > 0:
            inp, outs = eng.infer(inp)
            for b, o in enumerate(outs):
                if len(o):
                    rows[b] = o[-1]
        us = [float(rng.random()) for _ in range(3)]
        want, margin = [], []
        for b in range(3):
            x = rows[b].astype(np.float32).copy()
            for tk, dv in dev[b].adjustments().items():            # -penalty (typical.rs:62-68) + bias (run.rs:681-683)
                x[tk] += np.float32(dv)
            pr = R.softmax_ref(x[None])[0]
            alts = [R.typical_ref(pr, dev[b].tau, dev[b].top_k, dev[b].temperature, us[b], h_shift=d)
                    for d in (0.0, 1e-5, -1e-5, 4e-5, -4e-5)]   # H is a 65k-term fp32 sum: order-dependent last bits
            want.append({a[0] for a in alts})
            margin.append(min(a[1] for a in alts))
        for b in range(3):
            eng.state.write(snaps[b], b)
        inp = rt.RnnInput([rt.RnnInputBatch(list(pending[b]), rt.RnnOption.Last) for b in range(3)])
        got = [None] * 3
        while inp.num_token() > 0:
            inp, outs = eng.infer_sample(inp, dev, us)
            for b, o in enumerate(outs):
                if o is not None:
                    got[b] = o
        for b in range(3):
            if margin[b] > 1e-4:
                assert got[b][0] in want[b], (step, b, got[b], want[b], margin[b])
                checked += len(want[b]) == 1                       # the answer did not depend on the last bits of H
            else:
                skipped += 1
            dev[b].update(got[b][0])
            pending[b] = [got[b][0]]
            assert 0.0 < got[b][1] <= 1.0
    assert checked >= 50 and skipped <= 8, (checked, skipped)
    eng.close()


def test_on_device_mirostat_sampling_matches_reference_sampler():
    """rwkv_infer_sample with kind = Mirostat (sort descending -> truncate at max_surprise -> u * sum against the running
    sum, token surprise returned) against the restatement of sampler/mirostat.rs on the SAME logits; the host state
    machine (max_surprise update, mirostat.rs:85-87) runs on both sides and must stay in step.  One slot mixes in a
    nucleus sampler so that both kernel instantiations run in one call."""
    from ai00_server_amd.harness import MirostatSampler, NucleusSampler
    t, eng = build("v6-small", rt.Precision.Fp16, B=3, chunk=16)
    ref = R.RwkvRef(t)
    rng = np.random.default_rng(79)
    dev = [MirostatSampler(tau=3.0, rate=0.1), MirostatSampler(tau=2.0, rate=0.3), NucleusSampler(top_p=0.8, top_k=50)]
    orc_ms = [np.float32(6.0), np.float32(4.0)]
    pending = [prompt(ref, 90 + b, 5 + b) for b in range(3)]
    checked = skipped = 0
    for step in range(20):
        snaps = [eng.state.read(b) for b in range(3)]
        inp = rt.RnnInput([rt.RnnInputBatch(list(pending[b]), rt.RnnOption.Last) for b in range(3)])
        rows = [None] * 3
        while inp.num_token() > 0:
            inp, outs = eng.infer(inp)
            for b, o in enumerate(outs):
                if len(o):
                    rows[b] = o[-1]
        us = [float(rng.random()) for _ in range(3)]
        want = [R.mirostat_ref(R.softmax_ref(rows[b][None])[0], float(orc_ms[b]), us[b]) for b in range(2)]
        for b in range(3):
            eng.state.write(snaps[b], b)
        inp = rt.RnnInput([rt.RnnInputBatch(list(pending[b]), rt.RnnOption.Last) for b in range(3)])
        got = [None] * 3
        while inp.num_token() > 0:
            inp, outs = eng.infer_sample(inp, dev, us)
            for b, o in enumerate(outs):
                if o is not None:
                    got[b] = o
        for b in range(2):
            tok, surprise, margin = want[b]
            assert abs(float(dev[b].max_surprise) - float(orc_ms[b])) < 1e-3
            if margin > 1e-5:
                assert got[b][0] == tok, (step, b, got[b], want[b])
                assert abs(got[b][1] - surprise) < 1e-3 * max(1.0, abs(surprise))
                checked += 1
            else:
                skipped += 1
            # both state machines follow the reference trajectory (mirostat.rs:85-87)
            tgt, rate = (3.0, 0.1) if b == 0 else (2.0, 0.3)
            orc_ms[b] = np.float32(min(np.float32(orc_ms[b] - np.float32(rate) * np.float32(np.float32(surprise) - np.float32(tgt))), np.float32(4.0 * tgt)))
            dev[b].update(surprise)
            pending[b] = [tok]
        dev[2].update(got[2][0])
        pending[2] = [got[2][0]]
    assert checked >= 36 and skipped <= 4, (checked, skipped)
    eng.close()


def test_full_width_v7_shapes_two_layers_nf4():
    """BASELINE config #4 shapes (RWKV-V7 2.9B: C=2560, V=65536; 2 layers), NF4: a chunked prompt (chunk WKV kernel,
    tile-free GEMMs) followed by single-token steps (LayerNorm prologue in the grouped r/k/v/LoRA GEMM and in Fk) against
    the oracle, logits and state."""
    _, _, C, F, V = R.CONFIGS["v7-2.9b"]
    tens = R.synth_checkpoint(7, 2, C, F, V, seed=13)
    st = R.st_serialize(tens)
    ref = R.RwkvRef(tens, 2, R.QUANT_NF4)
    eng = rt.ModelBuilder(st).quant(2, rt.Quant.NF4).build(max_batch=2, token_chunk_size=16, precision=rt.Precision.Fp16)
    p = [int(x) for x in R.synth_prompt(14, 19)]
    s = ref.init_state()
    want = ref.forward(p, s, full=True)
    got = run_prompts(eng, [[], p[:13]])[1][0]                    # slot 1: prompt in one chunked call
    assert np.abs(got - want[12]).max() <= tol(rt.Precision.Fp16, want[12])
    for i in range(13, 19):                                        # then token by token (T = 1 steps)
        got = run_prompts(eng, [[], [p[i]]])[1][0]
        assert np.abs(got - want[i]).max() <= tol(rt.Precision.Fp16, want[i])
        assert int(np.argmax(got)) == int(np.argmax(want[i]))
    assert np.abs(eng.state.back(1) - s).max() <= tol(rt.Precision.Fp16, s)
    eng.close()


@pytest.mark.parametrize("name", ["v6-small", "v7-small"])
def test_repeated_dense_steps_reuse_the_plan_and_survive_pattern_changes(name):
    """A decode loop repeats one slot pattern, so the engine reuses the previous call's plan and reads the token ids of a dense
    step from pinned host memory instead of uploading metadata.  Patterns that alternate — all three slots, slot 0 alone (dense
    with one row), slot 1 alone (not dense: row 0 is slot 1), slots {0, 2}, all three again, through rwkv_infer, rwkv_infer_sample
    and rwkv_decode_greedy — must leave every slot exactly where the oracle's independent per-slot recurrences are."""
    t, eng = build(name, rt.Precision.Fp32, B=3, chunk=16)
    ref = R.RwkvRef(t)
    V = ref.info.num_vocab
    states = [ref.init_state() for _ in range(3)]
    rng = np.random.default_rng(5)
    patterns = [(0, 1, 2), (0, 1, 2), (0,), (0,), (1,), (1,), (0, 2), (0, 1, 2), (0, 1, 2), (2,), (0, 1, 2)]
    for step, act in enumerate(patterns):
        toks = {b: int(rng.integers(1, V)) for b in act}
        want = {b: ref.forward([toks[b]], states[b])[-1] for b in act}
        if step % 3 == 2:                                           # through the sampling entry point, top_k = 1 == arg-max
            class ArgMax:                                           # nucleus with top_k = 1 keeps the arg-max only (nucleus.rs:77-89)
                top_p, top_k, temperature = 0.5, 1, 1.0

                def adjustments(self):
                    return {}
            inp = rt.RnnInput([rt.RnnInputBatch([toks[b]] if b in act else []) for b in range(3)])
            _, out = eng.infer_sample(inp, [ArgMax() if b in act else None for b in range(3)], [0.3] * 3)
            for b in range(3):
                if b in act:
                    assert out[b] is not None and out[b][0] == int(np.argmax(want[b])), f"step {step} slot {b}"
                else:
                    assert out[b] is None
        else:
            inp = rt.RnnInput([rt.RnnInputBatch([toks[b]] if b in act else []) for b in range(3)])
            _, outs = eng.infer(inp)
            for b in range(3):
                if b in act:
                    assert np.abs(outs[b][-1] - want[b]).max() <= tol(rt.Precision.Fp32, want[b]), f"step {step} slot {b}"
                else:
                    assert len(outs[b]) == 0
    first = [int(rng.integers(1, V)) for _ in range(3)]
    ids, _ = eng.decode_greedy(first, 5)                            # uploads its own plan: the cache must notice
    cur = list(first)
    for s in range(5):
        for b in range(3):
            lg = ref.forward([cur[b]], states[b])[-1]
            cur[b] = int(np.argmax(lg))
            assert int(ids[s, b]) == cur[b], f"greedy step {s} slot {b}"
    toks = [int(rng.integers(1, V)) for _ in range(3)]
    _, outs = eng.infer(rt.RnnInput([rt.RnnInputBatch([toks[b]]) for b in range(3)]))
    for b in range(3):
        want = ref.forward([toks[b]], states[b])[-1]
        assert np.abs(outs[b][-1] - want).max() <= tol(rt.Precision.Fp32, want)
        assert np.abs(eng.state.back(b) - states[b]).max() <= tol(rt.Precision.Fp32, states[b])
    eng.close()


@pytest.mark.parametrize("Dd", [32, 96])
@pytest.mark.parametrize("chunk", [12, 64], ids=["short-rows", "long-rows"])
def test_v6_decay_lora_dims_other_than_64_and_128(Dd, chunk):
    """The loader accepts any decay-LoRA dim Dd <= 128 with Dd % 4 == 0; the chunked WKV kernels are compiled for 64 and 128 only.  Steps whose
    sequences have <= 8 rows each (chunk 12 over three slots) and longer ones (chunk 64) must both fall back to the generic kernel for other
    dims (round-5 advisor finding: the short form used to take the <6, 128, 8> instance and read D2 / td rows with the wrong extent)."""
    t = R.synth_named("v6-small")
    rng = np.random.default_rng(77 + Dd)
    C = t["blocks.0.att.time_decay"].shape[-1]
    for l in range(3):
        t[f"blocks.{l}.att.time_decay_w1"] = (rng.standard_normal((Dd, C), dtype=np.float32) * np.float32(0.05)).astype(np.float16)
        t[f"blocks.{l}.att.time_decay_w2"] = (rng.standard_normal((C, Dd), dtype=np.float32) * np.float32(0.05)).astype(np.float16)
    eng = rt.ModelBuilder(R.st_serialize(t)).build(max_batch=3, token_chunk_size=chunk, precision=rt.Precision.Fp32)
    ref = R.RwkvRef(t)
    ps = [prompt(ref, 50 + s, 21 + 4 * s) for s in range(3)]
    got = run_prompts(eng, ps)
    for b in range(3):
        s = ref.init_state()
        want = ref.forward(ps[b], s)[-1]
        assert np.abs(got[b][0] - want).max() <= tol(rt.Precision.Fp32, want)
        assert np.abs(eng.state.back(b) - s).max() <= tol(rt.Precision.Fp32, s)
    eng.close()
