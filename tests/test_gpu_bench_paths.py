"""GPU (-m gpu): parity of the code paths `bench.py` actually times, at the widths BASELINE.json names.

  * 32-slot decode of a full-width (C=2560, F=8960, V=65536) Int8 V6 model: `rwkv_infer` logits of every slot and the
    token ids of the device-resident greedy loop (`rwkv_decode_greedy`, the bench's timed call) against the oracle;
  * every prefill tile shape the engine can pick at real widths (forced through RWKV_TILE_SHAPE), fp16 / Int8 / NF4,
    logits + state against the oracle, and the shape the planner picks by itself for a 1024-row step of 7B shapes
    (C=4096, Dm=64, Dd=128: `wkv_chunk_kernel<6,128>`, the GLDS 128x64 tile);
  * two-layer full-width models of BASELINE configs #1 (V5 C=1024), #2 (V6 C=2048), #5 (V6 C=4096), V7-2.9B NF4 at
    32 slots (config #4's decode shape);
  * >= 256 greedy steps on the small models.

The oracle here is the lock-step form `RwkvRefBatch` (pinned to the per-token restatement by tests/test_oracle.py).
Tolerances as in test_gpu_parity.py: 1e-3 * max(1, |ref|_inf) in Precision::Fp16 mode, ids identical."""
import os

import numpy as np
import pytest

from ai00_server_amd import runtime as rt
from oracle import rwkv_ref as R

pytestmark = pytest.mark.gpu
FP16_TOL = 1e-3


def tol(want):
    return FP16_TOL * max(1.0, float(np.abs(want).max()))


def engine(st, quant, B, chunk, prec=rt.Precision.Fp16):
    return rt.ModelBuilder(st).quant(quant[0], rt.Quant(quant[1])).build(max_batch=B, token_chunk_size=chunk, precision=prec)


def feed(eng, prompts, option=rt.RnnOption.Last):
    B = eng.max_batch
    inp = rt.RnnInput([rt.RnnInputBatch(list(prompts[b]) if b < len(prompts) else [], option) for b in range(B)])
    rows = [[] for _ in range(B)]
    while inp.num_token() > 0:
        inp, outs = eng.infer(inp)
        for b, o in enumerate(outs):
            rows[b].extend(list(o))
    return rows


def prompts_for(V, B, base, lens):
    return [[t % V for t in R.synth_prompt(base + b, lens[b % len(lens)])] for b in range(B)]


def check_states(eng, states, slots):
    for b in slots:
        back = eng.state.back(b)
        assert np.abs(back - states[b]).max() <= tol(states[b]), f"state of slot {b}"


def decode_both_ways(eng, rb, states, first, n_steps):
    """`n_steps` decode steps over all slots: (1) rwkv_infer, teacher-forced with the oracle's ids, logits of every slot
    within tolerance and arg-max identical at every step; (2) state restored, the same steps through rwkv_decode_greedy
    (ids stay on the device): ids identical to the oracle's."""
    B = len(first)
    snaps = [eng.state.back(b) for b in range(B)]
    ref_states = states.copy()
    want_ids, _ = rb.greedy_batch(first, n_steps, ref_states)
    st2 = states.copy()
    cur = [int(t) for t in first]
    for s in range(n_steps):
        want = rb.step(cur, st2)
        inp = rt.RnnInput([rt.RnnInputBatch([cur[b]] if b < B else [], rt.RnnOption.Last) for b in range(eng.max_batch)])
        _, outs = eng.infer(inp)
        for b in range(B):
            got = outs[b][-1]
            assert np.abs(got - want[b]).max() <= tol(want[b]), f"step {s} slot {b}"
            assert int(np.argmax(got)) == int(want_ids[s, b]), f"arg-max, step {s} slot {b}"
        cur = [int(t) for t in want_ids[s]]
    check_states(eng, st2, range(B))
    for b in range(B):
        eng.state.load(snaps[b], b)
    toks, _ = eng.decode_greedy([int(t) for t in first], n_steps)
    np.testing.assert_array_equal(np.asarray(toks, dtype=np.int64)[:, :B], want_ids)
    check_states(eng, ref_states, range(B))


def test_bench_shape_32_slots_int8_infer_and_device_greedy():
    """The headline configuration's own path: 32 slots in one call (run.rs:1121-1157 hands `max_batch` slots to one
    infer), Int8 on every layer, full width and vocabulary, two layers."""
    tens = R.synth_checkpoint(6, 2, 2560, 8960, 65536, seed=11)
    rb = R.RwkvRefBatch(tens, 2, R.QUANT_INT8)
    B = 32
    eng = engine(R.st_serialize(tens), (2, 1), B, 512)
    ps = prompts_for(65536, B, 200, [3, 5, 4, 6, 2])
    states = rb.init_states(B)
    want = rb.prefill(ps, states)
    rows = feed(eng, ps)                                                  # one ragged 128-row step: the four-tile decode GEMM
    for b in range(B):
        assert np.abs(rows[b][-1] - want[b]).max() <= tol(want[b]), f"prefill slot {b}"
    first = np.argmax(want, axis=1)
    decode_both_ways(eng, rb, states, first, 16)
    eng.close()


def test_v7_nf4_32_slots_infer_and_device_greedy():
    """BASELINE config #4's engine (V7-2.9B shapes, NF4 on every layer) at 32 slots, two layers."""
    tens = R.synth_checkpoint(7, 2, 2560, 10240, 65536, seed=17)
    rb = R.RwkvRefBatch(tens, 2, R.QUANT_NF4)
    B = 32
    eng = engine(R.st_serialize(tens), (2, 2), B, 512)
    ps = prompts_for(65536, B, 300, [4, 2, 5])
    states = rb.init_states(B)
    want = rb.prefill(ps, states)
    rows = feed(eng, ps)
    for b in range(B):
        assert np.abs(rows[b][-1] - want[b]).max() <= tol(want[b]), f"prefill slot {b}"
    decode_both_ways(eng, rb, states, np.argmax(want, axis=1), 12)
    eng.close()


@pytest.fixture(scope="module")
def wide3b():
    """2 layers of V6-3B shapes with a small vocabulary, 4 slots x 160-token prompts: oracle results shared by the shape tests."""
    tens = R.synth_checkpoint(6, 2, 2560, 8960, 1024, seed=23)
    st = R.st_serialize(tens)
    ps = prompts_for(1024, 4, 400, [160, 131, 160, 97])
    out = {}
    for qt in (0, 1, 2):
        rb = R.RwkvRefBatch(tens, 2 if qt else 0, qt)
        states = rb.init_states(4)
        out[qt] = (rb.prefill(ps, states), states)
    return st, ps, out


@pytest.mark.parametrize("shape,qt", [(3, 0), (3, 1), (3, 2), (7, 0), (7, 1), (7, 2), (4, 1), (0, 1), (1, 1), (2, 1), (5, 1),
                                       (6, 1), (8, 1), (9, 1), (9, 0), (10, 0), (10, 1), (10, 2), (11, 0), (11, 1), (11, 2), (12, 0), (12, 1), (12, 2)])
def test_every_prefill_tile_shape_at_3b_width(wide3b, shape, qt):
    """gemm_tile_kernel in each of its ten shapes, the pipelined kernel on 128x128 (shape 10) and 128x64 tiles (shape 11) and the software-pipelined
    hi + lo kernel (shape 12: the promoted time-mix launch of the default Precision::Fp16; a forced pipelined shape a launch's operand form
    cannot take falls back to the 64x64 shape) — rwkv_engine.cpp picks by grid size and operand form, the rest are reachable through
    RWKV_TILE_SHAPE — on a 548-row ragged step of 3B-wide matrices, plus `wkv_chunk_kernel<6,64>` at H = 40."""
    st, ps, ref = wide3b
    want, states = ref[qt]
    os.environ["RWKV_TILE_SHAPE"] = str(shape)
    try:
        eng = engine(st, (2 if qt else 0, qt), 4, 1024)
        rows = feed(eng, ps)
    finally:
        os.environ.pop("RWKV_TILE_SHAPE", None)
    for b in range(4):
        assert np.abs(rows[b][-1] - want[b]).max() <= tol(want[b]), f"slot {b}"
    check_states(eng, states, range(4))
    eng.close()


def test_pipelined_tile_kernel_is_bit_identical_to_the_64x64_shape_over_repeated_runs(wide3b):
    """Race screen for the counted-vmcnt K loops of shapes 10, 11 and 12 (inline-asm loads, LDS-DMA ring, raw barriers; in the default precision the
    time-mix launch reads hi + lo operands: the 64x64 hi + lo form under shape 4, the software-pipelined kernel under 10 / 11 / 12): all kernels add the
    k-steps of a row in the same order into fp32 MFMA accumulators, so logits and state must be BIT-identical to the 64x64
    shape's, on every one of several runs (a landed-too-late tile would show up as a differing run)."""
    st, ps, _ = wide3b
    outs = {}
    os.environ["RWKV_TILE_KSPLIT"] = "0"              # K copies of the linear launches change the summation order, by design
    for shape in (4, 10, 11, 12):
        os.environ["RWKV_TILE_SHAPE"] = str(shape)
        try:
            for qt in (0, 1, 2):
                eng = engine(st, (2 if qt else 0, qt), 4, 1024)
                runs = []
                for rep in range(4):
                    for b in range(4):
                        eng.state.load(eng.state.init(), b)
                    rows = feed(eng, ps)
                    runs.append((np.stack([rows[b][-1] for b in range(4)]), np.stack([eng.state.back(b) for b in range(4)])))
                outs[(shape, qt)] = runs
                eng.close()
        finally:
            os.environ.pop("RWKV_TILE_SHAPE", None)
    os.environ.pop("RWKV_TILE_KSPLIT", None)
    for qt in (0, 1, 2):
        ref_l, ref_s = outs[(4, qt)][0]
        for shape in (4, 10, 11, 12):
            for rep, (lg, stt) in enumerate(outs[(shape, qt)]):
                assert np.array_equal(lg, ref_l), f"logits differ: shape {shape} quant {qt} run {rep}"
                assert np.array_equal(stt, ref_s), f"state differs: shape {shape} quant {qt} run {rep}"
    # Precision::Fp32: EVERY launch reads hi + lo operands — Wo, Fk / Fr, Fv and (at this vocabulary) the head run the software-pipelined kernel
    # too, the linear ones as a single K copy here.  Int8, three runs against the 64x64 shape's bits.
    os.environ["RWKV_TILE_KSPLIT"] = "0"
    res = {}
    try:
        for shape in (4, 12):
            os.environ["RWKV_TILE_SHAPE"] = str(shape)
            eng = engine(st, (2, 1), 4, 1024, prec=rt.Precision.Fp32)
            res[shape] = []
            for rep in range(3 if shape != 4 else 1):
                for b in range(4):
                    eng.state.load(eng.state.init(), b)
                rows = feed(eng, ps)
                res[shape].append((np.stack([rows[b][-1] for b in range(4)]), np.stack([eng.state.back(b) for b in range(4)])))
            eng.close()
    finally:
        os.environ.pop("RWKV_TILE_SHAPE", None)
        os.environ.pop("RWKV_TILE_KSPLIT", None)
    for shape in (12,):
        for rep, (lg, stt) in enumerate(res[shape]):
            assert np.array_equal(lg, res[4][0][0]) and np.array_equal(stt, res[4][0][1]), f"Precision::Fp32: shape {shape} differs from the 64x64 shape, run {rep}"


def test_config5_7b_width_1024_row_prefill_and_8_slot_decode():
    """BASELINE config #5 shapes (C=4096, F=14336, Dm=64, Dd=128, fp16), two layers: an 8 x 128 = 1024-row prefill step
    (the planner's own choice: the pipelined 128x128 tile kernel on the 1032-tile r/k/v/g/decay launch, 64x64 tiles on the
    rest, `wkv_chunk_kernel<6,128>`), then 8-slot decode both ways."""
    tens = R.synth_checkpoint(6, 2, 4096, 14336, 2048, seed=29)
    rb = R.RwkvRefBatch(tens)
    B = 8
    eng = engine(R.st_serialize(tens), (0, 0), B, 1024)
    ps = prompts_for(2048, B, 500, [128])
    states = rb.init_states(B)
    want = rb.prefill(ps, states)
    rows = feed(eng, ps)
    for b in range(B):
        assert np.abs(rows[b][-1] - want[b]).max() <= tol(want[b]), f"prefill slot {b}"
    check_states(eng, states, range(B))
    decode_both_ways(eng, rb, states, np.argmax(want, axis=1), 8)
    eng.close()


@pytest.mark.parametrize("cfg", [(6, 2, 2048, 7168, 4096, 1), (5, 2, 1024, 3584, 4096, 1), (6, 2, 2560, 8960, 4096, 1),
                                 (6, 2, 2560, 8960, 4096, 8)], ids=["cfg2-v6-1.6b-b1", "cfg1-v5-0.4b-b1", "v6-3b-int8-b1", "v6-3b-int8-b8"])
def test_full_width_decode_small_batches(cfg):
    """Configs #2 / #1 at their own widths (fp16, one slot) and the 3B Int8 engine at 1 and 8 slots (the sweep points of
    the bench): single-token steps run the LayerNorm-prologue kernels, which the 32-slot test does not reach."""
    ver, L, C, F, V, B = cfg
    tens = R.synth_checkpoint(ver, L, C, F, V, seed=31)
    q = (L, 1) if C == 2560 else (0, 0)
    rb = R.RwkvRefBatch(tens, *q)
    eng = engine(R.st_serialize(tens), q, B, 64)
    ps = prompts_for(V, B, 600, [7, 3, 9])
    states = rb.init_states(B)
    want = rb.prefill(ps, states)
    rows = feed(eng, ps)
    for b in range(B):
        assert np.abs(rows[b][-1] - want[b]).max() <= tol(want[b])
    decode_both_ways(eng, rb, states, np.argmax(want, axis=1), 24)
    eng.close()


@pytest.mark.parametrize("name,quant", [("v5-small", (0, 0)), ("v6-small", (3, 1)), ("v7-small", (3, 2))])
def test_256_greedy_steps_identical_ids(name, quant):
    """identical greedy ids for >= 256 steps, through rwkv_infer and through the device-resident loop."""
    tens = R.synth_named(name)
    ref = R.RwkvRef(tens, *quant)
    V = ref.info.num_vocab
    p = [t % V for t in R.synth_prompt(70, 12)]
    want, _ = ref.greedy(p, 257)
    eng = engine(R.st_serialize(tens), quant, 2, 16)
    lg = feed(eng, [p])[0][-1]
    got = []
    for _ in range(256):
        t = int(np.argmax(lg))
        got.append(t)
        _, outs = eng.infer(rt.RnnInput([rt.RnnInputBatch([t]), rt.RnnInputBatch()]))
        lg = outs[0][-1]
    assert got == want[:256]
    eng.state.load(eng.state.init(), 0)
    feed(eng, [p])
    toks, _ = eng.decode_greedy([want[0]], 256)
    assert [int(x) for x in toks[:, 0]] == want[1:257]
    eng.close()
