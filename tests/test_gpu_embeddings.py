"""GPU (-m gpu): parity of the state-only path (`RWKV_OPTION_NONE`, runtime.RnnOption.NoOutput) that `bench.py`'s
embeddings leg times, at the widths BASELINE.json names (configs #3 and #4), through the C ABI.

`/embeddings` (docs/doc-api/openai.md:376-437) prefills a document and reads back ONE layer's WKV rows; no logits row
is ever consumed, so the engine skips the final LayerNorm, the head GEMM and the logits copy for slots that ask for
nothing.  What is compared here:

  * 32 x 256-token documents, V6-3B Int8 width and V7-2.9B NF4 width (two layers), `token_chunk_size` 2048 and 256:
    the whole state slab of every slot and `rwkv_state_back_layer` of the last layer against `RwkvRefBatch` fed the same
    documents token by token;
  * one step that mixes NoOutput, Last and Full slots; NoOutput followed by Last on the same slot;
  * the embedding of a document is the same whatever option prefilled it (NoOutput == Last, bit for bit).

Tolerance: 1e-3 * max(1, |ref|_inf) (Precision::Fp16), as in test_gpu_parity.py."""
import numpy as np
import pytest

from ai00_server_amd import runtime as rt
from oracle import rwkv_ref as R

pytestmark = pytest.mark.gpu
FP16_TOL = 1e-3
B, DOC = 32, 256


def tol(want):
    return FP16_TOL * max(1.0, float(np.abs(want).max()))


def docs_for(V, n=B, doc=DOC, base=100):
    return [[t % V for t in R.synth_prompt(base + b, doc)] for b in range(n)]


def oracle_states(rb, docs):
    """Lock-step prefill without logits (the head is not on this path): states [B, L, N+2, C]."""
    states = rb.init_states(len(docs))
    for t in range(max(len(d) for d in docs)):
        act = [b for b in range(len(docs)) if t < len(docs[b])]
        sub = states[act].copy()
        rb.step([docs[b][t] for b in act], sub, want_logits=False)
        states[act] = sub
    return states


def feed(eng, prompts, options):
    """Feed prompts with a per-slot option until everything is consumed; returns the rows every slot emitted."""
    nb = eng.max_batch
    inp = rt.RnnInput([rt.RnnInputBatch(list(prompts[b]) if b < len(prompts) else [], options[b] if b < len(options) else rt.RnnOption.Last)
                       for b in range(nb)])
    rows = [[] for _ in range(nb)]
    calls = 0
    while inp.num_token() > 0:
        inp, outs = eng.infer(inp)
        calls += 1
        for b, o in enumerate(outs):
            rows[b].extend(list(o))
    return rows, calls


@pytest.fixture(scope="module")
def v6_int8():
    tens = R.synth_checkpoint(6, 2, 2560, 8960, 4096, seed=41)
    rb = R.RwkvRefBatch(tens, 2, R.QUANT_INT8)
    docs = docs_for(4096)
    return R.st_serialize(tens), rb, docs, oracle_states(rb, docs)


@pytest.fixture(scope="module")
def v7_nf4():
    tens = R.synth_checkpoint(7, 2, 2560, 10240, 4096, seed=43)
    rb = R.RwkvRefBatch(tens, 2, R.QUANT_NF4)
    docs = docs_for(4096, base=300)
    return R.st_serialize(tens), rb, docs, oracle_states(rb, docs)


def check_embedding_job(st, quant, docs, want, chunk):
    eng = rt.ModelBuilder(st).quant(2, rt.Quant(quant)).build(max_batch=B, token_chunk_size=chunk, precision=rt.Precision.Fp16)
    rows, calls = feed(eng, docs, [rt.RnnOption.NoOutput] * B)
    assert all(len(r) == 0 for r in rows), "a state-only slot produced a logits row"
    assert calls == (B * DOC + chunk - 1) // chunk
    N = eng.info.head_size
    layer = eng.info.num_layer - 1
    for b in range(B):
        back = eng.state.back(b)
        assert np.abs(back - want[b]).max() <= tol(want[b]), f"state slab of slot {b}"
        emb = eng.state.embed(layer, b)
        np.testing.assert_array_equal(emb, back[layer, 1:1 + N])                   # the layer slice IS the slab's rows
        assert np.abs(emb - want[b, layer, 1:1 + N]).max() <= tol(want[b, layer, 1:1 + N]), f"embedding of slot {b}"
    # the same documents prefilled with Last give the same embeddings bit for bit (the head is the only difference)
    ref_emb = [eng.state.embed(layer, b) for b in range(B)]
    zero = eng.state.init()
    for b in range(B):
        eng.state.load(zero, b)
    rows, _ = feed(eng, docs, [rt.RnnOption.Last] * B)
    assert all(len(r) == 1 for r in rows)
    for b in range(B):
        np.testing.assert_array_equal(eng.state.embed(layer, b), ref_emb[b])
    eng.close()


@pytest.mark.parametrize("chunk", [2048, 256])
def test_state_only_prefill_v6_3b_int8_width(v6_int8, chunk):
    """BASELINE config #3's engine: 32 x 256-token documents with RWKV_OPTION_NONE (the embeddings leg of bench.py)."""
    st, rb, docs, want = v6_int8
    check_embedding_job(st, 1, docs, want, chunk)


@pytest.mark.parametrize("chunk", [2048, 256])
def test_state_only_prefill_v7_2_9b_nf4_width(v7_nf4, chunk):
    """BASELINE config #4: V7-2.9B shapes, NF4 on every layer, `/embeddings` job at SURVEY 8(d)'s chunk of 256 and at 2048."""
    st, rb, docs, want = v7_nf4
    check_embedding_job(st, 2, docs, want, chunk)


@pytest.mark.parametrize("ver,quant", [(6, 1), (7, 2), (5, 0)])
def test_one_step_mixes_none_last_and_full_slots(ver, quant):
    """Slots of one `infer` call ask for different things (run.rs:716, 819 + the NoOutput extension): a step with
    NoOutput, Last and Full rows emits exactly the rows asked for, with the oracle's logits, and every slot's state —
    the silent ones included — follows the oracle.  Then the NoOutput slot continues with Last."""
    C, F = (2560, 8960) if ver != 5 else (1024, 3584)
    F = 10240 if ver == 7 else F
    V = 2048
    tens = R.synth_checkpoint(ver, 2, C, F, V, seed=47 + ver)
    ql = 2 if quant else 0
    rb = R.RwkvRefBatch(tens, ql, quant)
    ref = R.RwkvRef(tens, ql, quant)
    nb = 6
    lens = [9, 5, 12, 1, 7, 3]
    opts = [rt.RnnOption.NoOutput, rt.RnnOption.Last, rt.RnnOption.Full, rt.RnnOption.NoOutput, rt.RnnOption.Last, rt.RnnOption.Full]
    ps = [[t % V for t in R.synth_prompt(700 + b, lens[b])] for b in range(nb)]
    eng = rt.ModelBuilder(R.st_serialize(tens)).quant(ql, rt.Quant(quant)).build(max_batch=nb, token_chunk_size=64, precision=rt.Precision.Fp16)
    rows, calls = feed(eng, ps, opts)
    assert calls == 1                                                      # 37 rows: ONE step carries all three kinds
    states = []
    for b in range(nb):
        s = ref.init_state()
        full = ref.forward(ps[b], s, full=True)
        states.append(s)
        if opts[b] == rt.RnnOption.NoOutput:
            assert len(rows[b]) == 0
        elif opts[b] == rt.RnnOption.Last:
            assert len(rows[b]) == 1
            assert np.abs(rows[b][0] - full[-1]).max() <= tol(full[-1]), f"Last row of slot {b}"
        else:
            got = np.stack(rows[b])
            assert got.shape == full.shape
            assert np.abs(got - full).max() <= tol(full), f"Full rows of slot {b}"
        back = eng.state.back(b)
        assert np.abs(back - s).max() <= tol(s), f"state of slot {b}"
    # NoOutput, then Last on the same slots: the logits continue from the silently advanced state
    more = [[t % V for t in R.synth_prompt(800 + b, 4)] for b in range(nb)]
    sel = [0, 3]
    ps2 = [more[b] if b in sel else [] for b in range(nb)]
    rows2, _ = feed(eng, ps2, [rt.RnnOption.Last] * nb)
    for b in sel:
        want = ref.forward(more[b], states[b])[-1]
        assert len(rows2[b]) == 1
        assert np.abs(rows2[b][0] - want).max() <= tol(want), f"Last after NoOutput, slot {b}"
    # a step in which NO slot emits (the head is skipped entirely) next to one in which one does: same state either way
    snap = [eng.state.back(b) for b in range(nb)]
    ps3 = [[t % V for t in R.synth_prompt(900 + b, 2)] for b in range(nb)]
    feed(eng, ps3, [rt.RnnOption.NoOutput] * nb)
    silent = [eng.state.back(b) for b in range(nb)]
    for b in range(nb):
        eng.state.load(snap[b], b)
    feed(eng, ps3, [rt.RnnOption.NoOutput] * (nb - 1) + [rt.RnnOption.Last])
    for b in range(nb):
        np.testing.assert_array_equal(eng.state.back(b), silent[b])
    eng.close()


@pytest.mark.parametrize("ver,quant", [(6, 1), (7, 2)])
def test_state_job_with_slot_turnover_and_async_read_back(ver, quant):
    """harness.StateJob — what bench.py's embeddings leg times: more documents than slots, RAGGED lengths (a slot that finishes takes
    the next document while the others are mid-flight), embeddings leaving through rwkv_state_back_layer_async into pinned memory while
    the next step runs.  Every embedding against the oracle's state of that document prefilled alone; the asynchronous read-back
    against the blocking one on a slot that is then overwritten."""
    from ai00_server_amd.harness import StateJob
    C, F, V = 2560, (8960 if ver == 6 else 10240), 2048
    tens = R.synth_checkpoint(ver, 2, C, F, V, seed=61 + ver)
    ref = R.RwkvRef(tens, 2, quant)
    nb = 4
    eng = rt.ModelBuilder(R.st_serialize(tens)).quant(2, rt.Quant(quant)).build(max_batch=nb, token_chunk_size=64, precision=rt.Precision.Fp16)
    lens = [40, 7, 33, 1, 64, 12, 0, 25, 3, 50, 18]
    docs = [[t % V for t in R.synth_prompt(1500 + i, n)] for i, n in enumerate(lens)]
    layer = 1
    job = StateJob(eng, layer)
    emb, calls = job.run(docs)
    assert emb.shape == (len(docs), 64, C)
    for i, d in enumerate(docs):
        s = ref.init_state()
        ref.forward(d if len(d) else [0], s)
        want = s[layer, 1:65]
        assert np.abs(emb[i] - want).max() <= tol(want), f"document {i} ({lens[i]} tokens)"
    # asynchronous == blocking, and the slot may be overwritten as soon as the call returns
    feed(eng, [docs[4]], [rt.RnnOption.NoOutput])
    sync_copy = eng.state.embed(layer, 0)
    arena = rt.PinnedArena((64, C))
    eng.state.embed_async(layer, 0, arena.array)
    eng.state.load(eng.state.init(), 0)                                    # ordered behind the pack by the engine
    feed(eng, [docs[9]], [rt.RnnOption.NoOutput])
    eng.state.sync()
    np.testing.assert_array_equal(arena.array, sync_copy)
    with pytest.raises(rt.RwkvError):                                      # pageable memory is refused, not silently staged
        eng.state.embed_async(layer, 0, np.empty((64, C), np.float32))
    # the rows must END inside the pinned block they start in (advisor, round 4): an offset that leaves less than 64 x C floats is an
    # RWKV_ERR_INVALID, not a DMA past the block; the last position that fits is fine
    big = rt.PinnedArena((2, 64, C))
    flat = big.array.reshape(-1)
    with pytest.raises(rt.RwkvError):
        eng.state.embed_async(layer, 0, flat[64 * C + 1:])
    with pytest.raises(rt.RwkvError):
        eng.state.embed_async(layer, 0, flat[2 * 64 * C - 8:])
    eng.state.embed_async(layer, 0, flat[64 * C:])
    eng.state.sync()
    np.testing.assert_array_equal(big.array[1], eng.state.embed(layer, 0))
    big.close()
    arena.close()
    job.close()
    eng.close()


@pytest.mark.parametrize("ver,quant", [(6, 1), (7, 2)])
def test_cpp_embed_job_equals_the_python_state_job(ver, quant, tmp_path):
    """harness/embed_job.cpp — rwkv::Scheduler::embed_documents (documents as state-only requests through queue / step, slot turnover,
    rwkv::State::embed_async into a rwkv::PinnedBuffer) over the real engine: the same documents give the embeddings of harness.StateJob
    (itself checked against the oracle above) within the parity bound."""
    import struct
    import subprocess
    from ai00_server_amd import build as B
    from ai00_server_amd.harness import StateJob
    B.build_harness(verbose=False)
    C, F, V = 2560, (8960 if ver == 6 else 10240), 2048
    tens = R.synth_checkpoint(ver, 2, C, F, V, seed=61 + ver)
    blob = R.st_serialize(tens)
    model = tmp_path / "m.st"
    model.write_bytes(blob)
    lens = [40, 7, 33, 1, 64, 12, 0, 25, 3, 50, 18, 130, 5]
    docs = [[t % V for t in R.synth_prompt(1700 + i, n)] for i, n in enumerate(lens)]
    raw = struct.pack("<I", len(docs)) + b"".join(struct.pack(f"<I{len(d)}I", len(d), *d) for d in docs)
    (tmp_path / "docs.bin").write_bytes(raw)
    nb, chunk, layer = 4, 64, 1
    r = subprocess.run([B.EMBED_BIN, str(model), "2", str(quant), str(nb), str(chunk), str(layer), str(tmp_path / "docs.bin"), str(tmp_path / "out.bin"), "2"],
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    words = r.stdout.split()
    steps = int(words[words.index("steps") + 1])
    got = np.fromfile(tmp_path / "out.bin", np.float32).reshape(len(docs), 64, C)
    eng = rt.ModelBuilder(blob).quant(2, rt.Quant(quant)).build(max_batch=nb, token_chunk_size=chunk, precision=rt.Precision.Fp16)
    job = StateJob(eng, layer)
    emb, calls = job.run(docs)
    # the scheduler hands out idle slots in the reference's order (longest idle, the LAST one on ties: run.rs:505-530), StateJob in slot
    # order: the documents meet different neighbours in a step, so the split of token_chunk_size between slots — and with it which GEMM
    # kernel reads a given token — may differ.  Same bound as against the oracle; the step count may differ by the odd remainder step.
    assert np.isfinite(got).all()
    for i in range(len(docs)):
        assert np.abs(got[i] - emb[i]).max() <= tol(emb[i]), f"document {i} ({lens[i]} tokens)"
    ref = R.RwkvRef(tens, 2, quant)
    for i, d in enumerate(docs):                                            # and against the oracle directly
        st = ref.init_state()
        ref.forward(d if len(d) else [0], st)
        assert np.abs(got[i] - st[layer, 1:65]).max() <= tol(st[layer, 1:65]), f"document {i} vs the oracle"
    total = sum(max(1, n) for n in lens)
    assert -(-total // chunk) <= steps <= calls + 2 and abs(steps - calls) <= 2
    job.close()
    eng.close()
