"""Inputs produced by the reference's OWN tooling (`assets/scripts/convert_safetensors.py`), not by this repo's writer.

`tests/golden/converted_v{5,6,7}.st` and `converted_lora.st` are the bytes that script wrote (through the real
`safetensors` library) for small BlinkDL-layout checkpoints; `tests/golden/make_converted.py` documents how and
rebuilds the original tensors from their seeds.  Checked here:

  CPU   * the committed bytes ARE the converter's output (re-run and compared where /root/reference exists);
        * `Loader::info` (lib.rs:587, `rwkv_model_info_from_st`) reads version / sizes off the real file;
        * the converter applied to the inverse-mapped checkpoint gives back the ai00 layout the oracle assumes
          (names, transposes, fp16: convert_safetensors.py:36-47, 62-72, 96-101);
        * oracle logits on the converted file == the literal BlinkDL functions on the ORIGINAL tensors.
  GPU   * the engine loaded from the converter's bytes (+ the converted LoRA file) against the same literal evaluation."""
import os
import sys
import tempfile

import numpy as np
import pytest

from ai00_server_amd import runtime as rt
from oracle import rwkv_ref as R
from tests.blinkdl_literal import Literal

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "golden"))
import make_converted as MC  # noqa: E402

NAMES = ["v5", "v6", "v7"]


def fixture_bytes(name):
    with open(os.path.join(HERE, "golden", f"converted_{name}.st"), "rb") as f:
        return f.read()


@pytest.fixture(scope="module")
def sources():
    return MC.sources()


@pytest.mark.skipif(not os.path.exists(MC.CONVERTER), reason="the reference checkout is not on this machine")
@pytest.mark.parametrize("name", NAMES + ["lora"])
def test_committed_fixture_is_the_reference_converters_output(sources, name):
    with tempfile.TemporaryDirectory() as d:
        assert MC.run_converter(sources[name], d) == fixture_bytes(name)


@pytest.mark.parametrize("name", NAMES)
def test_loader_info_on_a_file_written_by_the_reference_tooling(built_lib, name):
    ver, L, C, F, V, _ = MC.CASES[name]
    i = rt.Loader.info(fixture_bytes(name))
    assert (int(i.version), i.num_layer, i.num_emb, i.num_hidden, i.num_vocab, i.num_head, i.head_size) == (ver, L, C, F, V, C // 64, 64)


@pytest.mark.parametrize("name", NAMES)
def test_converter_output_is_the_layout_the_oracle_assumes(name):
    ver, L, C, F, V, seed = MC.CASES[name]
    want = R.synth_checkpoint(ver, L, C, F, V, seed=seed)
    got = R.st_deserialize(fixture_bytes(name))
    assert set(got) == set(want)
    for k in want:
        assert got[k].dtype == np.float16 and got[k].shape == want[k].shape, k
        np.testing.assert_array_equal(got[k], want[k], err_msg=k)


def test_converted_lora_file_layout():
    got = R.st_deserialize(fixture_bytes("lora"))
    assert set(got) == {"blocks.0.att.key.lora.0", "blocks.0.att.key.lora.1", "blocks.1.ffn.value.lora.0", "blocks.1.ffn.value.lora.1"}
    assert got["blocks.0.att.key.lora.0"].shape == (128, 8) and got["blocks.0.att.key.lora.1"].shape == (128, 8)   # [in, r], [out, r]


def literal_logits(src, tokens):
    lit = Literal(src, layout="blinkdl")
    ls = lit.new_state()
    return np.stack([lit.forward(t, ls) for t in tokens]), ls


@pytest.mark.parametrize("name", NAMES)
def test_oracle_on_converted_file_matches_literal_on_original_tensors(sources, name):
    ref = R.RwkvRef(R.st_deserialize(fixture_bytes(name)))
    toks = [t % ref.info.num_vocab for t in R.synth_prompt(21, 14)]
    want, _ = literal_logits(sources[name], toks)
    st = ref.init_state()
    got = ref.forward(toks, st, full=True)
    np.testing.assert_allclose(got, want, rtol=0, atol=5e-5)


def blended_source(src, lora_src, alpha):
    """W += alpha * B A on the ORIGINAL tensors (`LoraBlend::full(alpha)`, lib.rs:466-482): lora_B [out, r] @ lora_A [r, in]."""
    out = dict(src)
    for k in lora_src:
        if k.endswith("lora_A"):
            base = k[:-len(".lora_A")]
            A, B = lora_src[k].astype(np.float32), lora_src[base + ".lora_B"].astype(np.float32)
            out[base + ".weight"] = (src[base + ".weight"].astype(np.float32) + np.float32(alpha) * (B @ A)).astype(np.float16)
    return out


@pytest.mark.gpu
@pytest.mark.parametrize("name", NAMES)
@pytest.mark.parametrize("with_lora", [False, True], ids=["plain", "lora"])
def test_engine_loads_the_reference_converters_file(sources, name, with_lora):
    """`ModelBuilder::new(st).lora(..).build_vN()` (lib.rs:484-492) on the converter's own bytes, Precision::Fp32:
    logits of every prompt token (Full) against the literal BlinkDL functions on the original, un-transposed tensors."""
    alpha = 0.75
    b = rt.ModelBuilder(fixture_bytes(name))
    src = sources[name]
    if with_lora:
        b = b.lora(fixture_bytes("lora"), alpha)
        src = blended_source(src, sources["lora"], alpha)
    eng = b.build(max_batch=2, token_chunk_size=8, precision=rt.Precision.Fp32)
    V = eng.info.num_vocab
    toks = [t % V for t in R.synth_prompt(22, 19)]
    want, ls = literal_logits(src, toks)
    inp = rt.RnnInput([rt.RnnInputBatch(list(toks), rt.RnnOption.Full), rt.RnnInputBatch()])
    rows = []
    while inp.num_token() > 0:
        inp, outs = eng.infer(inp)
        rows.extend(list(outs[0]))
    got = np.stack(rows)
    assert got.shape == want.shape
    assert np.abs(got - want).max() <= 1e-4 * max(1.0, float(np.abs(want).max()))
    # state conventions straight from BlinkDL's own [H, N, N] state
    back = eng.state.back(0)
    N, H = eng.info.head_size, eng.info.num_head
    for l in range(eng.info.num_layer):
        S = back[l, 1:1 + N].reshape(N, H, N).transpose(1, 0, 2)
        assert np.abs(S - ls[l][1].numpy()).max() <= 1e-4 * max(1.0, float(np.abs(S).max()))
        assert np.abs(back[l, 0] - ls[l][0].numpy()).max() <= 1e-4 and np.abs(back[l, N + 1] - ls[l][2].numpy()).max() <= 1e-4
    eng.close()
