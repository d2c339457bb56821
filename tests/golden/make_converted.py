"""Fixtures made by the REFERENCE's own tooling: tests/golden/converted_v{5,6,7}.st (+ converted_lora.st).

    python tests/golden/make_converted.py          (needs /root/reference; run from the repo root)

`/root/reference/assets/scripts/convert_safetensors.py` (:28-80 conversion, :96-101 rename / transpose lists) is the
only executable artefact of the reference on this path: it turns a BlinkDL `.pth` into the `.st` that ai00_server loads
(lib.rs:580-588).  For each version this script

  1. builds a small checkpoint in BlinkDL's ORIGINAL layout — `time_maa_*` / `time_faaaa` names, LoRA matrices stored
     [in, r] / [r, out] as the trainer writes them, nothing transposed — (`blinkdl_layout`),
  2. saves it with torch.save and runs the reference converter on it in a subprocess (the script parses argv at import;
     `run_reference_converter.py` executes it unmodified and only adapts its one `serialize_file` call to safetensors 0.8),
  3. commits the converter's output bytes (written by the real `safetensors` library: header order, `__metadata__`,
     padding are the library's, not our `st_serialize`'s).  The ORIGINAL tensors are not stored: `sources()` rebuilds
     them from the seeds.

tests/test_converter.py then checks `Loader::info` on those bytes, the oracle and the HIP engine (-m gpu) loaded from
them, against `tests/blinkdl_literal.py` evaluated on the ORIGINAL tensors; where /root/reference is present it also
re-runs the converter and compares byte for byte, so the fixtures stay provably the reference tool's output."""
import os
import subprocess
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
from oracle import rwkv_ref as R  # noqa: E402

CONVERTER = "/root/reference/assets/scripts/convert_safetensors.py"
# convert_safetensors.py:96-101
RENAME = {"time_faaaa": "time_first", "time_maa": "time_mix", "lora_A": "lora.0", "lora_B": "lora.1"}
TRANSPOSE = ["time_mix_w1", "time_mix_w2", "time_decay_w1", "time_decay_w2", "w1", "w2", "a1", "a2", "g1", "g2", "v1", "v2",
             "time_state", "lora.0"]
CASES = {"v5": (5, 2, 128, 128, 64, 61), "v6": (6, 2, 128, 128, 64, 62), "v7": (7, 2, 128, 128, 64, 63)}   # version, L, C, F, V, seed


def blinkdl_layout(version: int, converted: dict) -> dict:
    """Inverse of the converter on a checkpoint in ai00's layout: original names, original orientation (fp16 values kept)."""
    out = {}
    for k, v in converted.items():
        if any(s in k for s in TRANSPOSE):
            v = np.swapaxes(v, -1, -2)
        name = k.replace("time_first", "time_faaaa")
        if version == 6:
            name = name.replace("time_mix", "time_maa")
        out[name] = np.ascontiguousarray(v)
    return out


def lora_pth(rng) -> dict:
    """A LoRA file as the trainer saves it: lora_A [r, in], lora_B [out, r]."""
    r, C = 8, 128
    return {"blocks.0.att.key.lora_A": (rng.standard_normal((r, C)) * 0.05).astype(np.float16),
            "blocks.0.att.key.lora_B": (rng.standard_normal((C, r)) * 0.05).astype(np.float16),
            "blocks.1.ffn.value.lora_A": (rng.standard_normal((r, C)) * 0.05).astype(np.float16),
            "blocks.1.ffn.value.lora_B": (rng.standard_normal((C, r)) * 0.05).astype(np.float16)}


def sources() -> dict:
    """name -> original-layout tensors (numpy fp16), deterministic."""
    src = {}
    for name, (ver, L, C, F, V, seed) in CASES.items():
        src[name] = blinkdl_layout(ver, R.synth_checkpoint(ver, L, C, F, V, seed=seed))
    src["lora"] = lora_pth(np.random.default_rng(64))
    return src


def run_converter(pth: dict, workdir: str) -> bytes:
    import torch
    inp, outp = os.path.join(workdir, "in.pth"), os.path.join(workdir, "out", "model.st")
    torch.save({k: torch.from_numpy(v.copy()) for k, v in pth.items()}, inp)
    # run_reference_converter.py executes the reference script as __main__; it only adapts the `serialize_file` call to the
    # installed safetensors version (see its docstring)
    subprocess.run([sys.executable, os.path.join(HERE, "run_reference_converter.py"), "--input", inp, "--output", outp],
                   check=True, capture_output=True)
    with open(outp, "rb") as f:
        return f.read()


if __name__ == "__main__":
    for name, pth in sources().items():
        with tempfile.TemporaryDirectory() as d:
            data = run_converter(pth, d)
        with open(os.path.join(HERE, f"converted_{name}.st"), "wb") as f:
            f.write(data)
        print(f"converted_{name}.st: {len(data)} bytes, {len(pth)} tensors")
