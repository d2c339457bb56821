"""CPU: the C-ABI library loads, exports every symbol include/rwkv_abi.h declares, and its host-only entry points
(model info, tokenizer, error reporting) behave.  No compute call is made without a GPU."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from ai00_server_amd import runtime as rt
from oracle import rwkv_ref as R

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
VOCAB = os.path.join(ROOT, "tests", "golden", "vocab_sample.json")


def test_library_exports_every_declared_symbol(built_lib):
    header = open(os.path.join(ROOT, "include", "rwkv_abi.h")).read()
    declared = set(re.findall(r"\b(rwkv_[a-z0-9_]+)\s*\(", header))
    l = C.CDLL(built_lib)
    for name in sorted(declared):
        assert hasattr(l, name), f"{name} declared in rwkv_abi.h but not exported"
    assert declared == set(rt.ABI_SYMBOLS), declared ^ set(rt.ABI_SYMBOLS)
    want = int(re.search(r"#define\s+RWKV_ABI_VERSION\s+(\d+)", header).group(1))
    assert rt.lib().rwkv_abi_version() == want


def test_model_info_and_format_errors(built_lib):
    for name, ver in [("v5-tiny", 5), ("v6-tiny", 6), ("v7-tiny", 7)]:
        t = R.synth_named(name)
        i = rt.Loader.info(R.st_serialize(t))
        o = R.model_info(t)
        assert (int(i.version), i.num_layer, i.num_emb, i.num_hidden, i.num_vocab, i.num_head) == \
               (ver, o.num_layer, o.num_emb, o.num_hidden, o.num_vocab, o.num_head)
    with pytest.raises(rt.RwkvError) as e:
        rt.Loader.info(b"\x10\x00\x00\x00\x00\x00\x00\x00{not json")
    assert e.value.code == -2                                    # RWKV_ERR_FORMAT
    with pytest.raises(rt.RwkvError) as e:                       # v4-style checkpoint: no ln_x / time_mix_x
        rt.Loader.info(R.st_serialize({"emb.weight": np.zeros((16, 64), np.float16),
                                       "blocks.0.ln1.weight": np.zeros(64, np.float16)}))
    assert e.value.code == -3                                    # RWKV_ERR_UNSUPPORTED
    t = R.synth_named("v6-tiny")
    trunc = R.st_serialize(t)[:-100]
    with pytest.raises(rt.RwkvError):
        rt.Loader.info(trunc)


def test_engine_create_fails_loudly_without_gpu(built_lib):
    if rt.lib().rwkv_device_count() > 0:
        pytest.skip("GPU present")
    with pytest.raises(rt.RwkvError) as e:
        rt.ModelBuilder(R.st_serialize(R.synth_named("v6-tiny"))).build()
    assert e.value.code == -4 and "no CPU fallback" in str(e.value)


def _py_encode(vocab: dict, text: bytes):
    """Reference greedy longest-match (the published RWKV World tokenizer algorithm)."""
    toks = {v: k for k, v in vocab.items()}
    maxlen = max(len(b) for b in toks)
    out, i = [], 0
    while i < len(text):
        for n in range(min(maxlen, len(text) - i), 0, -1):
            if text[i:i + n] in toks:
                out.append(toks[text[i:i + n]])
                i += n
                break
        else:
            raise ValueError("no match")
    return out


def test_tokenizer_matches_reference_algorithm(built_lib):
    import json
    raw = open(VOCAB, encoding="utf-8").read()
    tk = rt.Tokenizer(raw)
    vocab = {}
    for k, v in json.loads(raw).items():
        vocab[int(k)] = v.encode("utf-8") if isinstance(v, str) else bytes(v)
    table = tk.token_index_to_bytes()
    for i, b in vocab.items():
        assert table[i] == b
    samples = ["Hello world!\n\nUser: hi", "The quick brown fox; 12345 + 67 = ?", "你好，世界 — naïve café", "\t\n  x  ",
               "def f(x):\n    return x**2\n"]
    for s in samples:
        data = s.encode("utf-8")
        ids = tk.encode(data)
        assert ids == _py_encode(vocab, data)
        assert tk.decode(ids) == data
    assert tk.encode(b"") == []
    with pytest.raises(rt.RwkvError):
        tk.decode([10 ** 7])


REF_VOCAB = "/root/reference/assets/tokenizer/rwkv_vocab_v20230424.json"


@pytest.mark.skipif(not os.path.exists(REF_VOCAB), reason="the reference checkout (and its full World vocabulary) is not on this machine")
def test_tokenizer_on_the_full_reference_vocabulary(built_lib):
    """The reference's own 65 529-entry vocabulary file (lib.rs:375 loads exactly this asset), read in place: table identical,
    encode == greedy longest match over the whole vocabulary, decode round-trips, on the reference's README texts + random bytes."""
    import json
    raw = open(REF_VOCAB, encoding="utf-8").read()
    tk = rt.Tokenizer(raw)
    vocab = {int(k): (v.encode("utf-8") if isinstance(v, str) else bytes(v)) for k, v in json.loads(raw).items()}
    assert len(vocab) == 65529 and min(vocab) == 1 and max(vocab) == 65529
    table = tk.token_index_to_bytes()
    for i, b in vocab.items():
        assert table[i] == b
    texts = []
    for f in ("README.md", "README.zh.md"):
        path = os.path.join("/root/reference", f)
        if os.path.exists(path):
            texts.append(open(path, "rb").read())
    texts.append(np.random.default_rng(5).integers(0, 256, 5000, dtype=np.uint8).tobytes())
    # longest-match oracle with a first-byte index (the plain _py_encode is quadratic in the vocabulary size)
    by_first = {}
    for i, b in vocab.items():
        by_first.setdefault(b[0], []).append((len(b), b, i))
    for v in by_first.values():
        v.sort(reverse=True)
    for data in texts:
        want, pos = [], 0
        while pos < len(data):
            for n, b, i in by_first[data[pos]]:
                if data[pos:pos + n] == b:
                    want.append(i)
                    pos += n
                    break
        ids = tk.encode(data)
        assert ids == want
        assert tk.decode(ids) == data


def test_cpp_mirror_builds_and_reports_errors(built_lib, tmp_path):
    """The C++ host mirror (include/rwkv_runtime.hpp + harness/decode_loop.cpp) compiles against the header and
    surfaces engine errors as exit code 1 (no GPU here) instead of aborting."""
    import subprocess
    from ai00_server_amd import build as B
    exe = B.build_harness(verbose=False)
    assert subprocess.run([exe], capture_output=True).returncode == 2
    path = tmp_path / "m.st"
    path.write_bytes(R.st_serialize(R.synth_named("v6-tiny")))
    r = subprocess.run([exe, str(path), "0", "0", "1", "8", "2", "5", "6"], capture_output=True, text=True)
    if rt.lib().rwkv_device_count() == 0:
        assert r.returncode == 1 and "no HIP device" in r.stderr
    else:
        assert r.returncode == 0


def test_chunk_policy_water_filling(built_lib):
    """Host logic of rwkv_infer's chunk split: budget respected, decode slots never starved, everything consumed."""
    assert rt.plan_chunk([1, 1, 1, 0], 128) == [1, 1, 1, 0]
    assert rt.plan_chunk([4096, 1, 1, 0], 128) == [126, 1, 1, 0]              # long prefill cannot starve decode slots
    assert rt.plan_chunk([300, 300], 128) == [64, 64]
    assert rt.plan_chunk([10, 500, 3], 128) == [10, 115, 3]
    assert sum(rt.plan_chunk([50] * 8, 32)) == 32 and max(rt.plan_chunk([50] * 8, 32)) == 4
    got = rt.plan_chunk([1] * 40, 16)                                        # more slots than budget: first come first
    assert sum(got) == 16 and set(got) == {0, 1}
    assert rt.plan_chunk([0, 0], 8) == [0, 0]
    with pytest.raises(rt.RwkvError):
        rt.plan_chunk([1], 0)
    # repeated calls drain any workload in a bounded number of steps
    pending = [1000, 7, 0, 64, 1]
    calls = 0
    while sum(pending):
        take = rt.plan_chunk(pending, 128)
        assert 0 < sum(take) <= 128 and all(t <= p for t, p in zip(take, pending))
        pending = [p - t for p, t in zip(pending, take)]
        calls += 1
    assert calls == -(-1072 // 128)


def test_prefab_images_are_sniffed_and_bad_ones_rejected(built_lib):
    """`rwkv_model_info_from_st` takes safetensors or a prefab image by content (lib.rs:585-588); a truncated or
    wrong-version image is an error code, never an abort.  (Writing/reading a real image needs a GPU: test_gpu_parity.)"""
    import struct
    info = (6, 2, 128, 448, 256, 2, 64, 0)
    good = b"RWKVHIP\0" + struct.pack("<II", 1, 0) + struct.pack("<8i", *info) + struct.pack("<ii", 0, 0)
    got = rt.Loader.info(good)
    assert (got.version, got.num_layer, got.num_emb, got.num_hidden, got.num_vocab, got.num_head) == info[:6]
    bad_version = b"RWKVHIP\0" + struct.pack("<II", 99, 0) + good[16:]
    with pytest.raises(rt.RwkvError) as e:
        rt.Loader.info(bad_version)
    assert e.value.code == -3
    with pytest.raises(rt.RwkvError):
        rt.Loader.info(b"RWKVHIP\0" + b"\x01\x00")             # shorter than a header: falls through to the safetensors parser


def test_graft_entry_build_is_green(built_lib):
    """`__graft_entry__.build()` is the driver's does-it-build check: it must pass on the tree as committed (a stale
    hard-coded ABI version once made it fail while every other test was green)."""
    import __graft_entry__ as g
    g.build()


def _header_functions():
    """name -> number of parameters, from include/rwkv_abi.h"""
    import re
    hdr = open(os.path.join(ROOT, "include", "rwkv_abi.h")).read()
    hdr = re.sub(r"/\*.*?\*/", " ", hdr, flags=re.S)
    out = {}
    for m in re.finditer(r"(?:const\s+char\s*\*|rwkv_status|int32_t|int64_t|uint64_t|size_t|void)\s*\*?\s*(rwkv_\w+)\s*\(([^;]*?)\)\s*;", hdr, re.S):
        args = m.group(2).strip()
        out[m.group(1)] = 0 if args in ("", "void") else args.count(",") + 1
    return out


def test_rust_sys_crate_declares_every_export_with_the_headers_arity():
    """integration/rwkv-hip-sys/src/lib.rs (the FFI a maintainer links ai00-core against) against include/rwkv_abi.h: same
    function names, same number of arguments, same ABI version; and every one of them is exported by the built library."""
    import re
    want = _header_functions()
    assert len(want) >= 39 and set(want) == set(rt.ABI_SYMBOLS), sorted(set(want) ^ set(rt.ABI_SYMBOLS))
    rs = open(os.path.join(ROOT, "integration", "rwkv-hip-sys", "src", "lib.rs")).read()
    got = {}
    for m in re.finditer(r"pub fn (rwkv_\w+)\(([^)]*)\)", rs, re.S):
        args = m.group(2).strip()
        got[m.group(1)] = 0 if not args else len([a for a in args.split(",") if a.strip()])
    assert got == want, {k: (got.get(k), want.get(k)) for k in set(got) | set(want) if got.get(k) != want.get(k)}
    hv = int(re.search(r"#define\s+RWKV_ABI_VERSION\s+(\d+)", open(os.path.join(ROOT, "include", "rwkv_abi.h")).read()).group(1))
    assert int(re.search(r"RWKV_ABI_VERSION: i32 = (\d+)", rs).group(1)) == hv
    for name, (res, argt) in rt.ABI_SYMBOLS.items():                     # the ctypes mirror agrees on the arity too
        assert len(argt) == want[name], name
    md = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    assert "integration/rwkv-hip-sys" in md and "integration/ai00-core.patch" in md


def test_library_exports_only_the_c_abi(built_lib):
    """Linked with csrc/rwkv_abi.map: no C++ symbol of the engine or of the kernel launchers leaves the library."""
    import subprocess
    out = subprocess.run(["nm", "-D", "--defined-only", built_lib], capture_output=True, text=True, check=True).stdout
    names = [l.split()[-1] for l in out.splitlines() if l.strip()]
    assert names and all(n.startswith("rwkv_") for n in names), [n for n in names if not n.startswith("rwkv_")][:10]
    assert set(names) >= set(rt.ABI_SYMBOLS)


def test_header_is_plain_c99(tmp_path):
    """The boundary is a C ABI: the header must compile as C99 with no extensions."""
    import subprocess
    src = tmp_path / "t.c"
    src.write_text('#include "rwkv_abi.h"\nint main(void) { rwkv_load_desc d; rwkv_sample_params p; (void)d; (void)p; return RWKV_ABI_VERSION > 0 ? 0 : 1; }\n')
    subprocess.run(["gcc", "-std=c99", "-pedantic", "-Wall", "-Wextra", "-Werror", "-I", os.path.join(ROOT, "include"), "-fsyntax-only", str(src)], check=True)


def test_reference_patch_still_applies():
    """integration/ai00-core.patch against the reference checkout (only where it is present)."""
    import subprocess
    ref = "/root/reference"
    if not os.path.isdir(os.path.join(ref, "crates", "ai00-core")):
        pytest.skip("the reference checkout is not on this machine")
    r = subprocess.run(["git", "apply", "--check", os.path.join(ROOT, "integration", "ai00-core.patch")], cwd=ref, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr


def test_patched_workspace_names_only_items_the_wrapper_exports(tmp_path):
    """What can be checked of the patch without rustc: applied to a scratch copy of the reference's `crates/`, (1) no Rust source of the
    workspace still names `web_rwkv` (round 5 found reload.rs, sampler/bnf.rs and two ai00-server files the patch had not reached — the
    dependency was gone, the imports were not), (2) every item imported from `rwkv_hip` / `rwkv_hip::compat` is a `pub` item of
    integration/rwkv-hip, (3) the call of `load_model_state` matches its patched signature, (4) the derives ai00-core's types demand of
    the wrapper's items are there (serde on TensorCpu / ModelInfo / Quant, Debug on Tokenizer / Context / TensorGpu)."""
    import re, shutil, subprocess
    ref = "/root/reference"
    if not os.path.isdir(os.path.join(ref, "crates", "ai00-core")):
        pytest.skip("the reference checkout is not on this machine")
    shutil.copytree(os.path.join(ref, "crates"), tmp_path / "crates")
    subprocess.run(["patch", "-p1", "-s", "-i", os.path.join(ROOT, "integration", "ai00-core.patch")], cwd=tmp_path, check=True)
    sources = {}
    for d, _, files in os.walk(tmp_path / "crates"):
        for f in files:
            if f.endswith(".rs"):
                sources[os.path.relpath(os.path.join(d, f), tmp_path)] = open(os.path.join(d, f)).read()
    left = [p for p, s in sources.items() if re.search(r"\bweb_rwkv::|use web_rwkv\b", s)]
    assert not left, f"still importing web_rwkv after the patch: {left}"
    for toml in ("crates/ai00-core/Cargo.toml", "crates/ai00-server/Cargo.toml"):
        t = open(tmp_path / toml).read()
        assert "rwkv-hip = { path" in t and "web-rwkv.workspace" not in t, toml
    lib = open(os.path.join(ROOT, "integration", "rwkv-hip", "src", "lib.rs")).read()
    compat = open(os.path.join(ROOT, "integration", "rwkv-hip", "src", "compat.rs")).read()
    pub = lambda s: set(re.findall(r"\bpub (?:struct|enum|type|fn|use crate::)\s*([A-Za-z_]\w*)", s))
    exported = {"rwkv_hip": pub(lib) | {"compat"}, "rwkv_hip::compat": pub(compat)}
    n_imports = 0
    for path, s in sources.items():
        for m in re.finditer(r"use (rwkv_hip(?:::compat)?)::(\{[^;]*\}|\w+)\s*;", s):
            body = m.group(2)
            # nested groups: `rwkv_hip::{compat::{A, B}, Adapter}`
            for inner in re.finditer(r"compat::\{([^}]*)\}", body):
                for name in re.findall(r"(\w+)(?:\s+as\s+\w+)?\s*(?:,|$)", inner.group(1).strip()):
                    assert name in exported["rwkv_hip::compat"], f"{path}: rwkv_hip::compat::{name} is not exported"
                    n_imports += 1
            flat = re.sub(r"compat::\{[^}]*\}", "", body).strip("{} \n")
            for name in re.findall(r"(\w+)(?:\s+as\s+\w+)?\s*(?:,|$)", flat):
                assert name in exported[m.group(1)], f"{path}: {m.group(1)}::{name} is not exported"
                n_imports += 1
    assert n_imports >= 25, n_imports
    core_lib, run = sources["crates/ai00-core/src/lib.rs"], sources["crates/ai00-core/src/run.rs"]
    assert re.search(r"fn load_model_state\(state: &State, data: &\[u8\]\)", core_lib)
    calls = re.findall(r"load_model_state\(([^)]*)\)", core_lib + run)
    assert calls and all(c.count(",") == 1 for c in calls), calls               # two arguments everywhere: (&State, bytes)
    assert "state.len()" not in core_lib and "init_states.len()" in core_lib          # the request's list is not shadowed by the engine's State
    assert "struct Prefab" not in core_lib and "TensorError" in re.search(r"use rwkv_hip::\{[^;]*\};", core_lib).group(0)
    for item, needs in (("pub struct TensorCpu", ("Serialize", "Deserialize", "Debug", "Clone")), ("pub struct ModelInfo", ("Serialize", "Deserialize", "Debug", "Clone")),
                        ("pub enum Quant", ("Serialize", "Deserialize", "Default", "Debug", "Clone")), ("pub struct TensorGpu", ("Debug", "Clone")),
                        ("pub struct Context", ("Debug", "Clone")), ("pub struct Tokenizer", ("Debug",)), ("pub struct DeviceState", ("Debug",)), ("pub struct Engine", ("Debug",))):
        src = compat if item in compat else lib
        head = src[:src.index(item)]
        derive = head[head.rindex("#[derive("):]
        assert derive.count("\n") <= 2, item                                        # the derive belongs to this item, not an earlier one
        for d in needs:
            assert re.search(r"\b%s\b" % d, derive), f"{item} must derive {d}"
    assert 'features = ["derive", "rc"]' in open(os.path.join(ROOT, "integration", "rwkv-hip", "Cargo.toml")).read()


def test_mutated_inputs_never_abort(built_lib, tmp_path):
    """"Nothing aborts" (include/rwkv_abi.h): tests/cpp/fuzz_cpu_entry_points.cpp throws mutated safetensors headers, truncated files,
    mutated vocabularies, random byte strings / token ids / chunk plans at the entry points that run on a CPU; the process must come
    back with statuses only.  (The sanitizer script under scripts/ — CPU only, see scripts/README.md — runs the same driver, and the C++ host
    tests, against an instrumented build of the library's host code: 100,000 iterations clean at the time of writing.)"""
    import subprocess
    models = []
    for name in ("v5-tiny", "v6-tiny", "v7-tiny"):
        p = tmp_path / (name + ".st")
        p.write_bytes(R.st_serialize(R.synth_named(name)))
        models.append(str(p))
    exe = str(tmp_path / "fuzz")
    pkg = os.path.join(ROOT, "ai00_server_amd")
    subprocess.check_call(["g++", "-O1", "-std=c++17", "-Wall", "-Wextra", "-Werror", os.path.join(ROOT, "tests", "cpp", "fuzz_cpu_entry_points.cpp"),
                           "-o", exe, "-L" + pkg, "-lrwkv_hip", "-Wl,-rpath," + pkg])
    out = subprocess.run([exe, "4000", os.path.join(ROOT, "tests", "golden", "vocab_sample.json")] + models, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and "no crash" in out.stdout, out.stdout[-500:] + out.stderr[-2000:]
