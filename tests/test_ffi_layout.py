"""CPU: the `#[repr(C)]` structs and constants of integration/rwkv-hip-sys/src/lib.rs against include/rwkv_abi.h, without rustc.

rustc is not in this image, so the sys crate has never been compiled here (integration/check.sh is the procedure).  What CAN be
checked mechanically: `#[repr(C)]` lays a struct out by the C rules (fields in declaration order, each at the next multiple of its
alignment, size rounded up to the struct's alignment), and every field type the crate uses has a fixed size / alignment on the
x86-64 SysV target.  This test parses the Rust source, computes offset / size / alignment of every field by those rules, and compares
them with `offsetof` / `sizeof` / `_Alignof` printed by a C probe compiled against the header — for all six structs — together with
the field NAMES and order, every enum / #define value, and (test_abi_cpu.py) the function list.  Mirrors lib.rs:24-35: the items the
reference imports from web_rwkv are exactly the ones that cross this boundary."""
import os
import re
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HDR = os.path.join(ROOT, "include", "rwkv_abi.h")
RS = os.path.join(ROOT, "integration", "rwkv-hip-sys", "src", "lib.rs")

RUST_TYPES = {"i8": (1, 1), "u8": (1, 1), "i16": (2, 2), "u16": (2, 2), "i32": (4, 4), "u32": (4, 4), "i64": (8, 8), "u64": (8, 8),
              "f32": (4, 4), "f64": (8, 8), "c_float": (4, 4), "c_char": (1, 1), "usize": (8, 8), "isize": (8, 8),
              "rwkv_status": (4, 4)}
STRUCTS = ["rwkv_model_info", "rwkv_lora_desc", "rwkv_load_desc", "rwkv_slot_input", "rwkv_slot_output", "rwkv_sample_params"]


def rust_structs():
    src = open(RS).read()
    out = {}
    for m in re.finditer(r"#\[repr\(C\)\][^\n]*\n?\s*pub struct (\w+)\s*\{([^}]*)\}", src, re.S):
        name, body = m.group(1), m.group(2)
        fields = []
        for f in re.finditer(r"pub\s+(\w+)\s*:\s*([^,}]+)", body):
            ty = f.group(2).strip()
            if ty.startswith("*const") or ty.startswith("*mut"):
                sa = (8, 8)
            else:
                assert ty in RUST_TYPES, f"{name}.{f.group(1)}: unknown Rust type {ty!r}"
                sa = RUST_TYPES[ty]
            fields.append((f.group(1), ty, sa))
        if fields:
            out[name] = fields
    return out


def c_layout_rules(fields):
    off, align, res = 0, 1, []
    for name, _, (size, al) in fields:
        off = (off + al - 1) // al * al
        res.append((name, off, size))
        off += size
        align = max(align, al)
    return res, (off + align - 1) // align * align, align


def header_structs():
    hdr = re.sub(r"/\*.*?\*/", " ", open(HDR).read(), flags=re.S)
    out = {}
    for m in re.finditer(r"typedef struct (\w+)\s*\{([^}]*)\}\s*\w+\s*;", hdr, re.S):
        names = [re.search(r"(\w+)\s*(?:\[\w*\])?\s*$", d.strip()).group(1) for d in m.group(2).split(";") if d.strip()]
        out[m.group(1)] = names
    return out


def header_constants():
    hdr = re.sub(r"/\*.*?\*/", " ", open(HDR).read(), flags=re.S)
    consts = {}
    for m in re.finditer(r"enum\s*\{([^}]*)\}", hdr, re.S):
        for item in m.group(1).split(","):
            if "=" in item:
                k, v = item.split("=")
                consts[k.strip()] = int(v.strip(), 0)
    for m in re.finditer(r"#define\s+(RWKV_\w+)\s+(-?\d+)", hdr):
        consts[m.group(1)] = int(m.group(2))
    return consts


def test_repr_c_structs_match_the_header_layout(tmp_path):
    hs, rs = header_structs(), rust_structs()
    assert sorted(hs) == sorted(STRUCTS), sorted(hs)
    for s in STRUCTS:
        assert s in rs, f"{s} missing from the sys crate"
        assert [f[0] for f in rs[s]] == hs[s], f"{s}: field names / order differ: {[f[0] for f in rs[s]]} vs {hs[s]}"
    # the C side, measured by the compiler
    lines = ['#include <stdio.h>', '#include <stddef.h>', '#include "rwkv_abi.h"', 'int main(void) {']
    for s in STRUCTS:
        lines.append(f'  printf("{s} %zu %zu\\n", sizeof({s}), (size_t)_Alignof({s}));')
        for f in hs[s]:
            lines.append(f'  printf("{s}.{f} %zu %zu\\n", offsetof({s}, {f}), sizeof((({s} *)0)->{f}));')
    lines += ['  printf("ptr %zu %zu\\n", sizeof(void *), sizeof(size_t));', '  return 0;', '}']
    src = tmp_path / "probe.c"
    src.write_text("\n".join(lines) + "\n")
    exe = tmp_path / "probe"
    subprocess.run(["gcc", "-std=c11", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)], check=True)
    got = {}
    for line in subprocess.run([str(exe)], capture_output=True, text=True, check=True).stdout.splitlines():
        k, a, b = line.split()
        got[k] = (int(a), int(b))
    assert got["ptr"] == (8, 8), "the Rust type table above assumes a 64-bit target"
    for s in STRUCTS:
        fields, size, align = c_layout_rules(rs[s])
        assert got[s] == (size, align), f"{s}: C sizeof/alignof {got[s]}, repr(C) rules give {(size, align)}"
        for name, off, fsize in fields:
            assert got[f"{s}.{name}"] == (off, fsize), f"{s}.{name}: C (offset, size) {got[f'{s}.{name}']}, Rust {(off, fsize)}"


def test_every_constant_of_the_header_has_the_same_value_in_the_sys_crate():
    want = header_constants()
    src = open(RS).read()
    got = {m.group(1): int(m.group(2)) for m in re.finditer(r"pub const (RWKV_\w+)\s*:\s*\w+\s*=\s*(-?\d+)\s*;", src)}
    skip = {"RWKV_V5", "RWKV_V6", "RWKV_V7"}            # plain integers in `rwkv_model_info.version`; the wrapper matches on 5 / 6 / 7
    missing = {k: v for k, v in want.items() if k not in skip and got.get(k) != v}
    assert not missing, f"constants that differ or are absent in the sys crate: {missing}"
    extra = {k: v for k, v in got.items() if k not in want}
    assert not extra, f"constants the header does not define: {extra}"


def test_pointer_mutability_and_integer_widths_of_the_functions_match():
    """Per argument: C `const T *` <-> Rust `*const T`, `T *` <-> `*mut T`, and the integer / float width of scalars."""
    hdr = re.sub(r"/\*.*?\*/", " ", open(HDR).read(), flags=re.S)
    rs = open(RS).read()
    cfun = {m.group(1): m.group(2) for m in re.finditer(r"(rwkv_\w+)\s*\(([^;{}]*?)\)\s*;", hdr, re.S) if "typedef" not in m.group(0)}
    rfun = {m.group(1): m.group(2) for m in re.finditer(r"pub fn (rwkv_\w+)\(([^)]*)\)", rs, re.S)}
    scal = {"int32_t": "i32", "int64_t": "i64", "uint64_t": "u64", "uint32_t": "u32", "size_t": "usize", "float": "c_float", "rwkv_status": "rwkv_status"}

    def c_kind(arg):
        arg = " ".join(arg.split())
        arg = re.sub(r"\s*\w+\s*(\[\d*\])\s*$", r" *", arg) if "[" in arg else re.sub(r"\s*\b\w+$", "", arg) if not arg.endswith("*") else arg
        stars = arg.count("*")
        base = arg.replace("*", " ").replace("const", " ").split()[0]
        if stars == 0:
            return scal[base]
        # mutability of the OUTERMOST pointee: `const T *` / `T *` / `const T *const *` / `T *const *` / `T **`
        toks = arg.replace("*", " * ").split()
        last = len(toks) - 1 - toks[::-1].index("*")
        before = toks[:last]
        if stars == 1:
            const = "const" in before
        else:
            inner_last = len(before) - 1 - before[::-1].index("*")
            const = "const" in before[inner_last + 1:]
        return "*const" if const else "*mut"

    def r_kind(arg):
        ty = arg.split(":", 1)[1].strip()
        if ty.startswith("*const"):
            return "*const"
        if ty.startswith("*mut"):
            return "*mut"
        return ty

    for name, cargs in cfun.items():
        if name not in rfun:
            continue
        ca = [a for a in cargs.split(",") if a.strip() and a.strip() != "void"]
        ra = [a for a in rfun[name].split(",") if a.strip()]
        assert len(ca) == len(ra), name
        for i, (c, r) in enumerate(zip(ca, ra)):
            assert c_kind(c) == r_kind(r), f"{name} argument {i}: C `{c.strip()}` vs Rust `{r.strip()}`"


def test_web_rwkv_comparison_script_is_well_formed():
    """integration/compare_with_web_rwkv.sh (the one missing pin as one command: needs Rust + Vulkan + weights, none of which exist here):
    at least it parses, refuses to run without its inputs, and names the routes and request fields the reference's API has
    (api/oai/state.rs:24-27 `input`, completion.rs:49-67 `prompt` / `max_tokens` / `sampler`)."""
    import subprocess
    path = os.path.join(ROOT, "integration", "compare_with_web_rwkv.sh")
    assert os.access(path, os.X_OK)
    assert subprocess.run(["bash", "-n", path]).returncode == 0
    r = subprocess.run(["bash", path], env={k: v for k, v in os.environ.items() if k not in ("AI00", "MODEL")}, capture_output=True, text=True)
    assert r.returncode != 0 and "AI00" in r.stderr
    text = open(path).read()
    for needle in ("/api/oai/", '"completions"', '"states"', '"sampler"', '"top_k": 1', "ai00-core.patch", "cargo build"):
        assert needle in text, needle


def _rust_code(path):
    """a Rust source with comments, string literals and lifetimes blanked out (enough for identifier checks; no raw strings in these files)"""
    s = open(path).read()
    s = re.sub(r"//[^\n]*", "", s)
    s = re.sub(r'"(?:\\.|[^"\\])*"', '""', s)
    return re.sub(r"'[a-z_]+\b(?!')", "", s)


def test_every_type_name_the_safe_wrapper_uses_resolves():
    """No rustc in this image (integration/check.sh has never run here), so the cheapest class of compile error is checked by hand: every
    CamelCase identifier of rwkv-hip's two source files must be defined in the file, imported by a `use`, a variant of one of its enums, a
    generic parameter, or a name of the std prelude / derive set.  (Round 5: `PendingRows` held a `&Runtime` in lib.rs, where the type is
    `Engine` — `Runtime` only exists in compat.rs.)  The sys crate's version follows RWKV_ABI_VERSION."""
    std_names = set("""Vec String Option Some None Ok Err Result Self Box Send Sync Drop Default Debug Clone Copy PartialEq Eq From Into FnMut Fn
                       Arc Mutex CStr CString Deref Target Error Iterator Hash""".split())
    for name in ("lib.rs", "compat.rs"):
        s = _rust_code(os.path.join(ROOT, "integration", "rwkv-hip", "src", name))
        defined = set(re.findall(r"\b(?:struct|enum|type|trait|mod)\s+([A-Z]\w*)", s))
        for body in re.findall(r"\benum\s+\w+\s*\{([^}]*)\}", s):
            defined |= set(re.findall(r"\b([A-Z]\w*)\b", body))
        for group in re.findall(r"\buse\s+([^;]+);", s):
            defined |= set(re.findall(r"\b([A-Z]\w*)\b", group))
        defined |= set(re.findall(r"<\s*([A-Z])\s*[:>,]", s)) | set(re.findall(r",\s*([A-Z])\s*>", s)) | set(re.findall(r"<([A-Z])>", s))   # generic parameters T, U
        used = set(re.findall(r"(?<![\w:.])([A-Z][a-z]\w*)\b", s))                  # CamelCase not behind a path separator (sys::X, crate::X are checked by rustc's import rules above)
        unknown = sorted(u for u in used - defined - std_names if not u.isupper())
        assert not unknown, f"{name}: identifiers that resolve to nothing in scope: {unknown}"
    cargo = open(os.path.join(ROOT, "integration", "rwkv-hip-sys", "Cargo.toml")).read()
    header = open(os.path.join(ROOT, "include", "rwkv_abi.h")).read()
    abi = int(re.search(r"#define\s+RWKV_ABI_VERSION\s+(\d+)", header).group(1))
    assert re.search(r'^version = "0\.(\d+)\.', cargo, re.M).group(1) == str(abi)
    sys_rs = open(os.path.join(ROOT, "integration", "rwkv-hip-sys", "src", "lib.rs")).read()
    assert f"(ABI version {abi})" in sys_rs and f"RWKV_ABI_VERSION: i32 = {abi};" in sys_rs
