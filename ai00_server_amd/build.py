"""Build librwkv_hip.so in-tree with hipcc for gfx950 (no cmake needed; ~40 s cold).

    python -m ai00_server_amd.build [--force]
"""
from __future__ import annotations

import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "librwkv_hip.so")
SOURCES = ["rwkv_kernels.hip", "rwkv_engine.cpp", "tokenizer.cpp"]
DEPS = SOURCES + ["rwkv_kernels.h", "safetensors.hpp", "rwkv_abi.map", os.path.join("..", "..", "include", "rwkv_abi.h"),
               os.path.join("..", "..", "include", "rwkv_runtime.hpp"), os.path.join("..", "..", "include", "rwkv_scheduler.hpp"),
               os.path.join("..", "..", "include", "rwkv_router.hpp"),
               os.path.join("..", "..", "harness", "decode_loop.cpp"), os.path.join("..", "..", "harness", "serve_loop.cpp"),
               os.path.join("..", "..", "harness", "router_loop.cpp")]


def _hipcc() -> str:
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    return "hipcc"


STAMP = os.path.join(HERE, ".build_stamp")


def _sources_digest() -> str:
    import hashlib
    h = hashlib.sha256()
    for d in sorted(DEPS):
        with open(os.path.join(CSRC, d), "rb") as f:
            h.update(d.encode())
            h.update(f.read())
    return h.hexdigest()


def needs_build() -> bool:
    """Content-based (mtimes do not survive the copy to a GPU box): rebuild when a source differs from the stamp."""
    if not all(os.path.exists(p) for p in (LIB, HARNESS_BIN, SERVE_BIN, ROUTER_BIN, EMBED_BIN, STAMP)):
        return True
    return open(STAMP).read().strip() != _sources_digest()


KERNEL_PARTS = 6      # rwkv_kernels.hip is compiled once per part (-DRWKV_PART=k), in parallel


def build(force: bool = False, verbose: bool = True) -> str:
    if not force and not needs_build():
        return LIB
    from concurrent.futures import ThreadPoolExecutor
    hipcc = _hipcc()
    jobs = []
    for src in SOURCES:
        stem = os.path.splitext(src)[0]
        base = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC"]
        if src.endswith(".cpp"):
            base += ["-x", "hip"]
        if src == "rwkv_kernels.hip":
            for k in range(KERNEL_PARTS):
                obj = os.path.join(CSRC, f"{stem}.p{k}.o")
                jobs.append((base + [f"-DRWKV_PART={k}", "-c", os.path.join(CSRC, src), "-o", obj], obj))
        else:
            obj = os.path.join(CSRC, stem + ".o")
            jobs.append((base + ["-c", os.path.join(CSRC, src), "-o", obj], obj))

    def run(job):
        if verbose:
            print("[build]", " ".join(job[0]), flush=True)
        subprocess.check_call(job[0])
        return job[1]

    with ThreadPoolExecutor(max_workers=min(len(jobs), os.cpu_count() or 4)) as ex:
        objs = list(ex.map(run, jobs))
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-Wl,--version-script=" + os.path.join(CSRC, "rwkv_abi.map"), "-o", LIB] + objs
    if verbose:
        print("[build]", " ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    build_harness(verbose)
    with open(STAMP, "w") as f:
        f.write(_sources_digest())
    return LIB


HARNESS_SRC = os.path.join(HERE, "..", "harness", "decode_loop.cpp")
HARNESS_BIN = os.path.join(HERE, "..", "harness", "decode_loop")
SERVE_SRC = os.path.join(HERE, "..", "harness", "serve_loop.cpp")
SERVE_BIN = os.path.join(HERE, "..", "harness", "serve_loop")
ROUTER_SRC = os.path.join(HERE, "..", "harness", "router_loop.cpp")
ROUTER_BIN = os.path.join(HERE, "..", "harness", "router_loop")
EMBED_SRC = os.path.join(HERE, "..", "harness", "embed_job.cpp")
EMBED_BIN = os.path.join(HERE, "..", "harness", "embed_job")


def build_harness(verbose: bool = True) -> str:
    """C++ mirror of ai00-core's infer task + greedy loop (harness/decode_loop.cpp), linked against the .so."""
    for src, exe in ((HARNESS_SRC, HARNESS_BIN), (SERVE_SRC, SERVE_BIN), (ROUTER_SRC, ROUTER_BIN), (EMBED_SRC, EMBED_BIN)):
        cmd = ["g++", "-O2", "-std=c++17", "-pthread", src, "-o", exe, "-L" + HERE, "-lrwkv_hip", "-Wl,-rpath," + HERE]
        if verbose:
            print("[build]", " ".join(cmd), flush=True)
        subprocess.check_call(cmd)
    return HARNESS_BIN


if __name__ == "__main__":
    build(force="--force" in sys.argv)
    print(LIB)
