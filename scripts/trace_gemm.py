"""Dev tool: in-kernel timeline of the skinny GEMM (wall_clock64 probes compiled in with -DRWKV_TRACE=N).

    python scripts/trace_gemm.py build [level]     # on the build host: ai00_server_amd/librwkv_hip_trace.so
    python scripts/trace_gemm.py run               # on the GPU box
Probes per wave: 0 entry, 1 loads issued, (5 all loads landed, level 2), 2 MFMAs done + parked, 3 after barrier, 4 exit.
"""
import ctypes, os, subprocess, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "ai00_server_amd")
LIB = os.environ.get("TRACE_LIB") or os.path.join(PKG, "librwkv_hip_trace.so")

def build(level):
    cs = os.path.join(PKG, "csrc")
    srcs = ["rwkv_kernels.hip", "rwkv_engine.cpp", "tokenizer.cpp"]
    cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-Wl,--version-script=" + os.path.join(cs, "rwkv_abi.map"), f"-DRWKV_TRACE={level}", "-o", LIB] + os.environ.get("XFLAGS", "").split()
    for s in srcs:
        cmd += (["-x", "hip"] if s.endswith(".cpp") else []) + [os.path.join(cs, s)]
    subprocess.check_call(cmd)

def run():
    lib = ctypes.CDLL(LIB)
    us = ctypes.c_float(); blk = ctypes.c_float()
    cases = [("K3", 10240 + 64, 2560, 1), ("FkFr", 8960 + 2560, 2560, 1), ("Fv", 2560, 8960, 1), ("Wo", 2560, 2560, 1)]
    if os.environ.get("FMT"):
        cases = [(n, r, k, int(os.environ["FMT"])) for n, r, k, _ in cases]
    for T in [int(x) for x in os.environ.get("TS", "32,1").split(",")]:
        for name, rows, K, fmt in cases:
            buf = np.zeros(4096 * 8, dtype=np.uint64)
            lib.rwkv_debug_trace(buf.ctypes.data_as(ctypes.c_void_p), buf.size)      # reset
            rc = lib.rwkv_bench_gemm(rows, K, fmt, T, 0, 0, 8, 50, ctypes.byref(us), ctypes.byref(blk))
            assert rc == 0
            buf = np.zeros(4096 * 8, dtype=np.uint64)
            lib.rwkv_debug_trace(buf.ctypes.data_as(ctypes.c_void_p), buf.size)
            tr = buf.reshape(256, 16, 8).astype(np.int64)
            nb = int(min(blk.value, 256))
            tr = tr[:nb]
            act = tr[:, :, 0] > 0
            t0 = tr[:, :, 0][act].min()
            def st(i):
                v = (tr[:, :, i][act & (tr[:, :, i] > 0)] - t0) / 100.0   # us (100 MHz)
                return "n/a" if v.size == 0 else f"min {v.min():5.2f} med {np.median(v):5.2f} max {v.max():5.2f}"
            print(f"T={T} fmt={fmt} {name:5s} rows={rows} K={K}: {us.value:6.2f} us/launch, {blk.value:.0f} blocks, waves/block {act[0].sum()}")
            for i, lab in ((0, "entry"), (6, "slice start"), (7, "X issued"), (1, "X+W issued"), (5, "loads landed"), (2, "mfma+park done"), (3, "after barrier"), (4, "exit")):
                print(f"      {lab:15s} {st(i)}")

if __name__ == "__main__":
    if sys.argv[1] == "build":
        build(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
    else:
        run()
