"""Dev tool: build an A/B variant of the library with extra -D flags.
    python scripts/build_variant.py NAME -DFOO -DBAR=1            ->  ai00_server_amd/librwkv_hip_NAME.so   (one translation unit, ~3 min)
    python scripts/build_variant.py NAME --parts 0,3 -DFOO        ->  only kernel parts 0 and 3 are recompiled with the flags; the other
                                                                      objects are the product build's (run ai00_server_amd.build first)
    RWKV_HIP_LIB=$PWD/ai00_server_amd/librwkv_hip_NAME.so python bench.py ...
"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "ai00_server_amd")
name, flags = sys.argv[1], sys.argv[2:]
cs = os.path.join(PKG, "csrc")
out = os.path.join(PKG, f"librwkv_hip_{name}.so")
base = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC"]
if "--parts" in flags:
    i = flags.index("--parts")
    parts = [int(x) for x in flags[i + 1].split(",")]
    flags = flags[:i] + flags[i + 2:]
    sys.path.insert(0, ROOT)
    from ai00_server_amd import build as B
    B.build(verbose=False)
    objs = []
    for k in range(B.KERNEL_PARTS):
        if k in parts:
            o = os.path.join(cs, f"rwkv_kernels.{name}.p{k}.o")
            subprocess.check_call(base + flags + [f"-DRWKV_PART={k}", "-c", os.path.join(cs, "rwkv_kernels.hip"), "-o", o])
            objs.append(o)
        else:
            objs.append(os.path.join(cs, f"rwkv_kernels.p{k}.o"))
    objs += [os.path.join(cs, "rwkv_engine.o"), os.path.join(cs, "tokenizer.o")]
    subprocess.check_call(base[:2] + ["-shared", "-fPIC", "-Wl,--version-script=" + os.path.join(cs, "rwkv_abi.map"), "-o", out] + objs)
else:
    cmd = base + ["-shared", "-Wl,--version-script=" + os.path.join(cs, "rwkv_abi.map"), *flags, "-o", out]
    for s in ["rwkv_kernels.hip", "rwkv_engine.cpp", "tokenizer.cpp"]:
        cmd += (["-x", "hip"] if s.endswith(".cpp") else []) + [os.path.join(cs, s)]
    subprocess.check_call(cmd)
print(out)
