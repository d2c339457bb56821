"""Dev tool: build an A/B variant of the library with extra -D flags.
    python scripts/build_variant.py NAME -DFOO -DBAR=1   ->  ai00_server_amd/librwkv_hip_NAME.so
    RWKV_HIP_LIB=$PWD/ai00_server_amd/librwkv_hip_NAME.so python bench.py ...
"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "ai00_server_amd")
name, flags = sys.argv[1], sys.argv[2:]
cs = os.path.join(PKG, "csrc")
cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-Wl,--version-script=" + os.path.join(cs, "rwkv_abi.map"), *flags, "-o", os.path.join(PKG, f"librwkv_hip_{name}.so")]
for s in ["rwkv_kernels.hip", "rwkv_engine.cpp", "tokenizer.cpp"]:
    cmd += (["-x", "hip"] if s.endswith(".cpp") else []) + [os.path.join(cs, s)]
subprocess.check_call(cmd)
print(cmd[-7])
