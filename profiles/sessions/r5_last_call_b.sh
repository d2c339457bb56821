FMTS=1 python scripts/gemm_micro.py base 2>&1 | grep -v amdgpu.ids
FMTS=1 RWKV_HIP_LIB=$PWD/ai00_server_amd/librwkv_hip_onlyint8.so python scripts/gemm_micro.py onlyint8 2>&1 | grep -v amdgpu.ids
FMTS=0 python scripts/gemm_micro.py base 2>&1 | grep -v amdgpu.ids
FMTS=0 RWKV_HIP_LIB=$PWD/ai00_server_amd/librwkv_hip_onlyf16.so python scripts/gemm_micro.py onlyf16 2>&1 | grep -v amdgpu.ids
