export FMTS=1 TS=1,32 SHAPES=rkvg,fkfr,fv,wo
for i in 1 2; do
python scripts/gemm_micro.py base 2>&1 | grep -v amdgpu.ids
RWKV_HIP_LIB=$PWD/ai00_server_amd/librwkv_hip_earlypin.so python scripts/gemm_micro.py earlypin 2>&1 | grep -v amdgpu.ids
done
timeout 600 python scripts/ab_bench.py "v6 base::" "v6 earlypin::ai00_server_amd/librwkv_hip_earlypin.so" "v6 base::" "v6 earlypin::ai00_server_amd/librwkv_hip_earlypin.so" 2>&1 | grep -v amdgpu.ids
