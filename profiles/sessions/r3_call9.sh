#!/bin/bash
# round 3, session 9: the round's profile set (kernel statistics of the decode engines and prefill runs, MFMA counters, HBM traffic)
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
bash scripts/gpu_profiles.sh r3 2>&1 | tail -30
for f in gpurun_out/profiles_new/r3_kernel_stats_v6-3b_int8_b32.csv gpurun_out/profiles_new/r3_kernel_stats_v6-3b_int8_b1.csv; do echo $f; head -12 $f; done
cat gpurun_out/profiles_new/r3_pmc_mfma_decode_v6-3b_int8_b32.txt | head -8
python -c "import json;d=json.load(open('gpurun_out/r3_pmc_traffic_v6-3b_int8_b32.json'));print(d['layer_gemm'], d['calibration'])"
