#!/bin/bash
# round 4, session 18: the round's profile set (kernel statistics of the decode engines and prefill runs, MFMA counters, HBM traffic)
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
bash scripts/gpu_profiles.sh r4 2>&1 | tail -30
for f in gpurun_out/profiles_new/r4_kernel_stats_v6-3b_int8_b32.csv gpurun_out/profiles_new/r4_kernel_stats_prefill_v6-3b_int8_32x256.csv gpurun_out/profiles_new/r4_kernel_stats_prefill_v6-3b_int8_chunk256.csv; do echo $f; head -14 $f | cut -c1-180; done
python -c "import json;d=json.load(open('gpurun_out/r4_pmc_traffic_v6-3b_int8_b32.json'));print(d['layer_gemm'], d['calibration'])"
