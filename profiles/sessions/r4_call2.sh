#!/bin/bash
# round 4, session 2: in-kernel timeline of the decode GEMM at T = 1 / 32, Int8 and fp16 (probes: entry, slice start, X issued, all loads
# issued, [level 2: all loads landed], MFMAs done + parked, after the barrier, exit); full-depth parity with the near-tie arg-max rule; LoRA tests.
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
O=$R/gpurun_out
mkdir -p $O
: > $O/r4c2_trace.log
for lvl in 1 2; do
  for FMT in 1 0; do
    echo "=== trace level $lvl fmt $FMT" >> $O/r4c2_trace.log
    TRACE_LIB=$R/ai00_server_amd/librwkv_hip_trace$([ $lvl = 1 ] && echo 1).so FMT=$FMT TS=32,1 timeout 200 python scripts/trace_gemm.py run >> $O/r4c2_trace.log 2>&1
  done
done
cat $O/r4c2_trace.log
: > $O/full_depth_errors.jsonl
timeout 1200 python -m pytest tests/test_gpu_full_depth.py -x -q -s 2>&1 | grep -v "^$" | tail -40 > $O/r4c2_full_depth.log
cat $O/r4c2_full_depth.log
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -k "lora" 2>&1 | tail -3
