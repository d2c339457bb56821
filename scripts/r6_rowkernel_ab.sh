#!/bin/bash
# Round 6 (GPU box): the V6 token-shift LoRA's first stage sliced over K (steps of 64..1023 rows) and wkv_chunk's requests moved ahead of their phases:
# prefill rates with the slicing off (RWKV_V6_KSP_MAX=1) and on, then kernel statistics of the 256-row and 2048-row steps.
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O; export TMPDIR=/tmp; cd /tmp
for chunk in ${CHUNKS:-256 512 128}; do
  for k in 1 16; do echo "== chunk $chunk RWKV_V6_KSP_MAX=$k"; RWKV_V6_KSP_MAX=$k timeout 200 python $R/scripts/prefill_probe.py v6-3b 1 32 256 $chunk 2>&1 | grep "tok/s" | tail -1; done
done
stats () { name=$1; shift; rm -rf /tmp/prof_$name; DEBUG_CLR_GRAPH_PACKET_CAPTURE=0 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$name -o p -- "$@" > /tmp/prof_$name.log 2>&1
  tr=$(find /tmp/prof_$name -name "*kernel_trace.csv" | head -1); [ -n "$tr" ] && python $R/scripts/summarize_trace.py $tr $O/r6b_kernel_stats_$name.csv --skip-load; grep "tok/s" /tmp/prof_$name.log | tail -1; cat $O/r6b_kernel_stats_$name.csv | cut -c1-150; rm -rf /tmp/prof_$name; }
stats prefill_v6-3b_int8_chunk256 python $R/scripts/prefill_probe.py v6-3b 1 32 256 256
[ -n "${ONLY256:-}" ] || stats prefill_v6-3b_int8_32x256 python $R/scripts/prefill_probe.py v6-3b 1 32 256 2048
