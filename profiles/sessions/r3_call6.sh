#!/bin/bash
# round 3, session 6: K split of the pipelined prefill kernel — parity, A/B, fresh prefill kernel statistics
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O
cd $R
timeout 1200 python -m pytest tests/test_gpu_embeddings.py tests/test_gpu_knobs.py -m gpu -q -k "state_only or KSPLIT or TILE_SHAPE or frozen" > $O/r3_t6.log 2>&1; echo "tests rc=$?"; tail -5 $O/r3_t6.log
for ks in 0 1; do for C in 2048 1024 512; do echo -n "KSPLIT=$ks "; RWKV_TILE_KSPLIT=$ks timeout 200 python scripts/prefill_probe.py v6-3b 1 32 256 $C 2>&1 | tail -1; done; done > $O/r3_ksplit_ab.log 2>&1
for ks in 0 1; do for C in 2048 1024; do echo -n "KSPLIT=$ks "; RWKV_TILE_KSPLIT=$ks timeout 300 python scripts/prefill_probe.py v6-7b 0 8 2048 $C 2>&1 | tail -1; done; done >> $O/r3_ksplit_ab.log 2>&1
for ks in 0 1; do for C in 2048; do echo -n "KSPLIT=$ks "; RWKV_TILE_KSPLIT=$ks timeout 300 python scripts/prefill_probe.py v7-2.9b 2 32 256 $C 2>&1 | tail -1; done; done >> $O/r3_ksplit_ab.log 2>&1
cat $O/r3_ksplit_ab.log
export TMPDIR=/tmp
cd /tmp
for C in 2048 256; do
rm -rf $O/prof_c$C
DEBUG_CLR_GRAPH_PACKET_CAPTURE=0 timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_c$C -o p -- python $R/scripts/prefill_probe.py v6-3b 1 32 256 $C > $O/prof_c$C.log 2>&1; echo "rc=$?"
tr=$(find $O/prof_c$C -name "*kernel_trace.csv" | head -1)
[ -n "$tr" ] && python $R/scripts/summarize_trace.py $tr $O/r3_kernel_stats_prefill_v6-3b_int8_chunk$C.csv --skip-load
head -12 $O/r3_kernel_stats_prefill_v6-3b_int8_chunk$C.csv
rm -rf $O/prof_c$C
done
