#!/bin/bash
# Round 6 (GPU box): the shader clock idle, during the decode bench, and during 2048-row / 256-row prefill (scripts/clock_probe.hip beside the workload;
# the workload's process needs ~5 s to import, synthesise and load before the GPU is busy: read the series, not the mean).
R=${GRAFT_REPO_ROOT:-/root/repo}; cd /tmp; export TMPDIR=/tmp
python -c "import torch" 2>/dev/null   # (page the image in first)
echo "idle:            $($R/scripts/clock_probe.bin 3)"
$R/scripts/clock_probe.bin 16 > /tmp/clk_decode.txt & p=$!
python $R/bench.py --decode-only --no-cpu-baseline --sweep= --verify-steps 0 --steps 1500 --warmup 5 --repeats 2 > /tmp/bench_clk.log 2>&1; echo "bench rc=$?"; wait $p
echo "decode B=32:     $(cat /tmp/clk_decode.txt)"
$R/scripts/clock_probe.bin 16 > /tmp/clk_p2048.txt & p=$!
for i in 1 2; do python $R/scripts/prefill_probe.py v6-3b 1 32 2048 2048 > /dev/null 2>&1; done; wait $p
echo "prefill 2048:    $(cat /tmp/clk_p2048.txt)"
$R/scripts/clock_probe.bin 16 > /tmp/clk_p256.txt & p=$!
python $R/scripts/prefill_probe.py v6-3b 1 32 1024 256 > /dev/null 2>&1; wait $p
echo "prefill 256:     $(cat /tmp/clk_p256.txt)"
