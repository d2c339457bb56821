#!/bin/bash
# round 3, session 17: wide v6_mix numbered by XCD (parity + per-kernel time at 2048 / 1024 rows), ln_shift thread policy parity
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_knobs.py -m gpu -q -k "wide_mix or LN_THREADS or TILE_XCD" > $O/r3_t17.log 2>&1; echo "tests rc=$?"; tail -4 $O/r3_t17.log
timeout 900 python -m pytest tests/test_gpu_embeddings.py -m gpu -q > $O/r3_t17b.log 2>&1; echo "emb tests rc=$?"; tail -3 $O/r3_t17b.log
export TMPDIR=/tmp
cd /tmp
for x in 1 0; do
  for chunk in 2048 1024; do
    rm -rf $O/prof_v6xcd
    RWKV_TILE_XCD=$x timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_v6xcd -o p -- python $R/scripts/prefill_probe.py v6-3b 1 32 256 $chunk > $O/prof_v6xcd.log 2>&1
    tr=$(find $O/prof_v6xcd -name "*kernel_trace.csv" | head -1)
    python $R/scripts/summarize_trace.py $tr $O/r3_v6xcd_${x}_$chunk.csv --skip-load > /dev/null
    echo "TILE_XCD=$x chunk=$chunk: $(grep v6_mix $O/r3_v6xcd_${x}_$chunk.csv | cut -c1-120)"; tail -1 $O/prof_v6xcd.log
  done
done
cd $R
for rep in 1 2; do for x in 1 0; do RWKV_TILE_XCD=$x python scripts/prefill_probe.py v6-3b 1 32 256 2048 | tail -1 | sed "s/^/TILE_XCD=$x /"; done; done
