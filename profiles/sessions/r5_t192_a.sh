#!/bin/bash
cd $GRAFT_REPO_ROOT
echo "== isolated launch, 256 rows (shape 11 = 128x64, 12 = 192x64)"
SHAPES=11,12 TS=256,320 timeout 120 python scripts/tile_by_rows.py
ROWS=11520 SHAPES=11,12 TS=256 timeout 120 python scripts/tile_by_rows.py
echo "== embeddings-shaped prefill at chunk 256: new rule, then rule off"
timeout 200 python scripts/prefill_probe.py v6-3b 1 32 256 256
RWKV_DEV_NO_T192=1 timeout 200 python scripts/prefill_probe.py v6-3b 1 32 256 256
timeout 200 python scripts/prefill_probe.py v7-2.9b 2 32 256 256
RWKV_DEV_NO_T192=1 timeout 200 python scripts/prefill_probe.py v7-2.9b 2 32 256 256
