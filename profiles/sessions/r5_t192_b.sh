#!/bin/bash
# shape 12 rebuilt: 192 x 64 tiles in eight waves (two K halves, one block per CU) against 128 x 64 (shape 11): isolated, parity, model
cd $GRAFT_REPO_ROOT
SHAPES=11,12 TS=256,320 timeout 100 python scripts/tile_by_rows.py
ROWS=11520 SHAPES=11,12 TS=256 timeout 100 python scripts/tile_by_rows.py
echo "== parity (oracle tolerance), shape 12"
timeout 300 python -m pytest -x -q "tests/test_gpu_bench_paths.py::test_every_prefill_tile_shape_at_3b_width[12-1]" "tests/test_gpu_bench_paths.py::test_every_prefill_tile_shape_at_3b_width[12-2]" "tests/test_gpu_bench_paths.py::test_every_prefill_tile_shape_at_3b_width[12-0]" 2>&1 | tail -3
echo "== model: rule on / off"
timeout 200 python scripts/prefill_probe.py v6-3b 1 32 256 256
RWKV_DEV_NO_T192=1 timeout 200 python scripts/prefill_probe.py v6-3b 1 32 256 256
timeout 200 python scripts/prefill_probe.py v7-2.9b 2 32 256 256
RWKV_DEV_NO_T192=1 timeout 200 python scripts/prefill_probe.py v7-2.9b 2 32 256 256
