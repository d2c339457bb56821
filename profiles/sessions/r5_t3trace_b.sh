#!/bin/bash
# the pipelined tile kernel after moving its addresses to scalar base + lane offset: stage trace, isolated launches, parity of shapes 10 / 11
cd $GRAFT_REPO_ROOT
for shape in 11 10; do
  echo "== trace build, shape $shape T=256"
  RWKV_HIP_LIB=$PWD/ai00_server_amd/librwkv_hip_t3trace.so SHAPES=$shape TS=256 timeout 100 python scripts/tile_by_rows.py 2>&1 | grep -v 'wave [123]' | sort | uniq -c | sort -rn | head -8
done
echo "== product build, isolated launches"
SHAPES=11,10 TS=256,512,2048 timeout 100 python scripts/tile_by_rows.py
echo "== parity"
timeout 300 python -m pytest -x -q "tests/test_gpu_bench_paths.py::test_every_prefill_tile_shape_at_3b_width[11-1]" "tests/test_gpu_bench_paths.py::test_every_prefill_tile_shape_at_3b_width[10-0]" "tests/test_gpu_bench_paths.py::test_every_prefill_tile_shape_at_3b_width[10-2]" 2>&1 | tail -3
