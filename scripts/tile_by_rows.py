"""Dev tool (GPU box): one tile-GEMM launch (r/k/v/g shape, 10240 x 2560) by row count and tile shape: does the time grow with the work or with the rounds?"""
import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from ai00_server_amd import runtime as rt
for fmt in (1, 0):
    for shape in [int(x) for x in os.environ.get("SHAPES", "11,4,3,10").split(",")]:
        cells = []
        for T in [int(x) for x in os.environ.get("TS", "256,384,512,768,1024,2048").split(",")]:
            rows = int(os.environ.get("ROWS", "10240"))
            us, blk = rt.bench_gemm(rows, 2560, fmt, T, False, shape, 24 if fmt else 12, 60)
            cells.append(f"T={T}: {us:6.1f} us ({int(blk)} blk, {2.0 * rows * 2560 * T / us / 1e6:4.0f} TF)")
        print(f"fmt{fmt} shape {shape:2d} | " + " | ".join(cells), flush=True)
