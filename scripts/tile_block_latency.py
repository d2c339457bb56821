import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from ai00_server_amd import runtime as rt
# per-BLOCK time of the pipelined 128x64 kernel when its weights come from L2 (one small matrix repeated) / from HBM (many matrices): 8 row blocks x 4 token tiles = 32 blocks, one per CU
for fmt in (1, 0):
    for shape in (11, 12, 4):
        for rows in (1024, 4096):
            cells = []
            for nmat, lab in ((1, "L2-hot"), (96, "cold")):
                us, blk = rt.bench_gemm(rows, 2560, fmt, 256, False, shape, nmat, 100)
                cells.append(f"{lab} {us:6.1f} us ({int(blk)} blk)")
            print(f"fmt{fmt} shape {shape:2d} rows {rows}: " + " | ".join(cells), flush=True)
