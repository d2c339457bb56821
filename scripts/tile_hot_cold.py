import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from ai00_server_amd import runtime as rt
for name, rows, K in (("rkvg", 10240, 2560), ("fkfr", 11520, 2560)):
    for fmt in (1, 0, 2):
        for shape in (4, 11, 10):
            cells = []
            for nmat, lab in ((1, "hot(1 matrix)"), (32 if fmt else 16, "cold")):
                us, blk = rt.bench_gemm(rows, K, fmt, 256, False, shape, nmat, 100)
                cells.append(f"{lab} {us:6.1f} us ({int(blk)} blk)")
            print(f"{name} fmt{fmt} T=256 shape {shape:2d}: " + " | ".join(cells), flush=True)
