"""Dev tool (GPU box): the decode GEMM launches of a V6-3B layer, isolated (rwkv_bench_gemm: graph-captured dependent launches over
matrices larger than the Infinity Cache).  One line per (shape, format, T): us per launch and TB/s of stored weight bytes.
    RWKV_HIP_LIB=... python scripts/gemm_micro.py [label]
Env: FMTS=1,0  TS=1,16,32  SHAPES=rkvg,fkfr,fv,wo"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ai00_server_amd import runtime as rt

label = sys.argv[1] if len(sys.argv) > 1 else "base"
SH = {"rkvg": (10240, 2560), "fkfr": (11520, 2560), "fv": (2560, 8960), "wo": (2560, 2560), "head": (65536, 2560)}
BPW = {0: 2.0, 1: 1.03125, 2: 0.53125}
fmts = [int(x) for x in os.environ.get("FMTS", "1,0").split(",")]
Ts = [int(x) for x in os.environ.get("TS", "1,16,32").split(",")]
for name in os.environ.get("SHAPES", "rkvg,fkfr,fv,wo").split(","):
    rows, K = SH[name]
    for fmt in fmts:
        cells = []
        for T in Ts:
            wb = rows * K * BPW[fmt]
            nmat = max(2, min(64, int(700e6 / wb)))
            us, blocks = rt.bench_gemm(rows, K, fmt, T, False, 0, nmat, 300)
            cells.append(f"T={T:2d} {us:6.2f} us {wb / us / 1e6:5.2f} TB/s ({int(blocks)} blk)")
        print(f"{label:10s} {name:5s} fmt{fmt} | " + " | ".join(cells), flush=True)
