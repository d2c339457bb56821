"""GEMM kernel microbench sweep (dev tool): us/launch and GB/s of weight streaming."""
import sys, os, itertools
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ai00_server_amd import runtime as rt

def wbytes(rows, K, fmt):
    return rows * K * {0: 2.0, 1: 1.03125, 2: 0.53125}[fmt]

shapes = [("rkvg", 10240, 2560), ("wo", 2560, 2560), ("fk", 8960, 2560), ("fv", 2560, 8960), ("w1", 160, 2560), ("head", 65536, 2560)]
fmts = [int(x) for x in os.environ.get("FMTS", "1,0").split(",")]
Ts = [int(x) for x in os.environ.get("TS", "1,8,32").split(",")]
ksws = [int(x) for x in os.environ.get("SPBS", "0,1,2,4,8").split(",")]
for name, rows, K in shapes:
    for fmt in fmts:
        if name in ("w1", "head") and fmt != 0:
            continue
        for T in Ts:
            res = []
            for ksw in ksws:
                nmat = max(2, min(64, int(600e6 / wbytes(rows, K, fmt))))
                try:
                    us, lds = rt.bench_gemm(rows, K, fmt, T, False, ksw, nmat, 200 if rows < 60000 else 40)
                    res.append(f"spb{ksw}:{us:6.2f}us {wbytes(rows,K,fmt)/us/1e3:5.0f}GB/s {lds:.0f}blk")
                except Exception as e:
                    res.append(f"spb{ksw}: ERR {e}")
            print(f"{name:5s} fmt{fmt} T={T:3d} | " + " | ".join(res), flush=True)
