"""(Needs the library of commit 0a1ea2f: the stream-K code was measured and removed.)  Round 6 (GPU box): stream-K over the pipelined tile kernels (RWKV_BENCH_SK=<blocks>) against the classic grids, isolated launches of the 3 B / 7 B layer's
matrices; the engine library prints a check line per configuration on stderr (dealt-out against classic on the same operands)."""
import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from ai00_server_amd import runtime as rt
MATS = {"rkvg": (10240, 2560), "fkfr": (11520, 2560), "fv": (2560, 8960), "wo": (2560, 2560), "rkvg7b": (16384, 4096), "fv7b": (4096, 14336)}
fmts = [int(x) for x in os.environ.get("FMTS", "1").split(",")]
ts = [int(x) for x in os.environ.get("TS", "256,2048").split(",")]
mats = os.environ.get("MATS", "rkvg,fkfr,fv,wo").split(",")
for mat in mats:
    rows, K = MATS[mat]
    for fmt in fmts:
        for hilo, shapes in ((0, (11, 10)), (1, (12,))):
            for shape in shapes:
                for T in ts:
                    cells = []
                    for sk in [int(x) for x in os.environ.get("SKS", "0,512,256").split(",")]:
                        if sk: os.environ["RWKV_BENCH_SK"] = str(sk)
                        else: os.environ.pop("RWKV_BENCH_SK", None)
                        us, blk = rt.bench_gemm(rows, K, fmt, T, bool(hilo), shape, 24 if fmt else 12, 40)
                        cells.append(f"sk={sk}: {us:7.1f} us ({int(blk)} blk, {2.0 * rows * K * T * (2 if hilo else 1) / us / 1e6:5.0f} TF)")
                    print(f"{mat:6s} fmt{fmt} hilo{hilo} shape {shape:2d} T={T:5d} | " + " | ".join(cells), flush=True)
os.environ.pop("RWKV_BENCH_SK", None)
