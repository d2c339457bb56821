"""Dev tool (GPU box): the software-pipelined tile kernel (shapes 12 / 13) against gemm_tile3 (10 / 11) and the 64x64 shape (4 / 3), isolated launches
of the 3 B layer's matrices on random operands: r/k/v/g (10240 x 2560), Fk (8960 x 2560), Fv (2560 x 8960), Wo (2560 x 2560); fp16 / Int8 / NF4;
plain and hi + lo operands.  FMTS, TS, SHAPES, MATS select."""
import os
import sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from ai00_server_amd import runtime as rt

MATS = {"rkvg": (10240, 2560), "fk": (8960, 2560), "fv": (2560, 8960), "wo": (2560, 2560), "rkvg7b": (16384, 4096)}
fmts = [int(x) for x in os.environ.get("FMTS", "1,0,2").split(",")]
ts = [int(x) for x in os.environ.get("TS", "256,512,2048").split(",")]
shapes = [int(x) for x in os.environ.get("SHAPES", "4,11,10,12").split(",")]
mats = os.environ.get("MATS", "rkvg,fv").split(",")
hilos = [int(x) for x in os.environ.get("HILO", "0,1").split(",")]
for mat in mats:
    rows, K = MATS[mat]
    for fmt in fmts:
        for hilo in hilos:
            for shape in shapes:
                if (hilo and shape in (10, 11)) or (not hilo and shape == 12):
                    continue
                cells = []
                for T in ts:
                    try:
                        us, blk = rt.bench_gemm(rows, K, fmt, T, bool(hilo), shape, 24 if fmt else 12, 40)
                        flop = 2.0 * rows * K * T * (2 if hilo else 1)
                        cells.append(f"T={T}: {us:7.1f} us ({int(blk):4d} blk, {flop / us / 1e6:5.0f} TF)")
                    except Exception as e:                          # a shape the launch cannot take
                        cells.append(f"T={T}: n/a ({str(e)[:40]})")
                print(f"{mat:6s} fmt{fmt} hilo{hilo} shape {shape:2d} | " + " | ".join(cells), flush=True)
