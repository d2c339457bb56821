"""Dev tool (GPU box): tile shapes on grids that fill the chip an integral number of times (no tail round)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ai00_server_amd import runtime as rt
cases = [("512blk fp16", 8192, 2560, 0, 1024), ("1024blk fp16", 16384, 4096, 0, 1024), ("512blk int8", 8192, 2560, 1, 1024),
         ("1024blk int8", 16384, 4096, 1, 1024), ("512blk nf4", 8192, 2560, 2, 1024), ("2048blk fp16", 16384, 4096, 0, 2048),
         ("256blk fp16", 4096, 2560, 0, 1024), ("768blk fp16", 12288, 2560, 0, 1024),
         ("k3 int8 2048", 10240, 2560, 1, 2048), ("fkfr int8 2048", 11520, 2560, 1, 2048), ("fv int8 2048", 2560, 8960, 1, 2048), ("k3 nf4 2048", 10240, 2560, 2, 2048),
         ("k3 int8 1024", 10240, 2560, 1, 1024), ("k3 int8 512", 10240, 2560, 1, 512)]
for name, rows, K, fmt, T in cases:
    out = []
    for shape in [int(x) for x in os.environ.get('SHAPES', '4,7,10').split(',')]:
        if shape == 11 and fmt == 0:
            continue
        us, blk = rt.bench_gemm(rows, K, fmt, T, False, shape, 4, 20)
        out.append(f"s{shape}: {us:7.1f}us {2.0 * rows * K * T / us / 1e6:6.0f}TF {blk:.0f}blk")
    print(f"{name:14s} {rows}x{K} T={T} | " + " | ".join(out), flush=True)
