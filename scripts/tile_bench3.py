"""Dev tool (GPU box): prefill tile shapes on steps of a few hundred rows (the 256-token chunk of BASELINE config #4).
profiles/r3_exp_tile_128x64.log was taken with two extra shapes (11 / 12: 128 rows x 64 tokens, 8 waves) that were removed afterwards."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ai00_server_amd import runtime as rt
cases = []
for T in (256, 512):
    cases += [(f"k3 int8 {T}", 10240, 2560, 1, T), (f"fkfr int8 {T}", 11520, 2560, 1, T), (f"wo int8 {T}", 2560, 2560, 1, T), (f"fv int8 {T}", 2560, 8960, 1, T),
              (f"k3 nf4 {T}", 10240, 2560, 2, T), (f"fkfr nf4 {T}", 10240 + 2560, 2560, 2, T), (f"k3 fp16 {T}", 10240, 2560, 0, T), (f"7b fk fp16 {T}", 14336 + 4096, 4096, 0, T)]
for name, rows, K, fmt, T in cases:
    out = []
    for shape in [int(x) for x in os.environ.get('SHAPES', '4,3,7').split(',')]:
        us, blk = rt.bench_gemm(rows, K, fmt, T, False, shape, 4, 20)
        out.append(f"s{shape}: {us:7.1f}us {2.0 * rows * K * T / us / 1e6:6.0f}TF {blk:.0f}blk")
    print(f"{name:14s} {rows}x{K} T={T} | " + " | ".join(out), flush=True)
