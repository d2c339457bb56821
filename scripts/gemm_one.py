import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ai00_server_amd import runtime as rt
rows, K, fmt, T, spb = [int(x) for x in sys.argv[1:6]]
us, blk = rt.bench_gemm(rows, K, fmt, T, False, spb, 8, 20)
print(rows, K, fmt, T, spb, f"{us:.2f}us {blk:.0f}blk")
