import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ai00_server_amd import runtime as rt
def wb(rows,K,fmt): return rows*K*{0:2.0,1:1.03125,2:0.53125}[fmt]
for name,(rows,K) in {"w1":(160,2560),"wo":(2560,2560),"rkvg":(10240,2560),"fkfr":(11520,2560),"fv":(2560,8960),"head":(65536,2560)}.items():
    for fmt in ([0] if name in("w1","head") else [1,0,2]):
        out=[]
        for T in (1,8,16,32):
            us, blk = rt.bench_gemm(rows, K, fmt, T, False, 0, max(2,min(64,int(600e6/wb(rows,K,fmt)))), 100 if rows<60000 else 30)
            out.append(f"T{T}: {us:6.2f}us {wb(rows,K,fmt)/us/1e3:5.0f}GB/s {blk:.0f}blk")
        print(f"{name:5s} fmt{fmt} | "+" | ".join(out), flush=True)
