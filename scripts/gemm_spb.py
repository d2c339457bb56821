import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ai00_server_amd import runtime as rt
for name,(rows,K,fmt) in {"rkvg-int8":(10240,2560,1),"fv-int8":(2560,8960,1),"wo-int8":(2560,2560,1),"w1":(160,2560,0)}.items():
    for T in (1,32):
        out=[]
        for spb in (0,1,2,3,4):
            a,blk = rt.bench_gemm(rows,K,fmt,T,False,spb,16,96)
            out.append(f"spb{spb}: {a:.2f}us/{blk:.0f}blk")
        print(f"{name} T={T}: "+" | ".join(out), flush=True)
