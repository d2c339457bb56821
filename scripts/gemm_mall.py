import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ai00_server_amd import runtime as rt
for name,(rows,K,fmt) in {"rkvg-int8":(10240,2560,1),"fkfr-int8":(11520,2560,1),"fv-int8":(2560,8960,1),"wo-int8":(2560,2560,1),"rkvg-fp16":(10240,2560,0)}.items():
    for T in (1,32):
        a,_ = rt.bench_gemm(rows,K,fmt,T,False,0,24,96)
        b,_ = rt.bench_gemm(rows,K,fmt,T,False,0,1,96)
        c,_ = rt.bench_gemm(rows,K,fmt,T,False,0,4,96)
        print(f"{name} T={T}: rotated(24 copies) {a:.2f}us | resident(1 copy) {b:.2f}us | 4 copies {c:.2f}us", flush=True)
