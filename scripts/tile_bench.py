import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ai00_server_amd import runtime as rt
for name,(rows,K,fmt) in {"w1":(160,2560,0),"wo-int8":(2560,2560,1),"rkvg-int8":(10240,2560,1),"fv-int8":(2560,8960,1),"rkvg-fp16":(10240,2560,0),"7b-fk-fp16":(14336,4096,0)}.items():
    for T in [int(x) for x in os.environ.get('TS', '128,512,1024').split(',')]:
        out=[]
        for shape in [int(x) for x in os.environ.get('SHAPES', '1,3,4,5').split(',')]:
            us, blk = rt.bench_gemm(rows,K,fmt,T,False,shape,4,20)
            tf = 2.0*rows*K*T/us/1e6
            out.append(f"s{shape}: {us:7.1f}us {tf:6.0f}TF {blk:.0f}blk")
        print(f"{name:10s} T={T:4d} | "+" | ".join(out), flush=True)
