"""Dev tool: strips-per-block (spb) and K-split (RWKV_KSB) sweep of the skinny GEMM for the 3B shapes."""
import sys, os, subprocess
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if len(sys.argv) > 1 and sys.argv[1] == "child":
    from ai00_server_amd import runtime as rt
    Ts = [int(x) for x in os.environ.get("TS", "32,16,8").split(",")]
    for name, (rows, K, fmt) in {"k3-int8": (10304, 2560, 1), "fkfr-int8": (11520, 2560, 1), "fv-int8": (2560, 8960, 1), "wo-int8": (2560, 2560, 1),
                                 "k3-f16": (10304, 2560, 0), "fv-f16": (2560, 8960, 0)}.items():
        for T in Ts:
            out = []
            for spb in (0, 1, 2, 3, 4, 6):
                try:
                    a, blk = rt.bench_gemm(rows, K, fmt, T, False, spb, 16, 96)
                    out.append(f"spb{spb}: {a:5.2f}us/{blk:.0f}blk")
                except Exception as e:
                    out.append(f"spb{spb}: err")
            print(f"{name} T={T}: " + " | ".join(out), flush=True)
else:
    for ksb in os.environ.get("KSBS", "0").split(","):
        print("RWKV_KSB =", ksb, flush=True)
        env = dict(os.environ)
        if ksb != "0":
            env["RWKV_KSB"] = ksb
        subprocess.run([sys.executable, __file__, "child"], env=env)
