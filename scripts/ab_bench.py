"""Dev tool (GPU box): A/B of whole decode steps.  Each variant = (label, extra env, library path); prints ms/step at B=32, 8, 1.
    python scripts/ab_bench.py "base::" "lean::ai00_server_amd/librwkv_hip_lean.so"
"""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for spec in sys.argv[1:]:
    label, envs, lib = (spec.split(":") + ["", ""])[:3]
    env = dict(os.environ)
    for kv in envs.split(","):
        if kv:
            k, v = kv.split("=")
            env[k] = v
    if lib:
        env["RWKV_HIP_LIB"] = os.path.join(ROOT, lib)
    wl = env.get("AB_WORKLOAD", "v6-3b")
    q = env.get("AB_QUANT", "int8")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--decode-only", "--no-cpu-baseline", "--sweep", "8,1", "--steps", "40",
                        "--warmup", "5", "--workload", wl, "--quant", q], env=env, capture_output=True, text=True)
    try:
        d = json.loads(r.stdout.strip().splitlines()[-1])
        sw = d["sweep"]
        print(f"{label:12s} B=32 {d['ms_per_step']:.3f} ms ({d['value']:.0f} tok/s, frac {d['roofline']['step']['frac_of_peak']:.3f})  "
              f"B=8 {sw['8']['ms_per_step']:.3f} ms  B=1 {sw['1']['ms_per_step']:.3f} ms  verified={d['tokens_verified']}", flush=True)
    except Exception as e:
        print(label, "FAILED", r.stdout[-500:], r.stderr[-1500:], flush=True)
