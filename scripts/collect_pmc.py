"""HBM traffic of the decode step from rocprofv3 PMC counters (run on the GPU box; writes profiles/<round>_pmc_traffic.json).

Follows /opt/skills/guides/MI355X_MICROARCH.md "HBM" + "rocprofv3 PMC slots":
  * FETCH_SIZE and WRITE_SIZE in SEPARATE passes (TCC has 4 slots: FETCH_SIZE = 3, WRITE_SIZE = 2), kernel trace only;
  * unit = KB as reported; gfx950 correction: FETCH_SIZE x2 for wide coalesced streams (128-B requests tallied at 64 B);
  * calibration on a known stream of OUR access pattern: the head GEMM reads V x C fp16 weights exactly once.
    python scripts/collect_pmc.py [--round r1] [--batch 32] [--workload v6-3b] [--quant int8]
"""
import argparse, csv, json, os, subprocess, sys, collections
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

def run_pass(counter, tag, bench_args, parse_only=False):
    out = os.path.join(ROOT, "gpurun_out", f"pmc_{tag}")
    env = dict(os.environ, TMPDIR="/tmp")
    cmd = ["rocprofv3", "--kernel-trace", "--pmc", counter, "--output-format", "csv", "-d", out, "-o", "c", "--",
           sys.executable, os.path.join(ROOT, "bench.py"), "--decode-only", "--sweep", "", "--verify-steps", "0", "--steps", "12", "--warmup", "4"] + bench_args
    if not parse_only:
        import shutil
        shutil.rmtree(out, ignore_errors=True)
        subprocess.run(cmd, cwd="/tmp", env=env, check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    path = None
    for dp, _, fs in os.walk(out):
        for f in fs:
            if f == "c_counter_collection.csv":
                path = os.path.join(dp, f)
    acc = collections.defaultdict(lambda: [0.0, 0])
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != counter:
            continue
        a = acc[r["Kernel_Name"] + " grid=" + r["Grid_Size"]]     # one kernel serves several launch shapes
        a[0] += float(r["Counter_Value"]); a[1] += 1
    return acc

def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--round", default="r1"); ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--workload", default="v6-3b"); ap.add_argument("--quant", default="int8")
    ap.add_argument("--parse-only", action="store_true", help="re-parse gpurun_out/pmc_* without running rocprofv3")
    a = ap.parse_args()
    bargs = ["--batch", str(a.batch), "--workload", a.workload, "--quant", a.quant]
    fetch = run_pass("FETCH_SIZE", "fetch", bargs, a.parse_only)
    write = run_pass("WRITE_SIZE", "write", bargs, a.parse_only)
    def per_launch(acc, pred, scale):
        tot = sum(v[0] for k, v in acc.items() if pred(k)); n = sum(v[1] for k, v in acc.items() if pred(k))
        return (tot * 1024.0 * scale / n if n else None), n
    is_gemm = lambda k: ("gemm_kernel" in k or "v6_mix_kernel" in k)
    # head GEMM = the gemm launch shape with the largest fetch per launch (V x C fp16 once per step)
    heads = sorted(((v[0] / v[1], k) for k, v in fetch.items() if "gemm_kernel" in k), reverse=True)
    head_name = heads[0][1]
    layer = lambda k: is_gemm(k) and k != head_name
    rd, n = per_launch(fetch, layer, 2.0)
    wr, _ = per_launch(write, layer, 1.0)
    head_rd, hn = per_launch(fetch, lambda k: k == head_name, 2.0)
    res = {"config": {"workload": a.workload, "quant": a.quant, "batch": a.batch},
           "method": "rocprofv3 --kernel-trace --pmc FETCH_SIZE / WRITE_SIZE (separate passes), KB x1024; FETCH_SIZE x2 (gfx950)",
           "layer_gemm": {"launches": n, "read_bytes_per_launch": rd, "write_bytes_per_launch": wr,
                          "hbm_bytes_per_launch": (rd or 0) + (wr or 0)},
           "calibration": {"kernel": head_name, "launches": hn, "read_bytes_per_launch_x2": head_rd,
                           "expected": "V*C*2 bytes (65536 x 2560 fp16 = 335.5 MB for v6-3b)"},
           "per_kernel_read_bytes_per_launch_x2": {k[-100:]: v[0] * 2048.0 / v[1] for k, v in sorted(fetch.items(), key=lambda kv: -kv[1][0])[:12]},
           "per_kernel_write_bytes_per_launch": {k[-100:]: v[0] * 1024.0 / v[1] for k, v in sorted(write.items(), key=lambda kv: -kv[1][0])[:12]}}
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    name = f"{a.round}_pmc_traffic_{a.workload}_{a.quant}_b{a.batch}.json"
    for d in ("gpurun_out", "profiles"):
        json.dump(res, open(os.path.join(ROOT, d, name), "w"), indent=1)
    print(json.dumps(res, indent=1))

if __name__ == "__main__":
    main()
