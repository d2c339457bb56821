"""rocprofv3 kernel trace -> per-launch-shape statistics (a kernel template serves several launches of a step: the head GEMM
and the Fk/Fr GEMM are the same `gemm_kernel` instantiation; they are told apart by grid size here).

    python scripts/summarize_trace.py <*_kernel_trace.csv> <out.csv> [--skip-load]
Writes: name, grid (workgroups), block, calls, total_us, avg_us, min_us, max_us, pct; sorted by total time.  `--skip-load`
drops the model-load kernels (quantisers, tilers, converters) so that the table is the steady-state step only.
"""
import collections, csv, sys

LOAD = ("quant_int8_kernel", "quant_nf4_kernel", "tile_f16_kernel", "f16_to_f32_kernel", "lora_blend_kernel", "__amd_rocclr")


def short(name: str) -> str:
    name = name.replace("void rwkv::", "").replace("rwkv::", "")
    for a, b in (("(rwkv::GemmLaunch)", ""), ("(rwkv::V6MixArgs)", ""), ("(rwkv::LnShiftArgs)", ""), ("(rwkv::WkvArgs)", ""),
                 ("(rwkv::EmbedArgs)", ""), ("(rwkv::LnOutArgs)", "")):
        name = name.replace(a, b)
    return name


def main():
    src, dst = sys.argv[1], sys.argv[2]
    skip = "--skip-load" in sys.argv
    acc = collections.OrderedDict()
    for r in csv.DictReader(open(src)):
        n = r["Kernel_Name"]
        if skip and any(x in n for x in LOAD):
            continue
        wg = int(r["Workgroup_Size_X"]) * int(r["Workgroup_Size_Y"]) * int(r["Workgroup_Size_Z"])
        grid = int(r["Grid_Size_X"]) * int(r["Grid_Size_Y"]) * int(r["Grid_Size_Z"]) // max(1, wg)
        d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
        a = acc.setdefault((short(n), grid, wg), [0, 0.0, 1e30, 0.0])
        a[0] += 1; a[1] += d; a[2] = min(a[2], d); a[3] = max(a[3], d)
    tot = sum(a[1] for a in acc.values())
    with open(dst, "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["kernel", "workgroups", "threads", "calls", "total_us", "avg_us", "min_us", "max_us", "pct"])
        for (n, g, wg), a in sorted(acc.items(), key=lambda kv: -kv[1][1]):
            w.writerow([n, g, wg, a[0], f"{a[1]:.1f}", f"{a[1] / a[0]:.2f}", f"{a[2]:.2f}", f"{a[3]:.2f}", f"{100 * a[1] / tot:.2f}"])


if __name__ == "__main__":
    main()
