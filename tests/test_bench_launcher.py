"""`bench.py --gpus N` must really run N ranks (replicas only, one process per GPU, gloo barrier + MAX of the
timing on the host).  CPU: the launcher, rendezvous, barrier, reduction and the JSON line with a stand-in workload
(`--selftest-dist`), started both ways the driver may start it.  GPU (-m gpu): the real bench as two ranks sharing the
one device of the test box (`BENCH_SHARE_GPU=1`)."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def clean_env(**extra):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env.update(extra)
    return env


def last_json(out: str) -> dict:
    return json.loads([l for l in out.strip().splitlines() if l.startswith("{")][-1])


def test_gpus_2_without_a_launcher_spawns_two_ranks():
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--steps", "10", "--selftest-dist"], env=clean_env(),
                       capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr[-2000:]
    d = last_json(r.stdout)
    assert d["n_gpus"] == 2 and len(d["per_rank_tokens_per_s"]) == 2
    # MAX over ranks: rank 1 sleeps twice as long, so the aggregate is priced at the slower rank's time
    assert d["per_rank_tokens_per_s"][1] < d["per_rank_tokens_per_s"][0]
    assert abs(d["value"] - 2 * min(d["per_rank_tokens_per_s"])) <= 1e-6 * d["value"]
    assert len([l for l in r.stdout.splitlines() if l.startswith("{")]) == 1          # ONE line, from rank 0
    # the N > 1 line carries the WHOLE metric (BASELINE: decode tokens/s + embeddings/s), each leg per rank and aggregated over the
    # slowest rank like `value`
    for key in ("pcie_inclusive", "on_device_sampling"):
        assert len(d[key]["per_rank"]) == 2 and abs(d[key]["value"] - 2 * min(d[key]["per_rank"])) <= 1e-6 * d[key]["value"]
        assert d[key + "_tokens_per_s"] == d[key]["value"]
    e = d["embeddings"]
    assert e["unit"] == "embeddings/s" and len(e["per_rank_embeddings_per_s"]) == 2 and e["docs"] == 2 * e["docs_per_rank"]
    assert abs(e["value"] - 2 * min(e["per_rank_embeddings_per_s"])) <= 1e-6 * e["value"]


def test_each_rank_binds_to_the_numa_node_of_its_gpu():
    """8-GPU readiness: before anything is allocated a rank pins itself to the CPUs of its GPU's NUMA node, so the pinned
    arenas of `rwkv_host_alloc` are node-local; the line says what every rank did (`per_rank_numa`).  Faked topology: GPU 0 on node 0
    with the first allowed CPU, GPU 1 on node 1 with the last one."""
    cpus = sorted(os.sched_getaffinity(0))
    fake = {"0": [0, [cpus[0]]], "1": [1, [cpus[-1]]]}
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--steps", "5", "--selftest-dist"], env=clean_env(BENCH_FAKE_NUMA=json.dumps(fake)),
                       capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr[-2000:]
    numa = last_json(r.stdout)["per_rank_numa"]
    assert [n["rank"] for n in numa] == [0, 1] and [n["gpu"] for n in numa] == [0, 1]
    assert [n["numa_node"] for n in numa] == [0, 1] and [n["cpus_bound"] for n in numa] == [1, 1]
    assert numa[0]["affinity"] == [cpus[0]] and numa[1]["affinity"] == [cpus[-1]]
    # a GPU the topology does not know keeps the affinity it was started with
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--steps", "5", "--selftest-dist"], env=clean_env(BENCH_FAKE_NUMA=json.dumps({"0": fake["0"]})),
                       capture_output=True, text=True, timeout=120)
    numa = last_json(r.stdout)["per_rank_numa"]
    assert numa[1]["numa_node"] is None and numa[1]["affinity"] == cpus


def test_the_checkpoint_is_synthesised_once_per_node_and_mapped_by_the_other_ranks(tmp_path):
    """N > 1: local rank 0 generates the `.st` image into /dev/shm, the others map it (bench.shared_synth_st); every rank ends up with the
    same bytes and the same tensors, nothing is left in /dev/shm; when the image cannot be written there every rank falls back to its own copy."""
    script = tmp_path / "ranks.py"
    script.write_text(f"""
import hashlib, json, os, sys
sys.path.insert(0, {ROOT!r})
import numpy as np
import bench
from oracle import rwkv_ref as R
args = bench.parse_args(["--gpus", "2"])
job = bench.Job(args)
if os.environ.get("BREAK_SHM") and job.local_rank_env == 0:
    real_open = os.open
    def broken(path, *a, **k):
        if str(path).startswith("/dev/shm/"):
            raise OSError(28, "No space left on device")
        return real_open(path, *a, **k)
    bench.os.open = broken
img, tens = bench.shared_synth_st(R, "v6-tiny", job)
own, _ = R.synth_st("v6-tiny", fast=True)
info = R.model_info(tens)
print(json.dumps({{"rank": job.rank, "sha": hashlib.sha256(bytes(memoryview(np.ascontiguousarray(img)))).hexdigest(), "same_as_own": bool(np.array_equal(np.asarray(img), own)),
                  "layers": int(info.num_layer), "mapped": isinstance(img, np.memmap)}}), flush=True)
job.close()
""")
    for broken in (False, True):
        procs = []
        for r in range(2):
            env = clean_env(RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE="2", LOCAL_WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(29641 + int(broken)))
            if broken:
                env["BREAK_SHM"] = "1"
            procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
        outs = [p.communicate(timeout=180) for p in procs]
        assert all(p.returncode == 0 for p in procs), outs
        recs = sorted((last_json(o[0]) for o in outs), key=lambda d: d["rank"])
        assert recs[0]["sha"] == recs[1]["sha"] and all(r["same_as_own"] and r["layers"] > 0 for r in recs)
        assert [r["mapped"] for r in recs] == ([False, False] if broken else [True, True])
        assert not [f for f in os.listdir("/dev/shm") if f.startswith("rwkv_bench_v6-tiny")]


def test_world_size_8_the_rank_count_the_metric_names():
    """BASELINE's metric is quoted at 1 / 2 / 4 / 8 GPUs; no 8-GPU node has been available to any round, so the N = 8 path is exercised here with the
    stand-in workload: eight ranks spawned by the bench itself on a faked two-node topology (GPUs 0-3 on node 0, 4-7 on node 1), every rank bound
    to its node's CPUs, one JSON line from rank 0 priced at the slowest rank, every leg of the metric aggregated over eight ranks.  The shared
    checkpoint at eight ranks is the next test."""
    cpus = sorted(os.sched_getaffinity(0))
    half = max(1, len(cpus) // 2)
    fake = {str(g): [g // 4, (cpus[:half] if g < 4 else cpus[half:] or cpus[:half])] for g in range(8)}
    r = subprocess.run([sys.executable, BENCH, "--gpus", "8", "--steps", "4", "--selftest-dist"], env=clean_env(BENCH_FAKE_NUMA=json.dumps(fake)),
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    assert len([l for l in r.stdout.splitlines() if l.startswith("{")]) == 1
    d = last_json(r.stdout)
    assert d["n_gpus"] == 8 and len(d["per_rank_tokens_per_s"]) == 8 and d["scaling"] == "weak"
    assert abs(d["value"] - 8 * min(d["per_rank_tokens_per_s"])) <= 1e-6 * d["value"]            # eight ranks' tokens over the slowest rank's time
    numa = d["per_rank_numa"]
    assert [n["rank"] for n in numa] == list(range(8)) and [n["gpu"] for n in numa] == list(range(8))
    assert [n["numa_node"] for n in numa] == [0, 0, 0, 0, 1, 1, 1, 1]
    assert all(set(n["affinity"]) <= set(fake[str(n["gpu"])][1]) for n in numa)
    e = d["embeddings"]
    assert len(e["per_rank_embeddings_per_s"]) == 8 and e["docs"] == 8 * e["docs_per_rank"]
    for key in ("pcie_inclusive", "on_device_sampling"):
        assert len(d[key]["per_rank"]) == 8


def test_the_shared_checkpoint_at_eight_ranks(tmp_path):
    """`shared_synth_st` at the rank count of a full node: one synthesis, eight mappings of the same bytes, nothing left in /dev/shm."""
    script = tmp_path / "ranks8.py"
    script.write_text(f"""
import hashlib, json, os, sys
sys.path.insert(0, {ROOT!r})
import numpy as np
import bench
from oracle import rwkv_ref as R
job = bench.Job(bench.parse_args(["--gpus", "8"]))
img, tens = bench.shared_synth_st(R, "v6-tiny", job)
print(json.dumps({{"rank": job.rank, "sha": hashlib.sha256(bytes(memoryview(np.ascontiguousarray(img)))).hexdigest(), "mapped": isinstance(img, np.memmap)}}), flush=True)
job.close()
""")
    procs = []
    for r in range(8):
        env = clean_env(RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE="8", LOCAL_WORLD_SIZE="8", MASTER_ADDR="127.0.0.1", MASTER_PORT="29647")
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    outs = [p.communicate(timeout=300) for p in procs]
    assert all(p.returncode == 0 for p in procs), [o[1][-500:] for o in outs]
    recs = sorted((last_json(o[0]) for o in outs), key=lambda d: d["rank"])
    assert len({r["sha"] for r in recs}) == 1 and all(r["mapped"] for r in recs)
    assert not [f for f in os.listdir("/dev/shm") if f.startswith("rwkv_bench_v6-tiny")]


def test_under_torch_distributed_run_as_the_driver_launches_it():
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", "29631", BENCH, "--gpus", "2", "--steps", "10", "--selftest-dist"], env=clean_env(),
                       capture_output=True, text=True, timeout=180)
    assert r.returncode == 0, r.stderr[-2000:]
    assert last_json(r.stdout)["n_gpus"] == 2


def test_a_failing_rank_fails_the_run_promptly():
    """rank 1 dies before the rendezvous completes its first barrier: the launcher must come back with its exit code instead of
    leaving rank 0 parked in the barrier."""
    import time
    t = time.time()
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--steps", "10", "--selftest-dist"], env=clean_env(BENCH_SELFTEST_FAIL_RANK="1"),
                       capture_output=True, text=True, timeout=120)
    assert r.returncode != 0 and time.time() - t < 60


def test_world_size_and_gpus_flag_must_agree():
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--selftest-dist"], env=clean_env(WORLD_SIZE="1", RANK="0"),
                       capture_output=True, text=True, timeout=60)
    assert r.returncode != 0 and "WORLD_SIZE" in (r.stderr + r.stdout)


@pytest.mark.gpu
def test_real_bench_as_two_ranks_on_one_device():
    """Two ranks sharing the test box's one device: decode, the serving loops (logits over PCIe; on-device sampling) and the embeddings
    job (64 documents per rank here instead of 512) on BOTH ranks, each leg aggregated over the slower rank."""
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--steps", "10", "--warmup", "3", "--no-configs", "--no-cpu-baseline", "--sweep", "",
                        "--verify-steps", "4"], env=clean_env(BENCH_SHARE_GPU="1", BENCH_EMBED_DOCS="64"), capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    d = last_json(r.stdout)
    assert d["n_gpus"] == 2 and len(d["per_rank_tokens_per_s"]) == 2 and d["tokens_verified"] is True
    assert [n["rank"] for n in d["per_rank_numa"]] == [0, 1] and all("numa_node" in n and "cpus_bound" in n for n in d["per_rank_numa"])
    assert d["value"] > 0 and d["config"]["parallelism"].startswith("replicas x2")
    for key in ("pcie_inclusive", "on_device_sampling"):
        assert len(d[key]["per_rank"]) == 2 and 0 < d[key]["value"] <= 2 * min(d[key]["per_rank"]) * (1 + 1e-6)
    e = d["embeddings"]
    assert e["docs"] == 128 and e["docs_per_rank"] == 64 and len(e["per_rank_embeddings_per_s"]) == 2 and e["embeddings_verified"] is True
    assert e["token_chunk_size"] == 256 and 0 < e["value"] <= 2 * min(e["per_rank_embeddings_per_s"]) * (1 + 1e-6)


def test_committed_bench_line_keeps_the_contract():
    """The line the driver parses (profiles/r6_bench_default.json is the last one measured on the GPU box): every key of the bench
    contract, the roofline and cpu_baseline objects with their fields, value = tokens of all ranks / the slowest rank's time."""
    import json
    path = os.path.join(ROOT, "profiles", "r6_bench_default.json")
    line = json.loads(open(path).readline())
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config"):
        assert k in line, k
    assert line["higher_is_better"] is True and line["scaling"] == "weak" and line["vs_baseline"] is None and line["data"] == "synthetic"
    assert "workload" in line["config"] and "model" not in line["config"]
    rf = line["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in rf, k
    assert rf["bound"] == "hbm" and rf["unit"] == "GB/s" and abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-9
    assert rf["traffic"] is None or rf["traffic"] >= rf["algorithmic_bytes_per_launch"]      # HBM bytes cannot undercut the algorithmic ones
    cb = line["cpu_baseline"]
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in cb, k
    assert cb["kind"] in ("port", "reference") and cb["cores"] >= 1
    B = line["config"]["batch_per_gpu"]
    assert abs(line["value"] - line["n_gpus"] * B * 1000.0 / line["ms_per_step"]) / line["value"] < 1e-6
    assert line["tokens_verified"] is True and line["embeddings"]["embeddings_verified"] is True
    e = line["embeddings"]                                  # 512 documents x 256 tokens per rank at token_chunk_size 256
    assert e["unit"] == "embeddings/s" and e["docs_per_rank"] == 512 and e["doc_tokens"] == 256 and e["token_chunk_size"] == 256
    assert len(e["per_rank_embeddings_per_s"]) == line["n_gpus"] and len(line["pcie_inclusive"]["per_rank"]) == line["n_gpus"]
    # round 6: the headline runs in the library's default precision (ABI 7: holds 1e-3 at depth); Precision::Fp32 and the all-f16 opt-in are on the
    # line too, each verified like the headline; the timed region is repeated and the line carries the median region
    assert "Precision::Fp16" in line["config"]["precision"]
    for mode in ("fp32", "fp16raw"):
        f = line["other_precisions"][mode]
        assert f["tokens_verified"] is True and f["embeddings"]["embeddings_verified"] is True and set(f["decode"]) == {"32", "8", "1"}
    assert line["configs"]["config4_v7-2.9b_nf4"]["fp16_raw"]["tokens_verified"] is True
    tr = line["timed_regions"]
    assert tr["n"] >= 5 and tr["min"] <= tr["median"] <= tr["max"] and abs(tr["median"] - line["ms_per_step"]) < 1e-9
    assert len(line["per_rank_numa"]) == line["n_gpus"]


def test_committed_roofline_table_is_what_the_script_generates():
    """profiles/r6_roofline_table.md (and rounds 5 and 4's) is GENERATED from the committed rocprofv3 summaries and launch logs
    (scripts/roofline_table.py): the per-launch fractions DESIGN.md quotes cannot drift from the evidence without this test noticing.  The
    script asserts every fraction <= 1 and never prices two launches that differ in kernel, grid or block size as one row (round 4's table
    had a 1.074: two 256-workgroup launches of different block sizes joined on the grid alone)."""
    import re
    for rnd in ("r6", "r5", "r4"):
        r = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "roofline_table.py"), rnd], capture_output=True, text=True, timeout=120)
        assert r.returncode == 0, r.stderr[-2000:]
        assert r.stdout == open(os.path.join(ROOT, "profiles", f"{rnd}_roofline_table.md")).read()
        assert "(HBM)" in r.stdout and "(MFMA)" in r.stdout
        fracs = [float(x) for x in re.findall(r"\| (\d\.\d+) \((?:HBM|MFMA)\) \|", r.stdout)]
        assert fracs and max(fracs) <= 1.0
    for rnd in ("r6", "r5"):                                                                           # since round 5 the logs carry the block size
        assert "ambiguous" not in open(os.path.join(ROOT, "profiles", f"{rnd}_roofline_table.md")).read()
    assert "v6_mix_apply_kernel" in open(os.path.join(ROOT, "profiles", "r6_roofline_table.md")).read()     # both launches of the split mix are priced


def test_roofline_table_never_prices_two_different_launches_as_one_row(tmp_path):
    """The join of scripts/roofline_table.py on a fixture: two decode launches with the SAME grid and different block sizes (round 4's
    1.074 row) are priced separately when the launch log carries block sizes, and reported as ambiguous — not priced — when it does not;
    a row kernel gets its bytes from a `row` log entry; a fraction above 1 makes the script fail instead of printing it."""
    script = os.path.join(ROOT, "scripts", "roofline_table.py")
    csv_text = ("kernel,workgroups,threads,calls,total_us,avg_us,min_us,max_us,pct\n"
                '"gemm_kernel<1, 8, false, true, false>(GemmLaunch)",256,640,100,2000.0,20.00,19.0,21.0,50.0\n'
                '"gemm_kernel<1, 8, false, true, false>(GemmLaunch)",256,512,100,1000.0,10.00,9.0,11.0,25.0\n'
                '"ln_shift_kernel<1, 1024>(LnShiftArgs)",8,1024,200,1000.0,5.00,4.0,6.0,25.0\n')
    big = {"kind": "decode", "variant": 1, "T": 8, "grid": 256, "threads": 640, "ksplit": 1, "rows": 4096, "bytes": 100_000_000, "flops": 1e9, "mats": "blocks.0.ffn.value.weight"}
    small = dict(big, threads=512, bytes=30_000_000, mats="blocks.0.att.output.weight")
    row = {"kind": "row", "kernel": "ln_shift_kernel", "T": 8, "grid": 8, "bytes": 1_000_000}

    def run(entries, name="tX"):
        d = tmp_path / name
        d.mkdir()
        (d / f"{name}_kernel_stats_v6-7b_fp16_b8.csv").write_text(csv_text)
        (d / f"{name}_launch_log_v6-7b_fp16_b8.jsonl").write_text("\n".join(json.dumps(e) for e in entries) + "\n")
        return subprocess.run([sys.executable, script, name, str(d)], capture_output=True, text=True, timeout=60)

    r = run([big, small, row], "ta")
    assert r.returncode == 0, r.stderr
    lines = [l for l in r.stdout.splitlines() if l.startswith("| `")]
    assert "ffn.value" in lines[0] and "att.output" not in lines[0] and "100.0 MB" in lines[0] and "0.625 (HBM)" in lines[0]      # 100 MB / 20 us = 5 TB/s
    assert "att.output" in lines[1] and "ffn.value" not in lines[1] and "30.0 MB" in lines[1] and "0.375 (HBM)" in lines[1]
    assert "1.0 MB" in lines[2] and "(HBM)" in lines[2]
    legacy = [{k: v for k, v in e.items() if k != "threads"} for e in (big, small)]
    r = run(legacy, "tb")
    assert r.returncode == 0 and r.stdout.count("ambiguous") == 2 and "MB |" not in "".join(l for l in r.stdout.splitlines() if "gemm_kernel" in l)
    r = run([dict(big, bytes=400_000_000), small], "tc")                       # 400 MB in 20 us = 20 TB/s: impossible, the script must refuse
    assert r.returncode != 0 and "AssertionError" in r.stderr
