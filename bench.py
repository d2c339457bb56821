#!/usr/bin/env python
"""bench.py — decode throughput of the RWKV hot path on MI355X, with roofline and CPU baseline.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload v6-3b] [--batch 32] [--quant int8]

A "step" is one decode step of the hot path over a batch of B slots (one token per slot: B tokens),
inputs (weights, recurrent state, token ids) resident in HBM: the arg-max token is fed back on the device
(`rwkv_decode_greedy`), so no PCIe traffic sits inside the timed region.

N > 1: one process per GPU, every rank an independent replica with its own weights, slots and stream (SURVEY 8e:
replicas only, no collective on the data path); weak scaling.  Launched either by the driver
(`python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N`: RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* come
from the environment) or by this file itself: `python bench.py --gpus N` with no WORLD_SIZE in the environment starts the N
ranks as child processes (gloo rendezvous on 127.0.0.1) and relays rank 0's line.  The barrier and the MAX of the timing go
over gloo on the host; `n_gpus` in the line is the world size that actually ran, and it must equal `--gpus`.

Legs of the default run (rank 0, N = 1; every leg carries its own roofline fraction and a verification flag):
  * the headline: BASELINE config #3 (V6-3B Int8, 32 slots) + sweep B = 1, 8, PCIe-inclusive and on-device-sampling rates,
    embeddings/s (256-token documents, state-only prefill, checked against a `Last` prefill of the same documents);
  * `configs`: #2 (V6-1.6B fp16, B = 1), V6-3B fp16 at B = 1 / 32, #4 (V7-2.9B NF4: B = 32 decode + `/embeddings` at
    token_chunk_size 256), #5 (V6-7B fp16: 8 x 4096-token prefill at chunk 2048 / 1024 + B = 8 decode);
  * `cpu_baseline`: the compiled restatement on the host cores (a reported baseline, never the target).

What this file takes from oracle/ (test infrastructure): the SYNTHETIC CHECKPOINT of the named shapes (`synth_st`,
`model_info`, `synth_prompt`: there is no network for real weights), the byte accounting of the roofline
(`algorithmic_bytes`), and the `cpu_baseline` leg.  The measured path never touches it: the engine is librwkv_hip.so
through ai00_server_amd.runtime.
"""
from __future__ import annotations

import argparse
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK = 8.0e12    # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured float4 copy)
MFMA_PEAK = 2.5e15   # dense fp16 / bf16 MFMA peak
EMBED_DOCS_PER_RANK = int(os.environ.get("BENCH_EMBED_DOCS", "512"))   # 4,096 documents over 8 replicas (test hook: fewer)
QT = {"none": 0, "int8": 1, "nf4": 2}
DTYPE = {"none": "f16", "int8": "u8->f16", "nf4": "nf4->f16"}


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--workload", default="v6-3b")
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--quant", default="int8", choices=["none", "int8", "nf4"])
    ap.add_argument("--precision", default="fp16", choices=["fp16", "fp32", "fp16raw"],
                    help="rwkv_load_desc.precision: fp16 = the library's default (ABI 7: f16 operands, the error-carrying launches hi + lo; holds 1e-3 at "
                         "32 layers), fp32 = hi + lo everywhere, fp16raw = f16 operands on every launch (fastest, outside 1e-3 at depth)")
    ap.add_argument("--repeats", type=int, default=5, help="timed regions of --steps decode steps each; the line reports the MEDIAN region (and min / max)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--decode-only", action="store_true", help="headline decode + roofline only (profiling runs)")
    ap.add_argument("--sweep", default="1,8", help="extra batch sizes reported under 'sweep' (north_star: batch 1-32)")
    ap.add_argument("--verify-steps", type=int, default=8, help="decode steps re-run through rwkv_infer + host arg-max and compared")
    ap.add_argument("--no-configs", action="store_true", help="skip the other BASELINE configurations (#2, #4, #5, V6-3B fp16)")
    ap.add_argument("--config5", action="store_true", help="kept for old command lines: config #5 is part of the default run")
    ap.add_argument("--selftest-dist", action="store_true",
                    help="exercise launcher + rendezvous + barrier + MAX reduction with a stand-in workload (no GPU, no engine): CPU test of the N > 1 path")
    return ap.parse_args(argv)


# ------------------------------------------------------------------------------------------------
# N > 1 without a launcher: start the ranks ourselves
# ------------------------------------------------------------------------------------------------
def free_port() -> int:
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def spawn_ranks(n: int, argv: list[str]) -> int:
    """`python bench.py --gpus N` outside torch.distributed.run: one child per GPU with the launcher's environment contract
    (RANK, LOCAL_RANK, WORLD_SIZE, MASTER_ADDR, MASTER_PORT).  Rank 0's stdout is ours; a failing rank fails the run."""
    port = free_port()
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv, env=env,
                                      stdout=None if r == 0 else subprocess.DEVNULL))
    # a rank that dies leaves the others parked in the gloo barrier: when one exits non-zero the rest (the exact children started
    # above) are terminated, so the run fails promptly instead of waiting for the rendezvous timeout
    rc, live = 0, list(procs)
    while live:
        for p in list(live):
            code = p.poll()
            if code is None:
                continue
            live.remove(p)
            if code != 0:
                rc = rc or code
                for q in live:
                    q.terminate()
        time.sleep(0.05)
    return rc


def gpu_numa_topology(n_gpus: int) -> dict:
    """{gpu index: {"numa_node": n, "cpus": [...]}} for the GPUs of this host, from the PCI device's sysfs `numa_node` and the node's
    `cpulist` (what `rocm-smi --showtoponuma` prints).  The GPU's PCI bus id comes from the HIP runtime (hipDeviceGetPCIBusId through
    ctypes: no torch import, no device context is created before the process has bound itself).  BENCH_FAKE_NUMA='{"0": [0, [0,1]], ...}'
    (gpu -> [node, cpus]) replaces the probe: the CPU test of the binding.  GPUs whose node is unknown (-1, single-node hosts, no
    sysfs) are left out: the rank then keeps the affinity it was started with."""
    fake = os.environ.get("BENCH_FAKE_NUMA")
    if fake:
        return {int(k): {"numa_node": int(v[0]), "cpus": [int(c) for c in v[1]]} for k, v in json.loads(fake).items()}
    topo = {}
    try:
        import ctypes
        hip = ctypes.CDLL("libamdhip64.so")
        for g in range(n_gpus):
            buf = ctypes.create_string_buffer(64)
            if hip.hipDeviceGetPCIBusId(buf, 64, g) != 0:
                continue
            bus = buf.value.decode().lower()
            node = int(open(f"/sys/bus/pci/devices/{bus}/numa_node").read().strip())
            if node < 0:
                continue
            cpus = []
            for part in open(f"/sys/devices/system/node/node{node}/cpulist").read().strip().split(","):
                lo, _, hi = part.partition("-")
                cpus.extend(range(int(lo), int(hi or lo) + 1))
            topo[g] = {"numa_node": node, "cpus": cpus, "pci_bus_id": bus}
    except (OSError, ValueError, AttributeError):
        pass
    return topo


def bind_to_gpu_numa_node(gpu: int, n_gpus: int) -> dict:
    """Pin this process (and every thread it starts later: OpenMP, the engine's callers) to the CPUs of its GPU's NUMA node BEFORE anything
    is allocated, so that the pinned arenas of `rwkv_host_alloc` (logits, embeddings) and the checkpoint pages are first touched on the
    node the GPU hangs off — on an 8-GPU host the PCIe-inclusive and embeddings legs otherwise cross the socket link.
    Returns what was done, for the line (`per_rank_numa`)."""
    info = {"gpu": gpu, "numa_node": None, "cpus_bound": None}
    t = gpu_numa_topology(n_gpus).get(gpu)
    if t:
        allowed = os.sched_getaffinity(0)
        cpus = sorted(set(t["cpus"]) & allowed)
        info["numa_node"] = t["numa_node"]
        if cpus:
            os.sched_setaffinity(0, cpus)
            info["cpus_bound"] = len(cpus)
            try:                                                      # memory policy follows first touch; prefer the node explicitly where libnuma exists
                import ctypes
                numa = ctypes.CDLL("libnuma.so.1")
                if numa.numa_available() >= 0:
                    numa.numa_set_preferred(t["numa_node"])
                    info["mempolicy"] = "preferred"
            except OSError:
                pass
    return info


def shared_synth_st(R, name: str, job):
    """The synthetic checkpoint ONCE PER NODE: local rank 0 generates the `.st` image and leaves it in /dev/shm, the other ranks map it
    (8 x 6 GB of numpy synthesis, and 8 private copies of the file image, become one).  Single-rank runs synthesise in place.
    The file is created exclusively under an unpredictable name (O_CREAT | O_EXCL | O_NOFOLLOW, mode 0600: nothing pre-planted in /dev/shm is
    followed or overwritten) and the name travels to the node's other ranks with the gather that doubles as the barrier behind the write; the
    decision to fall back is taken per NODE (its leader's record), and the leader unlinks the file in a `finally`, whatever a peer does."""
    if job.world == 1:
        return R.synth_st(name, fast=True)
    leader = job.local_rank_env == 0
    path, mine = None, None
    if leader:
        mine = R.synth_st(name, fast=True)
        cand = f"/dev/shm/rwkv_bench_{name}_{os.getuid()}_{os.urandom(8).hex()}.st"
        try:
            fd = os.open(cand, os.O_WRONLY | os.O_CREAT | os.O_EXCL | getattr(os, "O_NOFOLLOW", 0), 0o600)
            try:
                with open(fd, "wb", closefd=True) as f:
                    f.write(memoryview(mine[0]))
                path = cand
            except OSError:                                          # /dev/shm too small (a container's default is 64 MB)
                try:
                    os.unlink(cand)
                except OSError:
                    pass
        except OSError:                                              # /dev/shm absent / not writable
            path = None
    recs = job.gather_objects({"rank": job.rank, "local_rank": job.local_rank_env, "path": path})     # (doubles as the barrier behind the write)
    node_leader = job.rank - job.local_rank_env                      # ranks of a node are contiguous under torch.distributed.run and under spawn_ranks
    path = next((r["path"] for r in recs if r["rank"] == node_leader), None)
    if path is None:                                                 # this node synthesises per rank, as before round 5
        job.host_barrier()
        return mine if mine is not None else R.synth_st(name, fast=True)
    del mine
    try:
        img = np.memmap(path, dtype=np.uint8, mode="r")
        tensors = R.st_deserialize(img)
    finally:
        job.host_barrier()
        if leader:
            try:
                os.unlink(path)                                      # the mappings keep the pages; nothing is left behind
            except OSError:
                pass
    return img, tensors


class Job:
    """Rank bookkeeping + the host-side barrier / MAX reduction (gloo; there is no RCCL communicator in this repo)."""

    def __init__(self, args):
        self.rank = int(os.environ.get("RANK", "0"))
        self.local_rank = int(os.environ.get("LOCAL_RANK", "0"))
        self.local_rank_env = self.local_rank
        if os.environ.get("BENCH_SHARE_GPU"):       # test hook: all ranks on device 0 (the N > 1 code path on a 1-GPU box)
            self.local_rank = 0
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        if self.world != args.gpus:
            raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={self.world}: the line would report the wrong n_gpus")
        self.dist = None
        self.torch = None
        # first thing a rank does, before torch / the engine allocate anything: CPU affinity (and memory preference) of its GPU's NUMA node
        self.numa = bind_to_gpu_numa_node(self.local_rank, int(os.environ.get("LOCAL_WORLD_SIZE", str(self.world)))) \
            if self.world > 1 or os.environ.get("BENCH_FAKE_NUMA") else {"gpu": self.local_rank, "numa_node": None, "cpus_bound": None}
        if self.world > 1:
            import torch.distributed as dist
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            dist.init_process_group("gloo")         # replicas only
            self.dist = dist

    def use_gpu(self):
        import torch
        self.torch = torch
        if torch.cuda.is_available():
            torch.cuda.set_device(self.local_rank)

    def sync(self):
        if self.torch is not None and self.torch.cuda.is_available():
            self.torch.cuda.synchronize()

    def barrier(self):
        if self.dist is not None:
            self.dist.barrier()
        self.sync()

    def host_barrier(self):
        if self.dist is not None:
            self.dist.barrier()

    def gather_objects(self, obj):
        """[every rank's obj] (rank order)"""
        if self.dist is None:
            return [obj]
        out = [None] * self.world
        self.dist.all_gather_object(out, obj)
        return out

    def max_and_all(self, dt: float):
        """(MAX over ranks, [every rank's value])"""
        if self.dist is None:
            return dt, [dt]
        import torch
        x = torch.tensor([dt], dtype=torch.float64)
        xs = [torch.zeros(1, dtype=torch.float64) for _ in range(self.world)]
        self.dist.all_gather(xs, x)
        vals = [float(v.item()) for v in xs]
        return max(vals), vals

    def close(self):
        if self.dist is not None:
            self.dist.destroy_process_group()


# ------------------------------------------------------------------------------------------------
# legs
# ------------------------------------------------------------------------------------------------
PRECISIONS = {"fp16": "Fp16", "fp32": "Fp32", "fp16raw": "Fp16Raw"}
PRECISION_TEXT = {"fp16": "Precision::Fp16 (reload.rs:89-94; the library default, ABI 7: f16 GEMM operands, the launches that carry a model's operand-rounding "
                          "error read hi + lo — within 1e-3 at 32 layers, tests/test_gpu_full_depth.py)",
                  "fp32": "Precision::Fp32 (hi + lo f16 operands on every launch, fp32 accumulate)",
                  "fp16raw": "RWKV_PRECISION_FP16_RAW (f16 operands on every launch: fastest, 1.4e-3 (V6) / 4.7e-3 (V7) on the logits at 32 layers)"}


def build_engine(rt, st, local_rank, ql, qt, B, chunk, precision="fp16"):
    return (rt.ModelBuilder(st, adapter=local_rank).quant(ql, rt.Quant(qt))
            .build(max_batch=max(B, 1), token_chunk_size=chunk, precision=getattr(rt.Precision, PRECISIONS[precision])))


def first_tokens(R, V, B):
    return np.array([R.synth_prompt(s, 1)[0] % V for s in range(B)], dtype=np.uint32)


def verify_decode(rt, eng, first, n_steps):
    """The timed call's own output, checked: the same first steps from a zero state through rwkv_infer (logits to the host,
    arg-max there, run.rs:809-832) must give the ids rwkv_decode_greedy produced on the device, for every slot."""
    B = len(first)
    zero = eng.state.init()
    for b in range(B):
        eng.state.load(zero, b)
    dev_ids, _ = eng.decode_greedy(first, n_steps)
    for b in range(B):
        eng.state.load(zero, b)
    cur = [int(x) for x in first]
    host_ids = np.zeros((n_steps, B), dtype=np.int64)
    for s_ in range(n_steps):
        inp = rt.RnnInput([rt.RnnInputBatch([cur[b]] if b < B else [], rt.RnnOption.Last) for b in range(eng.max_batch)])
        _, outs = eng.infer(inp)
        cur = [int(np.argmax(outs[b][-1])) for b in range(B)]
        host_ids[s_] = cur
    ok = bool(np.array_equal(np.asarray(dev_ids, dtype=np.int64)[:, :B], host_ids))
    assert ok, "rwkv_decode_greedy ids differ from rwkv_infer + host arg-max"
    return ok


def decode_point(job, eng, first, steps, warmup):
    """EXACTLY `steps` timed decode steps bracketed by barrier + synchronize on both sides; MAX over ranks."""
    if warmup > 0:
        eng.decode_greedy(first, warmup)
    job.barrier()
    t = time.perf_counter()
    _, dev_ms = eng.decode_greedy(first, steps)
    job.sync()
    dt = time.perf_counter() - t
    job.barrier()
    dt_max, dt_all = job.max_and_all(dt)
    return dt_max, dt_all, dev_ms


def decode_regions(job, eng, first, steps, warmup, repeats):
    """`repeats` timed regions (each EXACTLY `steps` steps, bracketed like decode_point; the warm-up once in front of the first).  A 20-step
    region is 44 ms: one region's number moves by ~1 % from run to run, which is the size of most kernel A/B deltas —
    so the line carries the MEDIAN region and the spread.  Returns (median region as decode_point's triple, [ms_per_step of every region])."""
    regs = [decode_point(job, eng, first, steps, warmup if i == 0 else 0) for i in range(max(1, repeats))]
    order = sorted(range(len(regs)), key=lambda i: regs[i][0])
    med = regs[order[(len(regs) - 1) // 2]]
    return med, [r[0] * 1e3 / steps for r in regs]


def timed_on_all_ranks(job, fn, own_clock=False):
    """`fn()` bracketed by barrier + synchronize on both sides on every rank; returns (MAX over ranks, every rank's time, fn's result).
    `own_clock`: fn returns the seconds of its own timed region (the serving loops warm one step up before they start their clock); the
    ranks still enter together and the aggregate is still priced at the slowest rank."""
    job.barrier()
    t = time.perf_counter()
    res = fn()
    job.sync()
    dt = float(res) if own_clock else time.perf_counter() - t
    job.barrier()
    dt_max, dt_all = job.max_and_all(dt)
    return dt_max, dt_all, res


def embed_job_leg(job, rt, R, eng, info, n_docs, doc_len=256, verify=True):
    """Second half of BASELINE's metric, as SURVEY 8(d) config #4 specifies it: a batch of `doc_len`-token documents — 4,096 over 8
    replicas, i.e. `n_docs` = 512 per rank, sharded round-robin (document d belongs to rank d mod world) — prefilled state-only
    (RWKV_OPTION_NONE: no head GEMM, no logits) through `harness.StateJob`: GenerateKind::State requests with slot turnover, the layer
    slice of a finished document (docs/doc-api/openai.md:376-437) leaving on the engine's copy stream into pinned memory while the
    following documents prefill.  Every rank runs its shard inside one barrier-bracketed region; the rate is all documents over the
    slowest rank's time.  Verified on every rank: the first and the last `max_batch` documents of the shard equal, bit for bit, the
    same documents prefilled with `Last` (the reference-shaped call) in the same slots."""
    from ai00_server_amd.harness import StateJob, ReplicaRouter
    V, layer, B = info.num_vocab, info.num_layer - 1, eng.max_batch
    mine = list(ReplicaRouter.shard(n_docs * job.world, job.rank, job.world))           # global document ids of this rank
    docs = [[t % V for t in R.synth_prompt(100 + d, doc_len)] for d in mine]
    sj = StateJob(eng, layer)
    sj.run(docs[:B])                                                                    # warm: graphs of the step shapes, the arena
    dt_max, dt_all, (emb, calls) = timed_on_all_ranks(job, lambda: sj.run(docs))
    ok = bool(np.isfinite(emb).all() and np.abs(emb).max() > 0)
    if verify:
        zero = eng.state.init()
        for lo in sorted({0, max(0, len(docs) - B)}):
            group = docs[lo:lo + B]
            for b in range(len(group)):
                eng.state.load(zero, b)
            inp = rt.RnnInput([rt.RnnInputBatch(list(group[b]) if b < len(group) else [], rt.RnnOption.Last) for b in range(B)])
            while inp.num_token() > 0:
                inp, _ = eng.infer(inp)
            ok = ok and all(np.array_equal(eng.state.embed(layer, b), emb[lo + b]) for b in range(len(group)))
        assert ok, "state-only (RWKV_OPTION_NONE) embeddings of the job differ from a `Last` prefill of the same documents"
    total = len(docs) * job.world
    res = {"value": total / dt_max, "unit": "embeddings/s", "doc_tokens": doc_len, "docs": total, "docs_per_rank": len(docs),
           "prefill_tokens_per_s": total * doc_len / dt_max, "per_rank_embeddings_per_s": [len(docs) / d for d in dt_all],
           "infer_calls_per_rank": calls, "token_chunk_size": eng.token_chunk_size, "slots": B, "embeddings_verified": ok if verify else None,
           "checksum": float(np.abs(emb[:B]).astype(np.float64).sum()),
           "job": "harness.StateJob: slot turnover, rwkv_state_back_layer_async into pinned memory (read-back overlaps the next documents' prefill)",
           "embedding": f"layer {layer} WKV rows [64 x {info.num_emb}]"}
    sj.close()
    return res


def roofline_leg(rt, R, eng, info, shapes, first, ms_per_step, workload, quant, step_frac, ab):
    """Dominant kernel family (the layer GEMMs): per-launch hipEvent timing on the engine's stream (eager step, rwkv_profile_infer),
    calibrated against the graph-replayed step so that the family times tile the step like rocprofv3's kernel durations."""
    B, V = len(first), info.num_vocab
    fam_ms, nprof = {}, 5
    for it in range(nprof + 1):
        inp = rt.RnnInput([rt.RnnInputBatch([int(first[b])] if b < B else [], rt.RnnOption.Last) for b in range(eng.max_batch)])
        _, _, fam = eng.profile_infer(inp)
        if it == 0:
            continue        # first pass warms caches / clocks
        for k, (ms, n) in fam.items():
            a = fam_ms.setdefault(k, [0.0, 0])
            a[0] += ms
            a[1] += n
    g_ms, g_n = fam_ms["gemm_layers"]
    h_ms, h_n = fam_ms["gemm_head"]
    head_bytes = V * info.num_emb * 2
    vec_bytes = sum(int(np.prod(s)) * 2 for k, s in shapes.items() if len([d for d in s if d > 1]) <= 1)
    layer_gemm_bytes = eng.weight_bytes - head_bytes - vec_bytes          # weights the layer-GEMM family streams (LoRA matrices included)
    # An event pair adds marker processing (about 2 us) to what it brackets: the pairs of a step sum to more than the step itself
    # takes.  The excess over the graph-replayed step, spread evenly over the launches, is subtracted from every launch.
    n_launch = sum(v[1] for v in fam_ms.values()) / nprof
    pairs_ms = sum(v[0] for v in fam_ms.values()) / nprof
    marker_us = max(0.0, (pairs_ms - ms_per_step) / n_launch * 1e3)
    g_step_ms = g_ms / nprof - marker_us * 1e-3 * (g_n / nprof)
    achieved = layer_gemm_bytes / (g_step_ms * 1e-3)
    # HBM traffic per launch from the PMC passes (scripts/collect_pmc.py on the same workload; committed under profiles/).
    traffic, traffic_src = None, None
    import glob
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", f"*_pmc_traffic_{workload}_{quant}_b{B}.json"))):
        try:
            traffic = json.load(open(f))["layer_gemm"]["hbm_bytes_per_launch"]
            traffic_src = "profiles/" + os.path.basename(f)
        except Exception:
            pass
    per_layer = (g_n // nprof) // max(1, info.num_layer)
    return {"bound": "hbm",
            "kernel": f"layer GEMM family: gemm_kernel (time-mix projections + decay LoRA, output, channel-mix key/receptance, channel-mix value)"
                      f"{' + v6_mix_kernel (token-shift LoRA)' if int(info.version) == 6 else ''}, {per_layer} launches per layer",
            "achieved": achieved / 1e9, "peak": HBM_PEAK / 1e9, "unit": "GB/s", "frac": achieved / HBM_PEAK, "traffic": traffic,
            "traffic_unit": "HBM bytes per launch (PMC FETCH_SIZE x2 + WRITE_SIZE)", "traffic_source": traffic_src,
            "algorithmic_bytes_per_launch": layer_gemm_bytes / max(1, g_n // nprof),
            "launches_per_step": g_n // nprof, "avg_launch_us": g_step_ms / (g_n / nprof) * 1e3,
            "avg_launch_us_raw_event_pairs": g_ms / g_n * 1e3, "event_pair_overhead_us": marker_us,
            "bytes_per_step": layer_gemm_bytes,
            "head_gemm_GBps": head_bytes / max(1e-9, (h_ms / nprof - marker_us * 1e-3 * (h_n / nprof)) * 1e-3) / 1e9,
            "family_ms_per_step": {k: v[0] / nprof - marker_us * 1e-3 * (v[1] / nprof) for k, v in fam_ms.items()},
            "step": {"bytes": ab["per_step"], "frac_of_peak": step_frac, "W_q": ab["W_q"], "S": ab["S"]}}


def config_leg(job, rt, R, name, quant, batches, steps, verify_steps, embed_chunk=None, prefill=None, raw=False):
    """One other BASELINE configuration on its own engine: decode at `batches` (each with its whole-step fraction of 8 TB/s), the
    largest batch verified against rwkv_infer + host arg-max; optionally an embeddings leg at `embed_chunk`; optionally a long
    prefill (`prefill` = (prompt_tokens, [chunks])) priced against the MFMA peak."""
    st, tensors = R.synth_st(name, fast=True)
    info = R.model_info(tensors)
    shapes = {k: v.shape for k, v in tensors.items()}
    del tensors
    qt = QT[quant]
    ql = info.num_layer if qt else 0
    B = max(batches)
    out = {"workload": f"RWKV-{name} {quant}", "dtype": DTYPE[quant] + "/f32acc", "precision": PRECISION_TEXT["fp16"], "decode": {}}
    t0 = time.time()
    eng = build_engine(rt, st, job.local_rank, ql, qt, B, max(2048, B) if prefill is None else max(prefill[1]))
    out["load_s"] = time.time() - t0
    V = info.num_vocab
    first = first_tokens(R, V, B)
    for nb in batches:
        dt, _, _ = decode_point(job, eng, first[:nb], steps, min(5, steps))
        ab = R.algorithmic_bytes(info, shapes, ql, qt, nb)
        out["decode"][str(nb)] = {"tokens_per_s": nb * steps / dt, "ms_per_step": dt * 1e3 / steps,
                                  "frac_of_hbm_peak": ab["per_step"] * (steps / dt) / HBM_PEAK}
    out["tokens_verified"] = verify_decode(rt, eng, first, verify_steps) if verify_steps > 0 else None
    if prefill is not None:
        n_tok, chunks = prefill
        flops_tok = 2.0 * sum(int(np.prod(v)) for k, v in shapes.items() if k != "emb.weight")
        docs = [[t % V for t in R.synth_prompt(500 + b, n_tok)] for b in range(B)]
        out["prefill"] = {}
        for ch in chunks:
            e = eng if ch == eng.token_chunk_size else build_engine(rt, st, job.local_rank, ql, qt, B, ch)
            best, last = None, None
            for rep in range(2):
                z = e.state.init()
                for b in range(B):
                    e.state.load(z, b)
                job.sync()
                t = time.perf_counter()
                inp = rt.RnnInput([rt.RnnInputBatch(list(docs[b]), rt.RnnOption.Last) for b in range(B)])
                while inp.num_token() > 0:
                    inp, outs = e.infer(inp)
                d = time.perf_counter() - t
                best = d if best is None else min(best, d)
                ids = [int(np.argmax(outs[b][-1])) for b in range(B)]
                assert last is None or ids == last, "prefill is not reproducible run to run"
                last = ids
            out["prefill"][str(ch)] = {"tokens_per_s": B * n_tok / best, "s": best, "TFLOPs": B * n_tok * flops_tok / best / 1e12,
                                       "frac_of_mfma_peak": B * n_tok * flops_tok / best / MFMA_PEAK, "last_token_ids": last}
            if e is not eng:
                e.close()
        ids = [v["last_token_ids"] for v in out["prefill"].values()]
        out["prefill_chunks_agree"] = all(i == ids[0] for i in ids)      # chunking is exact for an RNN up to GEMM summation order
        out["prefill_workload"] = f"{B} x {n_tok}-token prompts, MFMA-bound: flops = 2 x (params - embedding) per token against {MFMA_PEAK / 1e15} PFLOP/s"
    eng.close()
    if embed_chunk is not None:
        e = build_engine(rt, st, job.local_rank, ql, qt, B, embed_chunk)
        out["embeddings"] = embed_job_leg(job, rt, R, e, info, EMBED_DOCS_PER_RANK)
        e.close()
    if raw:                                                  # the all-f16 opt-in next to the default: what the tolerance-holding default costs
        e = build_engine(rt, st, job.local_rank, ql, qt, B, embed_chunk or max(2048, B), "fp16raw")
        dt, _, _ = decode_point(job, e, first, steps, min(5, steps))
        out["fp16_raw"] = {"precision": PRECISION_TEXT["fp16raw"], "decode": {str(B): {"tokens_per_s": B * steps / dt, "ms_per_step": dt * 1e3 / steps}},
                           "default_vs_raw_ms_per_step": out["decode"][str(B)]["ms_per_step"] / (dt * 1e3 / steps),
                           "tokens_verified": verify_decode(rt, e, first, verify_steps) if verify_steps > 0 else None}
        if embed_chunk is not None:
            out["fp16_raw"]["embeddings"] = embed_job_leg(job, rt, R, e, info, EMBED_DOCS_PER_RANK)
            out["fp16_raw"]["embeddings"]["default_vs_raw_rate"] = out["embeddings"]["value"] / out["fp16_raw"]["embeddings"]["value"]
        e.close()
    return out


def cpu_leg(R, tensors, ql, qt, first, workload, quant):
    """CPU leg ("port"): the SAME configuration on the host cores through the compiled restatement oracle/cpu_backend.c (tests/test_oracle.py
    holds it against the numpy restatement; numpy itself if it cannot be built).  A baseline to stand next to the GPU number, never the target."""
    B = len(first)
    cur = [int(x) for x in first]
    host = host_description()
    try:
        from oracle.cpu_backend import CpuBackend
        ref = CpuBackend(tensors, ql, qt)
        how = ref.describe() if hasattr(ref, "describe") else f"C/OpenMP restatement (oracle/cpu_backend.c), {ref.threads} threads"
        cores = ref.threads
    except (NotImplementedError, OSError, subprocess.CalledProcessError):
        CpuBackend = None
        ref = R.RwkvRefBatch(tensors)
        how, cores = "numpy/BLAS fp32 oracle on all host cores, weights fp16-rounded (unquantised on the CPU side)", os.cpu_count()
    st_cpu = ref.init_states(B)
    ref.step(cur, st_cpu, want_logits=False)       # warm
    n_step, t1 = 0, time.time()
    while time.time() - t1 < 12.0 and n_step < 256:
        lg = ref.step(cur, st_cpu)
        cur = [int(x) for x in np.argmax(lg, axis=1)]
        n_step += 1
    cdt = time.time() - t1
    cpu = {"value": B * n_step / cdt, "unit": "tokens/s", "cores": cores, "kind": "port", "host": host,
           "sample": f"{n_step} lock-step decode steps of {B} slots ({B * n_step} tokens), {how}, {workload} {quant}"}
    if hasattr(ref, "stream_gbps"):
        cpu["effective_weight_stream_GBps"] = ref.stream_gbps(n_step / cdt)
        cpu["note"] = ("a stated baseline, not a tuned CPU implementation: at this batch the C GEMM (fp16 -> fp32 convert + FMA per slot) is compute-bound, "
                       "it streams weights well below what the same cores reach at batch 1 (config1_effective_weight_stream_GBps)")
    del ref, st_cpu
    # BASELINE config #1: RWKV-V5-World-0.4B fp16, batch 1, greedy, on the CPU path (the reference has no CPU backend,
    # lib.rs:339-368; this is the port)
    try:
        _, t5 = R.synth_st("v5-0.4b", fast=True)
        r5 = CpuBackend(t5) if CpuBackend is not None else R.RwkvRefBatch(t5)
        s5 = r5.init_states(1)
        lg = r5.step([int(first[0]) % r5.info.num_vocab], s5)
        n5, t1 = 0, time.time()
        while time.time() - t1 < 5.0 and n5 < 2048:
            lg = r5.step([int(np.argmax(lg[0]))], s5)
            n5 += 1
        d5 = time.time() - t1
        cpu["config1_v5_0.4b_b1_tokens_per_s"] = n5 / d5
        if hasattr(r5, "stream_gbps"):
            cpu["config1_effective_weight_stream_GBps"] = r5.stream_gbps(n5 / d5)
        del r5, t5
    except MemoryError:
        pass
    return cpu


def host_description() -> dict:
    """CPU model / sockets / cores of the host the CPU leg runs on (SURVEY 8d asks for them next to the number)."""
    d = {"logical_cpus": os.cpu_count()}
    try:
        txt = subprocess.run(["lscpu"], capture_output=True, text=True, timeout=5).stdout
        for line in txt.splitlines():
            k, _, v = line.partition(":")
            k, v = k.strip(), v.strip()
            if k == "Model name":
                d["model"] = v
            elif k == "Socket(s)":
                d["sockets"] = int(v)
            elif k == "Core(s) per socket":
                d["cores_per_socket"] = int(v)
            elif k == "Thread(s) per core":
                d["threads_per_core"] = int(v)
            elif k == "NUMA node(s)":
                d["numa_nodes"] = int(v)
    except Exception:
        pass
    return d


# ------------------------------------------------------------------------------------------------
def selftest_dist(job, args):
    """Stand-in workload for the CPU test of the N > 1 path: same barrier / timing / reduction / line, no engine."""
    if os.environ.get("BENCH_SELFTEST_FAIL_RANK") == str(job.rank):   # test hook: this rank dies mid-run
        os._exit(3)
    job.barrier()
    t = time.perf_counter()
    time.sleep(0.002 * args.steps * (1 + job.rank))          # ranks finish at different times: MAX must pick the slowest
    dt = time.perf_counter() - t
    job.barrier()
    dt_max, dt_all = job.max_and_all(dt)
    numa = job.gather_objects(dict(job.numa, rank=job.rank, affinity=sorted(os.sched_getaffinity(0))))
    if job.rank == 0:
        B = args.batch
        print(json.dumps({"per_rank_numa": numa,
                          "metric": "decode tokens/sec (whole job)", "value": B * job.world * args.steps / dt_max, "unit": "tokens/s",
                          "n_gpus": job.world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt_max * 1e3 / args.steps,
                          "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "selftest", "data": "none",
                          "config": {"workload": "launcher self-test (sleep)"},
                          "per_rank_tokens_per_s": [B * args.steps / d for d in dt_all], "selftest": True,
                          # the legs every rank of a real N > 1 run adds (same aggregation: all units over the slowest rank's time)
                          "pcie_inclusive_tokens_per_s": B * job.world * args.steps / dt_max,
                          "pcie_inclusive": {"value": B * job.world * args.steps / dt_max, "per_rank": [B * args.steps / d for d in dt_all]},
                          "on_device_sampling_tokens_per_s": B * job.world * args.steps / dt_max,
                          "on_device_sampling": {"value": B * job.world * args.steps / dt_max, "per_rank": [B * args.steps / d for d in dt_all]},
                          "embeddings": {"value": 8 * job.world / dt_max, "unit": "embeddings/s", "docs": 8 * job.world, "docs_per_rank": 8,
                                         "per_rank_embeddings_per_s": [8 / d for d in dt_all]}}), flush=True)


def main(argv=None):
    argv = list(sys.argv[1:] if argv is None else argv)
    args = parse_args(argv)
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        return spawn_ranks(args.gpus, argv)
    job = Job(args)
    if args.selftest_dist:
        selftest_dist(job, args)
        job.close()
        return 0
    job.use_gpu()
    rank, world = job.rank, job.world

    from ai00_server_amd import runtime as rt
    from oracle import rwkv_ref as R   # checkpoint synthesis + byte accounting + the cpu_baseline leg only

    t0 = time.time()
    st, tensors = shared_synth_st(R, args.workload, job)
    info = R.model_info(tensors)
    shapes = {k: v.shape for k, v in tensors.items()}
    qt = QT[args.quant]
    ql = info.num_layer if qt else 0
    B = args.batch
    t_synth = time.time() - t0

    t0 = time.time()
    eng = build_engine(rt, st, job.local_rank, ql, qt, B, max(2048, B), args.precision)
    t_load = time.time() - t0
    V = info.num_vocab
    first = first_tokens(R, V, B)

    (dt, dt_all, dev_ms), region_ms = decode_regions(job, eng, first, args.steps, args.warmup, args.repeats)
    ms_per_step = dt * 1e3 / args.steps
    value = B * world * args.steps / dt

    tokens_verified = verify_decode(rt, eng, first, args.verify_steps) if rank == 0 and args.verify_steps > 0 else None
    ab = R.algorithmic_bytes(info, shapes, ql, qt, B)
    step_frac = ab["per_step"] * (args.steps / dt) / HBM_PEAK
    single = rank == 0 and world == 1
    roof = roofline_leg(rt, R, eng, info, shapes, first, ms_per_step, args.workload, args.quant, step_frac, ab) if rank == 0 else None

    sweep = {}
    if single and args.sweep:
        for nb in [int(x) for x in args.sweep.split(",") if x]:
            if nb > eng.max_batch:
                continue
            d2, _, _ = decode_point(job, eng, first[:nb], args.steps, min(args.warmup, 10))
            abn = R.algorithmic_bytes(info, shapes, ql, qt, nb)
            sweep[str(nb)] = {"tokens_per_s": nb * args.steps / d2, "ms_per_step": d2 * 1e3 / args.steps,
                              "frac_of_peak": abn["per_step"] * (args.steps / d2) / HBM_PEAK}

    # On EVERY rank (the N > 1 line carries the whole metric, not only the device-resident loop — the host side is where replica
    # scaling can break): the PCIe-inclusive rate through rwkv_infer (logits of every slot D2H every token, as
    # run.rs:809-832 receives them) — never `value`; the serving path with the on-device sampling front-end (rwkv_infer_sample: 8 bytes
    # per slot over PCIe); and the embeddings job.  Each is bracketed by barrier + synchronize like `value`, aggregated over the
    # slowest rank, with every rank's own rate beside it.
    pcie = sampled = emb = None
    if not args.decode_only:
        nst = max(5, min(40, args.steps))
        d_max, d_all, _ = timed_on_all_ranks(job, lambda: eng.serve_loop_logits(first, nst), own_clock=True)
        pcie = {"value": B * nst * world / d_max, "unit": "tokens/s", "per_rank": [B * nst / d for d in d_all], "steps": nst}
        if V <= 65536:
            d_max, d_all, _ = timed_on_all_ranks(job, lambda: eng.serve_loop_sample(first, nst), own_clock=True)
            sampled = {"value": B * nst * world / d_max, "unit": "tokens/s", "per_rank": [B * nst / d for d in d_all], "steps": nst}
        # the embeddings job at SURVEY config #4's token_chunk_size (256 tokens per rwkv_infer call): a second engine over the same
        # checkpoint, since the chunk is a load-time parameter (ReloadRequest::token_chunk_size, lib.rs:221-223)
        e256 = build_engine(rt, st, job.local_rank, ql, qt, B, 256, args.precision)
        emb = embed_job_leg(job, rt, R, e256, info, EMBED_DOCS_PER_RANK)
        e256.close()
        if single:
            # the same job on the 2048-row engine of the decode legs (what a deployment that sets a large chunk gets)
            e = embed_job_leg(job, rt, R, eng, info, EMBED_DOCS_PER_RANK)
            # chunking is exact for an RNN up to the summation order of the GEMM that a chunk size selects
            assert abs(e["checksum"] - emb["checksum"]) <= 2e-3 * abs(emb["checksum"]), "embeddings depend on token_chunk_size"
            emb["at_token_chunk_size_2048"] = {"value": e["value"], "prefill_tokens_per_s": e["prefill_tokens_per_s"],
                                               "embeddings_verified": e["embeddings_verified"]}
    eng.close()

    # The other two modes on the record, same configuration, same checks (reload.rs:89-94 has Fp16 and Fp32; ABI 7 adds the all-f16 opt-in):
    # Precision::Fp32 — hi + lo f16 operands on every launch, measured <= 3e-5 at 32 layers — and RWKV_PRECISION_FP16_RAW — the fastest,
    # outside 1e-3 at depth.  What the tolerance-holding default costs is `default_vs_raw_ms_per_step`.
    others = None
    if single and not args.decode_only and args.precision == "fp16":
        others = {}
        for mode in ("fp32", "fp16raw"):
            eo = build_engine(rt, st, job.local_rank, ql, qt, B, max(2048, B), mode)
            rec = {"precision": PRECISION_TEXT[mode], "decode": {}}
            for nb in sorted({B, 8, 1}, reverse=True):
                if nb > B:
                    continue
                d2, _, _ = decode_point(job, eo, first[:nb], args.steps, min(args.warmup, 10))
                abn = R.algorithmic_bytes(info, shapes, ql, qt, nb)
                rec["decode"][str(nb)] = {"tokens_per_s": nb * args.steps / d2, "ms_per_step": d2 * 1e3 / args.steps,
                                          "frac_of_hbm_peak": abn["per_step"] * (args.steps / d2) / HBM_PEAK}
            rec["tokens_verified"] = verify_decode(rt, eo, first, args.verify_steps) if args.verify_steps > 0 else None
            rec["vs_default_ms_per_step"] = rec["decode"][str(B)]["ms_per_step"] / ms_per_step
            eo.close()
            eo = build_engine(rt, st, job.local_rank, ql, qt, B, 256, mode)
            rec["embeddings"] = embed_job_leg(job, rt, R, eo, info, EMBED_DOCS_PER_RANK)
            eo.close()
            if emb:
                rec["embeddings"]["vs_default_rate"] = rec["embeddings"]["value"] / emb["value"]
            others[mode] = rec
    del st

    cpu = None
    if single and not args.no_cpu_baseline and not args.decode_only:
        cpu = cpu_leg(R, tensors, ql, qt, first, args.workload, args.quant)
    del tensors

    configs = None
    if single and not args.no_configs and not args.decode_only:
        vs, cs = args.verify_steps, min(args.steps, 50)
        configs = {}
        for key, kw in [("config2_v6-1.6b_fp16", dict(name="v6-1.6b", quant="none", batches=[1])),
                        ("v6-3b_fp16", dict(name="v6-3b", quant="none", batches=[1, 32])),
                        ("config4_v7-2.9b_nf4", dict(name="v7-2.9b", quant="nf4", batches=[1, 32], embed_chunk=256, raw=True)),
                        ("config5_v6-7b_fp16", dict(name="v6-7b", quant="none", batches=[8], prefill=(4096, [2048, 1024])))]:
            configs[key] = config_leg(job, rt, R, steps=cs, verify_steps=vs, **kw)

    numa = job.gather_objects(dict(job.numa, rank=rank))
    if rank == 0:
        line = {"metric": "decode tokens/sec (whole job)", "value": value, "unit": "tokens/s", "n_gpus": world,
                "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_per_step, "higher_is_better": True,
                "scaling": "weak", "vs_baseline": None, "dtype": DTYPE[args.quant] + "/f32acc", "data": "synthetic",
                "config": {"workload": f"RWKV-{args.workload} {args.quant} decode, batch={B}/GPU, greedy, state+tokens resident in HBM",
                           "quant": args.quant, "batch_per_gpu": B, "precision": PRECISION_TEXT[args.precision],
                           "parallelism": f"replicas x{world} (no collective)"},
                "tokens_per_s_per_gpu": value / world, "per_rank_tokens_per_s": [B * args.steps / d for d in dt_all],
                "per_rank_numa": numa,
                "device_ms_per_step": dev_ms / args.steps,
                "roofline": roof, "cpu_baseline": cpu, "embeddings": emb, "other_precisions": others,
                "timed_regions": {"n": len(region_ms), "steps_each": args.steps, "ms_per_step": region_ms, "median": ms_per_step,
                                  "min": min(region_ms), "max": max(region_ms),
                                  "note": "`value` / `ms_per_step` are the MEDIAN region's (each region: exactly --steps steps, barrier + synchronize on both sides)"},
                "pcie_inclusive_tokens_per_s": pcie["value"] if pcie else None, "pcie_inclusive": pcie,
                "on_device_sampling_tokens_per_s": sampled["value"] if sampled else None, "on_device_sampling": sampled,
                "sweep": sweep or None, "tokens_verified": tokens_verified,
                "configs": configs, "load_s": t_load, "synth_s": t_synth}
        print(json.dumps(line), flush=True)
    job.close()
    return 0


if __name__ == "__main__":
    sys.exit(main())
