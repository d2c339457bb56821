#!/usr/bin/env python
"""bench.py — decode throughput of the RWKV hot path on MI355X, with roofline and CPU baseline.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload v6-3b] [--batch 32] [--quant int8]

A "step" is one decode step of the hot path over a batch of B slots (one token per slot: B tokens),
inputs (weights, recurrent state, token ids) resident in HBM: the arg-max token is fed back on the device
(`rwkv_decode_greedy`), so no PCIe traffic sits inside the timed region.  N > 1: one process per GPU
(launched by torch.distributed.run), every rank is an independent replica with its own weights, slots and
stream (replicas only, no collective on the data path); weak scaling.

What this file takes from oracle/ (test infrastructure): the SYNTHETIC CHECKPOINT of the named shapes (`synth_st`,
`model_info`, `synth_prompt`: there is no network for real weights), the byte accounting of the roofline
(`algorithmic_bytes`), and the `cpu_baseline` leg (the numpy restatement timed on the host cores).  The measured
path never touches it: the engine is librwkv_hip.so through ai00_server_amd.runtime.
"""
from __future__ import annotations

import argparse
import json
import subprocess
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK = 8.0e12   # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured float4 copy)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--workload", default="v6-3b")
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--quant", default="int8", choices=["none", "int8", "nf4"])
    ap.add_argument("--precision", default="fp16", choices=["fp16", "fp32"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--decode-only", action="store_true", help="skip the embeddings / PCIe / sampling / CPU legs (profiling)")
    ap.add_argument("--sweep", default="1,8", help="extra batch sizes reported under 'sweep' (north_star: batch 1-32)")
    ap.add_argument("--verify-steps", type=int, default=8, help="decode steps re-run through rwkv_infer + host arg-max and compared")
    ap.add_argument("--config5", action="store_true", help="also run BASELINE config #5 (V6-7B fp16, 8 x 4096-token prefill at chunk 2048, "
                                                           "then 256 decode steps at batch 8) and report it under 'config5'")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if os.environ.get("BENCH_SHARE_GPU"):       # test hook: all ranks on device 0 (validates the N>1 code path on a 1-GPU box)
        local_rank = 0
    world = int(os.environ.get("WORLD_SIZE", "1"))
    import torch
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(local_rank)
        dist.init_process_group("gloo")     # replicas only: the barrier and the MAX of the timing are host-side, no RCCL
    elif torch.cuda.is_available():
        torch.cuda.set_device(local_rank)

    from ai00_server_amd import runtime as rt
    from oracle import rwkv_ref as R   # checkpoint synthesis + the cpu_baseline leg only

    t0 = time.time()
    st, tensors = R.synth_st(args.workload, fast=True)
    info = R.model_info(tensors)
    shapes = {k: v.shape for k, v in tensors.items()}
    qt = {"none": 0, "int8": 1, "nf4": 2}[args.quant]
    ql = info.num_layer if qt else 0
    B = args.batch
    t_synth = time.time() - t0

    t0 = time.time()
    eng = (rt.ModelBuilder(st, adapter=local_rank).quant(ql, rt.Quant(qt))
           .build(max_batch=max(B, 1), token_chunk_size=max(2048, B),
                  precision=rt.Precision.Fp32 if args.precision == "fp32" else rt.Precision.Fp16))
    t_load = time.time() - t0

    V = info.num_vocab
    first = np.array([R.synth_prompt(s, 1)[0] % V for s in range(B)], dtype=np.uint32)

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    def timed(nb, steps, warmup):
        ft = first[:nb]
        if warmup > 0:
            eng.decode_greedy(ft, warmup)
        barrier()
        t = time.perf_counter()
        toks, dev_ms = eng.decode_greedy(ft, steps)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t
        barrier()
        if dist is not None:
            x = torch.tensor([dt], dtype=torch.float64, device="cpu")
            dist.all_reduce(x, op=dist.ReduceOp.MAX)
            dt = float(x.item())
        return dt, dev_ms, toks

    dt, dev_ms, toks = timed(B, args.steps, args.warmup)
    ms_per_step = dt * 1e3 / args.steps
    value = B * world * args.steps / dt

    # ---- the timed call's own output, checked: the same first steps from a zero state through rwkv_infer (logits to the
    # host, arg-max there, run.rs:809-832) must give the ids rwkv_decode_greedy produced on the device, for every slot
    tokens_verified = None
    if rank == 0 and args.verify_steps > 0:
        zero = eng.state.init()
        for b in range(B):
            eng.state.load(zero, b)
        dev_ids, _ = eng.decode_greedy(first, args.verify_steps)
        for b in range(B):
            eng.state.load(zero, b)
        cur = [int(x) for x in first]
        host_ids = np.zeros((args.verify_steps, B), dtype=np.int64)
        for s_ in range(args.verify_steps):
            inp = rt.RnnInput([rt.RnnInputBatch([cur[b]] if b < B else [], rt.RnnOption.Last) for b in range(eng.max_batch)])
            _, outs = eng.infer(inp)
            cur = [int(np.argmax(outs[b][-1])) for b in range(B)]
            host_ids[s_] = cur
        tokens_verified = bool(np.array_equal(np.asarray(dev_ids, dtype=np.int64)[:, :B], host_ids))
        assert tokens_verified, "rwkv_decode_greedy ids differ from rwkv_infer + host arg-max"

    ab = R.algorithmic_bytes(info, shapes, ql, qt, B)
    step_frac = ab["per_step"] * (args.steps / dt) / HBM_PEAK

    # ---- roofline of the dominant kernel (the layer GEMMs): per-launch hipEvent timing on the engine stream
    roof = None
    if rank == 0:
        inp_tokens = [[int(first[b])] for b in range(B)]
        fam_ms = {}
        nprof = 5
        for it in range(nprof + 1):
            inp = rt.RnnInput([rt.RnnInputBatch(list(inp_tokens[b]) if b < B else [], rt.RnnOption.Last)
                               for b in range(eng.max_batch)])
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
        layer_gemm_bytes = eng.weight_bytes - head_bytes - vec_bytes          # weights the layer GEMM launches stream
        # The event pair around a launch adds marker processing (about 2 us) to what it brackets: the pairs of a step sum to more
        # than the step itself takes.  Calibration: the excess over the graph-replayed step (`ms_per_step`, the timed region
        # above), spread evenly over the launches, is subtracted from every launch — rocprofv3's kernel durations tile the step
        # the same way (profiles/*_kernel_stats_*.csv: their sum equals the step), so the calibrated averages agree with them.
        n_launch = sum(v[1] for v in fam_ms.values()) / nprof
        pairs_ms = sum(v[0] for v in fam_ms.values()) / nprof
        marker_us = max(0.0, (pairs_ms - ms_per_step) / n_launch * 1e3)
        g_step_ms = g_ms / nprof - marker_us * 1e-3 * (g_n / nprof)           # layer-GEMM family time inside one step
        achieved = layer_gemm_bytes / (g_step_ms * 1e-3)
        # HBM traffic per launch from the PMC passes (scripts/collect_pmc.py on the same workload; committed under
        # profiles/).  Not collectable inside this process: counters need rocprofv3 around the run.
        traffic, traffic_src = None, None
        import glob
        for f in sorted(glob.glob(os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles",
                                               f"*_pmc_traffic_{args.workload}_{args.quant}_b{B}.json"))):
            try:
                traffic = json.load(open(f))["layer_gemm"]["hbm_bytes_per_launch"]
                traffic_src = "profiles/" + os.path.basename(f)
            except Exception:
                pass
        roof = {"bound": "hbm", "kernel": "gemm_kernel (layer projections)", "achieved": achieved / 1e9,
                "peak": HBM_PEAK / 1e9, "unit": "GB/s", "frac": achieved / HBM_PEAK, "traffic": traffic,
                "traffic_unit": "HBM bytes per launch (PMC FETCH_SIZE x2 + WRITE_SIZE)", "traffic_source": traffic_src,
                "algorithmic_bytes_per_launch": layer_gemm_bytes / max(1, g_n // nprof),
                "launches_per_step": g_n // nprof, "avg_launch_us": g_step_ms / (g_n / nprof) * 1e3,
                "avg_launch_us_raw_event_pairs": g_ms / g_n * 1e3, "event_pair_overhead_us": marker_us,
                "bytes_per_step": layer_gemm_bytes,
                "head_gemm_GBps": head_bytes / max(1e-9, (h_ms / nprof - marker_us * 1e-3 * (h_n / nprof)) * 1e-3) / 1e9,
                "family_ms_per_step": {k: v[0] / nprof - marker_us * 1e-3 * (v[1] / nprof) for k, v in fam_ms.items()},
                "step": {"bytes": ab["per_step"], "frac_of_peak": step_frac, "W_q": ab["W_q"], "S": ab["S"]}}

    sweep = {}
    if rank == 0 and args.sweep:
        for nb in [int(x) for x in args.sweep.split(",") if x]:
            if nb > eng.max_batch:
                continue
            d2, _, _ = timed(nb, args.steps, min(args.warmup, 10)) if dist is None else (None, None, None)
            if d2:
                abn = R.algorithmic_bytes(info, shapes, ql, qt, nb)
                sweep[str(nb)] = {"tokens_per_s": nb * args.steps / d2, "ms_per_step": d2 * 1e3 / args.steps,
                                  "frac_of_peak": abn["per_step"] * (args.steps / d2) / HBM_PEAK}

    # PCIe-inclusive rate through rwkv_infer (logits of every slot D2H every token, as run.rs:809-832 receives them) — never
    # `value`.  Tight loops (runtime.serve_loop_*): one ABI call per step, nothing else on the host.
    pcie = None
    if rank == 0 and world == 1 and not args.decode_only:
        nst = max(5, min(40, args.steps))
        pcie = B * nst / eng.serve_loop_logits(first, nst)

    # serving path with the on-device sampling front-end (rwkv_infer_sample: nucleus defaults, 8 bytes/slot over PCIe)
    sampled = None
    if rank == 0 and world == 1 and V <= 65536 and not args.decode_only:
        nst = max(5, min(40, args.steps))
        sampled = B * nst / eng.serve_loop_sample(first, nst)

    # second half of BASELINE's metric: embeddings/s = documents prefilled (256 tokens each, one per slot) and read
    # back as one layer's WKV rows (rwkv_state_back_layer) per second, same engine, rank 0 only
    emb = None
    if rank == 0 and world == 1 and not args.decode_only:
        doc_len, layer = 256, info.num_layer - 1
        docs = [[t % V for t in R.synth_prompt(100 + b, doc_len)] for b in range(B)]

        def embed_rate(e):
            zero = e.state.init()
            best = None
            for rep in range(3):
                for b in range(B):
                    e.state.load(zero, b)
                t = time.perf_counter()
                inp = rt.RnnInput([rt.RnnInputBatch(list(docs[b]) if b < B else [], rt.RnnOption.NoOutput)   # state-only: no head GEMM, no logits
                                   for b in range(e.max_batch)])
                while inp.num_token() > 0:
                    inp, _ = e.infer(inp)
                vecs = [e.state.embed(layer, b) for b in range(B)]
                dt_e = time.perf_counter() - t
                best = dt_e if best is None else min(best, dt_e)
            return best

        best = embed_rate(eng)
        emb = {"value": B / best, "unit": "embeddings/s", "doc_tokens": doc_len, "docs": B,
               "prefill_tokens_per_s": B * doc_len / best, "token_chunk_size": eng.token_chunk_size,
               "embedding": f"layer {layer} WKV rows [64 x {info.num_emb}] via rwkv_state_back_layer"}
        # the same job at SURVEY config #4's token_chunk_size (256 tokens per rwkv_infer call): a second engine over the same
        # checkpoint, since the chunk is a load-time parameter (ReloadRequest::token_chunk_size, lib.rs:221-223)
        e256 = (rt.ModelBuilder(st, adapter=local_rank).quant(ql, rt.Quant(qt))
                .build(max_batch=max(B, 1), token_chunk_size=256,
                       precision=rt.Precision.Fp32 if args.precision == "fp32" else rt.Precision.Fp16))
        b256 = embed_rate(e256)
        e256.close()
        emb["at_token_chunk_size_256"] = {"value": B / b256, "prefill_tokens_per_s": B * doc_len / b256}
    del st

    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline and not args.decode_only:
        # CPU leg ("port"): the SAME configuration on the host cores — the same slots in lock step, the same (fake-)quantised fp16
        # weight values the GPU dequantises to — through the compiled restatement oracle/cpu_backend.c (C + OpenMP: threaded fp16
        # GEMM with fp32 accumulation; tests/test_oracle.py holds it against the numpy restatement; numpy itself if it cannot be built).
        # It is a baseline to stand next to the GPU number, never the target.
        cur = [int(x) for x in first]
        try:
            from oracle.cpu_backend import CpuBackend
            ref = CpuBackend(tensors, ql, qt)
            how = (f"C/OpenMP restatement (oracle/cpu_backend.c), fp16 weights ({'fake-quantised ' + args.quant if qt else 'unquantised'}), "
                   f"fp32 accumulate, {ref.threads} threads")
            cores = ref.threads
        except (NotImplementedError, OSError, subprocess.CalledProcessError):
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
        cpu = {"value": B * n_step / cdt, "unit": "tokens/s", "cores": cores, "kind": "port",
               "sample": f"{n_step} lock-step decode steps of {B} slots ({B * n_step} tokens), {how}, {args.workload}"}
        del ref, st_cpu
        # BASELINE config #1: RWKV-V5-World-0.4B fp16, batch 1, greedy, on the CPU path (the reference has no CPU backend,
        # lib.rs:339-368; this is the port)
        try:
            _, t5 = R.synth_st("v5-0.4b", fast=True)
            try:
                r5 = CpuBackend(t5)
            except (NameError, NotImplementedError, OSError, subprocess.CalledProcessError):
                r5 = R.RwkvRefBatch(t5)
            s5 = r5.init_states(1)
            lg = r5.step([int(first[0]) % r5.info.num_vocab], s5)
            n5, t1 = 0, time.time()
            while time.time() - t1 < 5.0 and n5 < 1024:
                lg = r5.step([int(np.argmax(lg[0]))], s5)
                n5 += 1
            cpu["config1_v5_0.4b_b1_tokens_per_s"] = n5 / (time.time() - t1)
            del r5, t5
        except MemoryError:
            pass
    eng.close()

    # BASELINE config #5 (optional leg): RWKV-V6-World-7B fp16, batch 8, 4096-token prompts (token_chunk_size 2048: one 2048-row
    # step per call), then streamed decode.  Prefill is bounded by the MFMA rate: algorithmic flops = 2 * (params - embedding) per
    # token against the 2.5 PFLOP/s dense fp16 peak; decode by HBM as above.
    cfg5 = None
    if rank == 0 and world == 1 and args.config5:
        st7, t7 = R.synth_st("v6-7b", fast=True)
        i7 = R.model_info(t7)
        sh7 = {k: v.shape for k, v in t7.items()}
        flops_tok = 2.0 * sum(int(np.prod(v)) for k, v in sh7.items() if k != "emb.weight")
        e7 = rt.ModelBuilder(st7, adapter=local_rank).build(max_batch=8, token_chunk_size=2048, precision=rt.Precision.Fp16)
        e7b = rt.ModelBuilder(st7, adapter=local_rank).build(max_batch=8, token_chunk_size=1024, precision=rt.Precision.Fp16)   # SURVEY's chunk for config #5
        del st7, t7
        docs = [[t % i7.num_vocab for t in R.synth_prompt(500 + b, 4096)] for b in range(8)]
        best = None
        for rep in range(2):
            z = e7.state.init()
            for b in range(8):
                e7.state.load(z, b)
            torch.cuda.synchronize()
            t = time.perf_counter()
            inp = rt.RnnInput([rt.RnnInputBatch(list(docs[b]), rt.RnnOption.Last) for b in range(8)])
            calls = 0
            while inp.num_token() > 0:
                inp, outs = e7.infer(inp)
                calls += 1
            dt7 = time.perf_counter() - t
            best = dt7 if best is None else min(best, dt7)
        def prefill7(e):
            bst = None
            for rep in range(2):
                z = e.state.init()
                for b in range(8):
                    e.state.load(z, b)
                torch.cuda.synchronize()
                t = time.perf_counter()
                inp = rt.RnnInput([rt.RnnInputBatch(list(docs[b]), rt.RnnOption.Last) for b in range(8)])
                while inp.num_token() > 0:
                    inp, _ = e.infer(inp)
                d = time.perf_counter() - t
                bst = d if bst is None else min(bst, d)
            return bst
        f7 = np.array([int(np.argmax(outs[b][-1])) for b in range(8)], dtype=np.uint32)
        e7.decode_greedy(f7, 8)
        _, dms = e7.decode_greedy(f7, 256)
        ab7 = R.algorithmic_bytes(i7, sh7, 0, 0, 8)
        cfg5 = {"workload": "RWKV-v6-7b fp16, batch 8, 4096-token prompts, token_chunk_size 2048, then 256 decode steps",
                "prefill_tokens_per_s": 8 * 4096 / best, "prefill_s": best, "infer_calls": calls,
                "prefill_TFLOPs": 8 * 4096 * flops_tok / best / 1e12, "prefill_frac_of_mfma_peak": 8 * 4096 * flops_tok / best / 2.5e15,
                "decode_tokens_per_s": 8 * 256 / (dms * 1e-3), "decode_ms_per_step": dms / 256,
                "decode_frac_of_hbm_peak": ab7["per_step"] / (dms / 256 * 1e-3) / HBM_PEAK}
        e7.close()
        cfg5["prefill_tokens_per_s_at_token_chunk_size_1024"] = 8 * 4096 / prefill7(e7b)
        e7b.close()

    if rank == 0:
        line = {"metric": "decode tokens/sec (whole job)", "value": value, "unit": "tokens/s", "n_gpus": world,
                "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_per_step, "higher_is_better": True,
                "scaling": "weak", "vs_baseline": None, "dtype": {"none": "f16", "int8": "u8->f16", "nf4": "nf4->f16"}[args.quant] + "/f32acc",
                "data": "synthetic",
                "config": {"workload": f"RWKV-{args.workload} {args.quant} decode, batch={B}/GPU, greedy, state+tokens resident in HBM",
                           "quant": args.quant, "batch_per_gpu": B, "precision": args.precision,
                           "parallelism": f"replicas x{world} (no collective)"},
                "tokens_per_s_per_gpu": value / world, "device_ms_per_step": dev_ms / args.steps,
                "roofline": roof, "cpu_baseline": cpu, "embeddings": emb, "pcie_inclusive_tokens_per_s": pcie, "on_device_sampling_tokens_per_s": sampled, "sweep": sweep or None, "tokens_verified": tokens_verified, "config5": cfg5,
                "load_s": t_load, "synth_s": t_synth}
        print(json.dumps(line), flush=True)
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
