"""Round 6: what does the vendor library (hipBLASLt / rocBLAS through torch.matmul) reach on the prefill GEMM shapes of the engines the bench
quotes?  A calibration of the practical ceiling for the hand-written tile kernels, NOT a product path (the product never imports torch).

Every shape is out[T, rows] = X[T, K] @ W[rows, K]^T in f16 with fp32 accumulation, weights rotating over enough copies that no launch finds its
matrix in the Infinity Cache (a model step touches every layer's weights once), timed as CUDA-graph-free back-to-back launches with events.

    python scripts/vendor_gemm_calibration.py            # prints one line per (shape, T)
"""
import sys, time
import torch

assert torch.cuda.is_available(), "needs the GPU"
dev = torch.device("cuda:0")
torch.backends.cuda.matmul.allow_fp16_reduced_precision_reduction = False

SHAPES = [  # name, rows, K  (V6-3B: C = 2560, ffn 8960; V6-7B: C = 4096, ffn 14336)
    ("3b r/k/v/g+D1", 10304, 2560), ("3b Fk+Fr", 11520, 2560), ("3b Fv", 2560, 8960), ("3b Wo", 2560, 2560),
    ("7b r/k/v/g+D1", 16448, 4096), ("7b Fk+Fr", 18432, 4096), ("7b Fv", 4096, 14336), ("7b Wo", 4096, 4096),
]
TS = [256, 1024, 2048, 4096]
PEAK = 2.5e15

def bench(rows, K, T, iters=40):
    copies = max(2, int(600e6 // (rows * K * 2)) + 1)              # > 256 MiB of distinct weights between two uses of one copy
    W = [torch.randn(rows, K, device=dev, dtype=torch.float16) * 0.02 for _ in range(copies)]
    X = torch.randn(T, K, device=dev, dtype=torch.float16)
    out = torch.empty(T, rows, device=dev, dtype=torch.float16)
    for i in range(6): torch.matmul(X, W[i % copies].t(), out=out)
    torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    ev[0].record()
    for i in range(iters): torch.matmul(X, W[i % copies].t(), out=out)
    ev[1].record(); torch.cuda.synchronize()
    us = ev[0].elapsed_time(ev[1]) * 1e3 / iters
    del W
    return us

print(f"# torch {torch.__version__}, {torch.cuda.get_device_name(0)}; f16 x f16 -> f16, fp32 accumulate; us per launch (TFLOP/s, fraction of 2.5 PFLOP/s)")
for name, rows, K in SHAPES:
    for T in TS:
        us = bench(rows, K, T)
        tf = 2.0 * rows * K * T / us / 1e6
        print(f"{name:14s} rows={rows:6d} K={K:6d} T={T:5d}: {us:8.1f} us  {tf:7.1f} TF  {tf * 1e12 / PEAK:5.3f}", flush=True)
