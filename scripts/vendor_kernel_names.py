"""Round 6: which hipBLASLt / rocBLAS kernels (macro tile in the name) torch.matmul picks for the prefill shapes — run under `rocprofv3 --kernel-trace --stats`."""
import torch
dev = torch.device("cuda:0")
for rows, K in [(10304, 2560), (11520, 2560), (2560, 8960), (2560, 2560), (16448, 4096), (4096, 14336)]:
    for T in (256, 2048, 4096):
        W = torch.randn(rows, K, device=dev, dtype=torch.float16); X = torch.randn(T, K, device=dev, dtype=torch.float16)
        for _ in range(3): torch.matmul(X, W.t())
        torch.cuda.synchronize()
