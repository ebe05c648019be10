#!/usr/bin/env python3
"""Which vendor (hipBLASLt / rocBLAS) kernels torch.matmul picks for the bf16 NT shapes of the step (run under rocprofv3 --kernel-trace --stats)."""
import torch
dev, dt = "cuda", torch.bfloat16
for M, N, K in ((65536, 512, 2048), (16384, 1024, 4096), (65536, 512, 4608), (4096, 4096, 4096), (65536, 2048, 512)):
    a = torch.randn(M, K, device=dev).to(dt)
    w = torch.randn(N, K, device=dev).to(dt)
    for _ in range(5):
        c = a @ w.t()
    torch.cuda.synchronize()
