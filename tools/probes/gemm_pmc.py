#!/usr/bin/env python3
"""One compute-bound gemm_nt (4096^3, bf16, bias) launched 12 times: the target of rocprofv3 --pmc passes (tools/probes/gemm_pmc.sh)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from stswincl_amd import hip
M = N = K = 4096
A = torch.randn(M, K, device="cuda").bfloat16()
W = (torch.randn(N, K, device="cuda") / K ** 0.5).bfloat16()
bias = torch.randn(N, device="cuda")
out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
for _ in range(12):
    hip.gemm_nt(A, W, out, M=M, bias=bias)
torch.cuda.synchronize()
