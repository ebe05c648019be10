#!/usr/bin/env python3
"""Main loop of the 4-wave ring kernel with one instruction stream removed at a time (libraries of tools/probes/w4_variants.sh, chosen by
STSWIN_HIP_LIB): time of a K = 4096 GEMM without an epilogue (debug flag bit 21), per 32-deep stage."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from stswincl_amd import hip
dev, dt = "cuda", torch.bfloat16


def timeit(fn, iters=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters * 1e3


for M, N, K in ((4096, 4096, 4096), (4096, 4096, 8192)):
    A = (torch.randn(M, K, device=dev) * 0.5).to(dt)
    W = (torch.randn(N, K, device=dev) / K ** 0.5).to(dt)
    out = torch.zeros(M, N, device=dev, dtype=dt)
    res = []
    for fl in (hip.GF_BIG, hip.GF_BIG | hip.GF_W4R):
        for noepi in (0, 1 << 21):
            res.append(timeit(lambda: hip.gemm_nt(A, W, out, M=M, flags=fl | noepi)))
    print(f"{os.path.basename(os.environ.get('STSWIN_HIP_LIB', 'product')):36s} K={K}: 8-wave {res[0]:7.1f} us (no epilogue {res[1]:7.1f})   4-wave {res[2]:7.1f} us "
          f"(no epilogue {res[3]:7.1f} = {res[3] / (K / 32) * 1e3:6.1f} ns per stage; MFMA-bound: 427)", flush=True)
