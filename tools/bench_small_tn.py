"""gemm_tn (+ its split-K combine) on the small-output weight gradients of the decode head / ResNet shortcuts: us per call."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from stswincl_amd import hip  # noqa: E402


def timeit(fn, n=100):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


for Mk, Ni, Nj in [(16384, 64, 512), (4096, 64, 1024), (16384, 64, 256), (4096, 256, 512), (4096, 512, 1024), (65536, 128, 64), (65536, 256, 128),
                   (16384, 512, 64), (65536, 512, 512)]:
    a = torch.randn(Mk, Ni, device="cuda").to(torch.bfloat16)
    b = torch.randn(Mk, Nj, device="cuda").to(torch.bfloat16)
    c = torch.empty(Ni, Nj, dtype=torch.float32, device="cuda")
    us = timeit(lambda: (hip.gemm_tn(a, b, c, Mk=Mk, overwrite=True), hip.tn_join()))
    v = hip.last_variant(1)
    print(f"Mk={Mk:6d} Ni={Ni:4d} Nj={Nj:5d}  {us:7.1f} us  {2.0 * Mk * Ni * Nj / us * 1e-6:7.1f} TFLOP/s  {v}")
