"""gemm_nt on the narrow, short shapes of the decode head (N <= 64, M = 4096 .. 16384): us per launch.
    python tools/bench_narrow.py          (STSWIN_HIP_LIB=<older libstswin_hip.so> for the other side of an A/B)"""
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


for M, N, K in [(4, 512, 1024), (4, 1024, 512), (16384, 64, 512), (4096, 64, 1024), (16384, 64, 256), (16384, 48, 256), (32768, 64, 512), (65536, 64, 512), (8192, 64, 2048)]:
    a = torch.randn(M, K, device="cuda").to(torch.bfloat16)
    b = torch.randn(N, K, device="cuda").to(torch.bfloat16)
    c = torch.empty(M, N, dtype=torch.bfloat16, device="cuda")
    us = timeit(lambda: hip.gemm_nt(a, b, c, M=M))
    ref = a.float() @ b.float().t()
    err = float((c.float() - ref).abs().max() / ref.abs().max())
    print(f"M={M:6d} N={N:3d} K={K:5d}  {us:7.1f} us  {2.0 * M * N * K / us * 1e-6:7.1f} TFLOP/s  variant {hip.last_variant(0)['kernel']}  err {err:.1e}")


print("few tiles, long K (tap-segmented gather): split-K ring vs the tiled kernels")
for M, N, Kseg, S in [(4096, 512, 1024, 9), (4096, 1024, 512, 9), (16384, 256, 448, 9)]:
    a = torch.randn(M, Kseg, device="cuda").to(torch.bfloat16)
    b = (torch.randn(N, S * Kseg, device="cuda") / (S * Kseg) ** 0.5).to(torch.bfloat16)
    rmap = torch.randint(-1, M, (S, M), device="cuda", dtype=torch.int32)
    c = torch.empty(M, N, dtype=torch.bfloat16, device="cuda")
    for on in (True, False):
        hip._NT_SPLITK = on
        us = timeit(lambda: hip.gemm_nt(a, b, c, M=M, a_rows=rmap, S=S), n=30)
        print(f"M={M:6d} N={N:4d} K={Kseg:5d} x {S}  split-K {'on ' if on else 'off'} {us:7.1f} us  {2.0 * M * N * Kseg * S / us * 1e-6:7.1f} TFLOP/s  {hip.last_variant(0)}")
    hip._NT_SPLITK = True
