import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from stswincl_amd import hip
def timeit(fn, iters=30):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / iters * 1e3
M = 32768
for N in (512, 2048):
    for K in (64, 128, 256, 512, 1024, 2048, 4096):
        A = torch.randn(M, K, device="cuda").bfloat16()
        W = (torch.randn(N, K, device="cuda") / K ** 0.5).bfloat16()
        out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
        t = timeit(lambda: hip.gemm_nt(A, W, out, M=M, flags=hip.GF_NOBIG))
        tiles = (M // 128) * (N // 128)
        print(f"N={N:5d} K={K:5d} {t:8.1f} us  {2.0*M*N*K/t/1e6:7.1f} TF/s   per-tile-slot {t/ (tiles/512):6.2f} us  ({K//64} k-tiles)")
