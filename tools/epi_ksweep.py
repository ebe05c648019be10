import os, sys
sys.path.insert(0, "/root/repo")
import torch
from stswincl_amd import hip
def timeit(fn, iters=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / iters * 1e3
dt, dev = torch.bfloat16, "cuda"
for M, N in ((65536, 2048), (65536, 512), (65536, 256)):
    for K in (64, 128, 256, 512, 1024, 2048):
        A = torch.randn(M, K, device=dev).to(dt)
        W = torch.randn(N, K, device=dev).to(dt)
        out = torch.empty(M, N, device=dev, dtype=dt)
        t_big = timeit(lambda: hip.gemm_nt(A, W, out, M=M, flags=hip.GF_BIG))
        t_8w = timeit(lambda: hip.gemm_nt(A, W, out, M=M, flags=hip.GF_BIG | (1 << 20)))
        t_ne = timeit(lambda: hip.gemm_nt(A, W, out, M=M, flags=hip.GF_BIG | (1 << 21)))
        tiles = (M // 256) * (N // 256)
        print(f"M={M} N={N} K={K:5d} big {t_big:7.1f} us ({t_big / max(1, tiles / 256):6.1f} us/round)   big-nostore {t_8w:7.1f} us  big-noepilogue {t_ne:7.1f} us", flush=True)
