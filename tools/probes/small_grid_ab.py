import os, sys
sys.path.insert(0, "/root/repo")
import torch
from stswincl_amd import hip
def timeit(fn, iters=30):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / iters * 1e3
for M, N, K, S in [(4096, 512, 1024, 9), (4096, 1024, 512, 9), (4096, 256, 512, 1), (4096, 512, 2560, 1), (4096, 256, 256, 9), (8192, 512, 512, 1)]:
    A = torch.randn(M, K, device="cuda").bfloat16(); W = (torch.randn(N, S * K, device="cuda") / (S*K) ** 0.5).bfloat16()
    out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    rows = torch.randint(0, M, (S, M), device="cuda", dtype=torch.int32) if S > 1 else None
    t0 = timeit(lambda: hip.gemm_nt(A, W, out, M=M, a_rows=rows, S=S, flags=1 << 26))
    t1 = timeit(lambda: hip.gemm_nt(A, W, out, M=M, a_rows=rows, S=S))
    print(f"M={M} N={N} K={K} S={S}: 128x128 {t0:7.1f} us   auto {t1:7.1f} us")
