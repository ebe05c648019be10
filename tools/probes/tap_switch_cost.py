"""3x3 convolution as tap-segmented gather GEMM (S = 9, real tap maps) against the same flops as ONE plain GEMM (S = 1,
K = 9 * Kseg): what do the tap switches of the ring kernel's stage loop cost?"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
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
for M, N, K in [(65536, 512, 512), (65536, 256, 256), (65536, 512, 256), (16384, 256, 448)]:
    side = int((M // 16) ** 0.5)
    A = torch.randn(M, K, device="cuda").bfloat16()
    W = (torch.randn(N, 9 * K, device="cuda") / (9 * K) ** 0.5).bfloat16()
    out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    rows = hip.conv3x3_rowmap(16, side, side, 1)
    t9 = timeit(lambda: hip.gemm_nt(A, W, out, M=M, a_rows=rows, S=9))
    A1 = torch.randn(M, 9 * K, device="cuda").bfloat16()
    t1 = timeit(lambda: hip.gemm_nt(A1, W, out, M=M))
    fl = 2.0 * M * N * 9 * K
    print(f"M={M} N={N} Kseg={K}: conv S=9 {t9:7.1f} us ({fl / t9 / 1e6:6.0f} TF/s)   plain K={9 * K} {t1:7.1f} us ({fl / t1 / 1e6:6.0f} TF/s)")
