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
Ni, Nj = 2048, 512
for Mk in (512, 2048, 4096, 8192, 16384, 32768, 65536):
    At = torch.randn(Mk, Ni, device="cuda").bfloat16()
    Bt = torch.randn(Mk, Nj, device="cuda").bfloat16()
    out = torch.zeros(Ni, Nj, device="cuda")
    t = timeit(lambda: hip.gemm_tn(At, Bt, out, Mk=Mk, splits=8))
    kt = Mk // 64 // 8
    print(f"Mk={Mk:6d} ({kt:3d} k-tiles/block, 512 blocks) {t:8.1f} us  {2.0*Mk*Ni*Nj/t/1e6:7.1f} TF/s")
