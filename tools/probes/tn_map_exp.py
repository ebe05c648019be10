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
Mk, Ni, Nj = 65536, 1536, 512
At = torch.randn(Mk, Ni, device="cuda").bfloat16(); Bt = torch.randn(Mk, Nj, device="cuda").bfloat16()
out = torch.zeros(Ni, Nj, device="cuda")
ident = torch.arange(Mk, device="cuda", dtype=torch.int32)
win = hip.win_rowmap(4, 4, 64, 64, 8, 4)
perm = torch.randperm(Mk, device="cuda").to(torch.int32)
for name, m in (("no map", None), ("identity map", ident), ("window map", win), ("random permutation", perm)):
    print(f"{name:20s} {timeit(lambda: hip.gemm_tn(At, Bt, out, Mk=Mk, bt_rows=m)):7.1f} us")

# experiments on the identity map: indices loaded but ignored (-2) / not loaded at all (-3); results are not written
# (negative ldc) - timing only
for name, d in (("identity, loaded+ignored", 2), ("identity, not loaded", 3)):
    print(f"{name:26s} {timeit(lambda: hip.gemm_tn(At, Bt, out, Mk=Mk, bt_rows=ident, atomics=True, debug_ts=d)):7.1f} us")
print(f"{'identity (atomics path)':26s} {timeit(lambda: hip.gemm_tn(At, Bt, out, Mk=Mk, bt_rows=ident, atomics=True)):7.1f} us")
print(f"{'no map (atomics path)':26s} {timeit(lambda: hip.gemm_tn(At, Bt, out, Mk=Mk, atomics=True)):7.1f} us")
