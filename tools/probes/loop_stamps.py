import os, sys
sys.path.insert(0, "/root/repo")
import torch
from stswincl_amd import hip
M, N, K = 65536, 2048, 512
A = torch.randn(M, K, device="cuda").bfloat16(); W = (torch.randn(N, K, device="cuda") / K ** 0.5).bfloat16()
out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
nblk = (M // 256) * (N // 256)
for extra in (0, 1 << 20, 1 << 21):   # full epilogue, epilogue without the global stores, no epilogue
    for _ in range(3):
        ts = torch.zeros(nblk * 16, dtype=torch.int64, device="cuda")
        hip.gemm_nt(A, W, out, M=M, flags=hip.GF_BIG | hip.GF_NOSTREAM | (1 << 19) | extra, colsum_out=ts.view(torch.float32))
    torch.cuda.synchronize()
    t = ts.view(nblk, 16).cpu().double() / 100.0
    # slots: 0 start, 1 kt=4, 2 kt=8, 7 kt=12, 3 loop end
    seg = [("start->kt4 (prologue + 4 stages)", 0, 1), ("kt4->kt8", 1, 2), ("kt8->kt12", 2, 7), ("kt12->end (4 stages)", 7, 3)]
    order = t[:, 0].argsort()
    first, later = order[:256], order[256:]
    print("flags", extra)
    for name, a, b in seg:
        d = t[:, b] - t[:, a]
        print(f"  {name:34s} all {float(d.mean()):6.2f}  first round {float(d[first].mean()):6.2f}  later rounds {float(d[later].mean()):6.2f}")
