#!/usr/bin/env python3
"""In-kernel timeline of the stage-1 attention backward (debug flag): per-workgroup timestamps (100 MHz) of the LAST
problem each persistent workgroup processed."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from stswincl_amd import hip
dev, dt = "cuda", torch.bfloat16
rows, C, heads, ws, T = 65536, 512, 4, 8, 2
N = ws * ws; ntok = T * N; nB_ = rows // ntok; nW = 64
qkv = (torch.randn(rows, 3 * C, device=dev) * 0.5).to(dt)
do = torch.randn(rows, C, device=dev).to(dt)
biasT = torch.randn(heads, N, N, device=dev)
dbiasT = torch.zeros(heads, N, N, device=dev)
for _ in range(3):
    ts = torch.zeros(256 * 16, dtype=torch.int64, device=dev)
    hip.win_attn_bwd(qkv, do, biasT, None, dbiasT, nB_=nB_, nW=nW, T=T, ws=ws, heads=heads, C=C, scale=0.1,
                     colsum_out=ts.view(torch.float32), debug_ts=True)
torch.cuda.synchronize()
t = ts.view(256, 16).cpu().double()[:, :11] / 100.0
d = t[:, 1:] - t[:, :-1]
names = ["K/V wait + sync (+dO request)", "scores + softmax", "dP^T", "dS, dbias acc, P/dS -> LDS", "sync (+dO wait, Q request)",
         "dV multiply", "Q wait + sync + dV stores + next K request", "dK multiply + stores", "dQ multiply + stores", "end sync"]
print("phase durations of one problem (us): mean / p10 / p90 over 256 workgroups")
for i, n in enumerate(names):
    c = d[:, i]
    print(f"  {n:44s} {float(c.mean()):6.2f} {float(c.quantile(0.1)):6.2f} {float(c.quantile(0.9)):6.2f}")
print(f"  total per problem {float((t[:, 10] - t[:, 0]).mean()):6.2f}")
