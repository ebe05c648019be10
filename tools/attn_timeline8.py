#!/usr/bin/env python3
"""In-kernel timeline of the 8-wave stage-1 attention backward (debug flag): per-workgroup timestamps (100 MHz) of the last
STEADY-STATE problem each persistent workgroup processed (the one before its final problem, which prefetches nothing), for wave 0 (dV + first dQ half) and wave 4 (second dQ half + dK)."""
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
    ts = torch.zeros(256 * 32, dtype=torch.int64, device=dev)
    hip.win_attn_bwd(qkv, do, biasT, None, dbiasT, nB_=nB_, nW=nW, T=T, ws=ws, heads=heads, C=C, scale=0.1,
                     colsum_out=ts.view(torch.float32), debug_ts=True)
torch.cuda.synchronize()
qpf = os.environ.get("STSWIN_ATTN_BWD_QPF", "1") != "0"
if qpf:      # round-4 schedule (K, Q, V prefetched; dV | dK then the two dQ halves)
    names = ["wait for the prefetched K / Q / V + sync", "dO request, Q pieces from LDS, scores, softmax", "dO wait + sync (stats exchange)",
             "merge, dO pieces, dP^T, row sums", "sync (exchange 2), dS, dbias acc", "sync, P/dS -> LDS, sync",
             "product 1 (dV | dK)", "sync", "stores 1 (dV | dK) + next K, Q, V requests", "product 2 (dQ half) + stores + next table values",
             "end sync"]
else:        # round-3 schedule (STSWIN_ATTN_BWD_QPF=0)
    names = ["Q row pieces + K/V wait + sync", "dO request, table loads, scores, softmax", "dO wait + sync (stats exchange)",
             "merge, dO pieces, dP^T, row sums", "sync (exchange 2), dS, dbias acc", "sync, P/dS -> LDS, sync",
             "Q request, product 1 (dV | dQ half)", "Q wait + sync", "stores 1 + next K request", "product 2 (dQ half | dK) + stores",
             "end sync"]
for hw, who in ((0, "wave 0 (dV, dQ cols 0-63)"), (1, "wave 4 (dK, dQ cols 64-127)" if qpf else "wave 4 (dQ cols 64-127, dK)")):
    t = ts.view(256, 2, 16)[:, hw, :12].cpu().double() / 100.0
    d = t[:, 1:] - t[:, :-1]
    print(f"{who}: phase durations of one problem (us): mean / p10 / p90 over 256 workgroups")
    for i, n in enumerate(names):
        c = d[:, i]
        print(f"  {n:44s} {float(c.mean()):6.2f} {float(c.quantile(0.1)):6.2f} {float(c.quantile(0.9)):6.2f}")
    print(f"  total per problem {float((t[:, 11] - t[:, 0]).mean()):6.2f}")
    full = ts.view(256, 2, 16)[:, hw, :].cpu().double() / 100.0
    life = full[:, 14] - full[:, 12]
    print(f"  workgroup lifetime (start -> end of its last problem): mean {float(life.mean()):6.1f} us, max {float(life.max()):6.1f}; "
          f"first start -> last end over the launch {float(full[:, 14].max() - full[:, 12].min()):6.1f} us; start spread {float(full[:, 12].max() - full[:, 12].min()):5.1f} us")
