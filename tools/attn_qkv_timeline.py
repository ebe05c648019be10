#!/usr/bin/env python3
"""In-kernel timeline of the QKV-fused attention forward (s_memrealtime stamps, 100 MHz): per wave class, the mean time between
the phase boundaries of a workgroup's SECOND problem over all 256 workgroups."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from stswincl_amd import hip, ops

dev, dt = "cuda", torch.bfloat16
B, H, W, C, heads, ws, T = 8, 64, 64, 512, 4, 8, 2
N, d = ws * ws, C // heads
nW = (H // ws) * (W // ws)
nB_ = B * nW
M = B * T * H * W
x = torch.randn(M, C, device=dev).to(dt)
w = (torch.randn(3 * C, C, device=dev) / C ** 0.5).to(dt)
bq = torch.randn(3 * C, device=dev) * 0.1
biasT = torch.randn(4, heads, N, N, device=dev)
bidx = (torch.arange(nW, device=dev) % 4).to(torch.int32)
rmap = ops.window_rowmap(B, T, H, W, ws, 4, dev)
ts = torch.zeros(256, 8, 8, dtype=torch.int64, device=dev)
for _ in range(3):
    hip.win_attn_qkv_fwd(x, rmap, w, bq, biasT, nB_=nB_, nW=nW, T=T, ws=ws, heads=heads, C=C, scale=d ** -0.5, bias_index=bidx, debug_ts=ts)
torch.cuda.synchronize()
t = ts.cpu().double() / 100.0      # us
names = ["problem start -> first chunk landed", "projection main loop (16 chunks)", "last reads done + barrier", "bias/scale, tiles to LDS + barrier",
         "attention core (waves 0-3) / idle (4-7)", "end-of-problem barrier"]
for cls, sl in (("waves 0-3", slice(0, 4)), ("waves 4-7", slice(4, 8))):
    print(cls)
    for i, n in enumerate(names):
        dlt = (t[:, sl, i + 1] - t[:, sl, i])
        print(f"  {n:48s} {dlt.mean():7.2f} us  (min {dlt.min():6.2f}, max {dlt.max():6.2f})")
    tot = t[:, sl, 6] - t[:, sl, 0]
    print(f"  {'whole problem':48s} {tot.mean():7.2f} us")
