#!/usr/bin/env python3
"""What does the row gather of a weight-gradient GEMM cost, and is it the map handling or the access pattern?  gemm_tn on the qkv / proj
weight-gradient shapes of stage 1: no map, an IDENTITY map (map handling only), the window map of the model (runs of 8 consecutive
tokens), a random permutation."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from stswincl_amd import hip


def timeit(fn, iters=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters * 1e3


dev, dt = "cuda", torch.bfloat16
Mk = 65536
ident = torch.arange(Mk, dtype=torch.int32, device=dev)
win = hip.win_rowmap(4, 2, 64, 64, 8, 0) if False else hip.win_rowmap(8, 2, 64, 64, 8, 4)     # 8 pairs x 2 frames x 64 x 64 tokens, shifted windows
assert win.numel() == Mk
perm = torch.randperm(Mk, device=dev).to(torch.int32)
print(f"{'shape':34s} {'no map':>8s} {'identity':>9s} {'window':>8s} {'random':>8s}   (us; gathered operand in brackets)")
for name, Ni, Nj, side in (("qkv wgrad  dqkv^T x X[map]", 1536, 512, "b"), ("proj wgrad dx1[map]^T x o", 512, 512, "a"), ("fc1 wgrad (plain in the model)", 2048, 512, "b")):
    At = torch.randn(Mk, Ni, device=dev).to(dt)
    Bt = torch.randn(Mk, Nj, device=dev).to(dt)
    out = torch.empty(Ni, Nj, device=dev)
    cells = []
    for m in (None, ident, win, perm):
        kw = {} if m is None else ({"bt_rows": m} if side == "b" else {"at_rows": m})
        cells.append(timeit(lambda: hip.gemm_tn(At, Bt, out, Mk=Mk, overwrite=True, **kw)))
    fl = 2.0 * Mk * Ni * Nj
    print(f"{name:34s} " + " ".join(f"{c:8.1f}" for c in cells) + "   TF/s: " + " ".join(f"{fl / c / 1e6:6.0f}" for c in cells) + f"   [{side}]", flush=True)
