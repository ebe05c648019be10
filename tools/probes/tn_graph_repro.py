#!/usr/bin/env python3
"""Isolate the graph-vs-eager difference of the ASPP dilated-convolution weight gradients: a small hipGraph of N gemm_tn launches
(the failing shape: Mk = 1024 rows, Ni = 512, Nj = 9 x 1024 tap-segmented gathered B) sharing the split-K workspace like consecutive backward
calls do, replayed with fresh operands each time, against the same launches run eagerly."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from stswincl_amd import hip

dev = "cuda"
frames, Hh = 4, 16
Mk, Ni, Cin = frames * Hh * Hh, 512, 1024
dils = (18, 12, 6)
maps = [hip.conv_rowmap(frames, Hh, Hh, Hh, Hh, 3, 1, d, d, False, dev) for d in dils]
g = [torch.randn(Mk, Ni, device=dev).bfloat16() for _ in dils]
X = torch.randn(Mk, Cin, device=dev).bfloat16()
outs = [torch.empty(Ni, 9 * Cin, dtype=torch.float32, device=dev) for _ in dils]


def launches():
    for i in range(len(dils)):
        hip.gemm_tn(g[i], X, outs[i], Mk=Mk, bt_rows=maps[i], bseg=Cin, overwrite=True, tapminor=True)
        # something between them, like the backward has (an elementwise pass over another buffer)
        X2 = X.float().mul_(1.0).bfloat16()


launches()
torch.cuda.synchronize()
print("variant of the launch:", hex(hip.load().stswin_last_variant(1)))
side = torch.cuda.Stream()
with torch.cuda.stream(side):
    launches()
torch.cuda.synchronize()
hip.note_capture()
graph = torch.cuda.CUDAGraph()
with torch.cuda.graph(graph):
    launches()
bad = 0
for it in range(40):
    torch.manual_seed(it)
    for t in g:
        t.copy_(torch.randn(Mk, Ni, device=dev).bfloat16())
    X.copy_(torch.randn(Mk, Cin, device=dev).bfloat16())
    graph.replay()
    torch.cuda.synchronize()
    got = [o.clone() for o in outs]
    launches()
    torch.cuda.synchronize()
    for i, (a, b) in enumerate(zip(got, outs)):
        if not torch.equal(a, b):
            bad += 1
            n = int((a != b).sum())
            print(f"replay {it} launch {i} (dilation {dils[i]}): {n} of {a.numel()} elements differ, max {float((a - b).abs().max()):.4g} (scale {float(b.abs().max()):.3g})", flush=True)
print("mismatching (replay, launch) pairs:", bad, "of", 40 * len(dils))
