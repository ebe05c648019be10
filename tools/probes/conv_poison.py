#!/usr/bin/env python3
"""Do the convolution kernels read or leave uninitialised memory?  Every free block of the caching allocator is poisoned with NaN bit patterns
before a conv + BN forward / backward on ASPP's dilated shapes; outputs and gradients are compared with torch's own fp32 convolution."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import torch.nn.functional as F
from stswincl_amd import headops as H, hip


def poison(gb=6):
    bufs = [torch.full((1 << 28,), float("nan"), device="cuda") for _ in range(gb)]      # 1 GiB each
    small = [torch.full((n,), float("nan"), device="cuda") for n in (1 << 10, 1 << 14, 1 << 18, 1 << 20, 1 << 22, 1 << 24) for _ in range(8)]
    del bufs, small                                                                       # back to the allocator's free lists, contents intact


def rel(a, b):
    return float((a.double().cpu() - b.double().cpu()).norm() / (b.double().cpu().norm() + 1e-30))


torch.manual_seed(0)
for (b, hh, cin, cout, dil) in [(4, 16, 1024, 512, 6), (4, 16, 1024, 512, 12), (4, 16, 1024, 512, 18), (4, 32, 1024, 512, 6), (4, 32, 1024, 512, 12),
                                (4, 32, 1024, 512, 18), (2, 8, 1024, 512, 18)]:
    conv = torch.nn.Conv2d(cin, cout, 3, padding=dil, dilation=dil).cuda()
    x = torch.randn(b, cin, hh, hh, device="cuda")
    g = torch.randn(b, cout, hh, hh, device="cuda")
    xr = x.clone().requires_grad_(True)
    yr = F.conv2d(xr, conv.weight, conv.bias, padding=dil, dilation=dil)
    (yr * g).sum().backward()
    dw_ref, dx_ref, db_ref = conv.weight.grad.clone(), xr.grad.clone(), conv.bias.grad.clone()
    for trial in range(3):
        conv.weight.grad = conv.bias.grad = None
        with torch.autocast("cuda", dtype=torch.bfloat16):
            xt = H.to_tokens(x).contiguous().requires_grad_(True)
            poison()
            y = H.conv_tokens(xt, conv, b, hh, hh)[0]
            poison()
            (y.float() * H.to_tokens(g)).sum().backward()
        torch.cuda.synchronize()
        dw, dx = conv.weight.grad, xt.grad
        print(f"B={b} {hh}x{hh} {cin}->{cout} dil {dil} trial {trial}: y {rel(H.from_tokens(y.float(), b, hh, hh), yr):.3e} dw {rel(dw, dw_ref):.3e} "
              f"dx {rel(H.from_tokens(dx.float(), b, hh, hh), dx_ref):.3e} db {rel(conv.bias.grad, db_ref):.3e} finite {bool(torch.isfinite(dw).all())} "
              f"{bool(torch.isfinite(dx).all())} {bool(torch.isfinite(y).all())}", flush=True)
