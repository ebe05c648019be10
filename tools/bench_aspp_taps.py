#!/usr/bin/env python3
"""Dilated 3x3 convolutions of ASPP (ASPP.py:13-20: 1024 -> 512 channels, dilation 6 / 12 / 18) on the 32x32 stage-2 map of a
512x512 frame, B = 4: the gather GEMM with and without tap skipping (STSWIN_GF_TAPSKIP), forward and input-gradient form."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from stswincl_amd import hip


def timeit(fn, iters=30):
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
F_, H, W = 4, 32, 32
M = F_ * H * W
for cin, cout, tag in ((1024, 512, "forward 1024->512"), (512, 1024, "input gradient 512->1024")):
    x = torch.randn(M, cin, device=dev).to(dt)
    wmat = (torch.randn(cout, 9 * cin, device=dev) / (9 * cin) ** 0.5).to(dt)
    y = torch.empty(M, cout, dtype=dt, device=dev)
    for d in (6, 12, 18):
        rmap = hip.conv_rowmap(F_, H, W, H, W, 3, 1, d, d, False, dev)
        valid = float((rmap >= 0).float().mean())
        t_on = timeit(lambda: hip.gemm_nt(x, wmat, y, M=M, a_rows=rmap, S=9, flags=hip.GF_TAPSKIP))
        y_on = y.clone()
        hip._NT_SPLITK = False
        t_off = timeit(lambda: hip.gemm_nt(x, wmat, y, M=M, a_rows=rmap, S=9))
        hip._NT_SPLITK = True
        same = torch.equal(y_on, y)
        if hip.load().stswin_gemm_nt_splitk_scratch(M, cout, cin, 9) > 0:
            t_sk = timeit(lambda: hip.gemm_nt(x, wmat, y, M=M, a_rows=rmap, S=9))
            print(f"{tag:28s} dilation {d:2d}: split-K ring (stswin_gemm_nt_splitk, {hip.last_variant(0)['splits']} splits) {t_sk:6.1f} us; "
                  f"max |difference| {float((y.float() - y_on.float()).abs().max()):.3g}")
        fl = 2.0 * M * cout * 9 * cin
        print(f"{tag:28s} dilation {d:2d}: {t_off:6.1f} us all taps -> {t_on:6.1f} us with tap skipping ({100 * valid:.0f} % of the tap rows are "
              f"inside the image); {fl / t_on / 1e6:6.0f} TF/s dense-equivalent; bitwise equal: {same}")
