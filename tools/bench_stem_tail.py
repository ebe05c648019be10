"""BatchNorm + ReLU + MaxPool2d(3, 2, 1) on the stem output (16 x 256 x 256 x 64): fused one-pass kernels against the separate operators."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from stswincl_amd import hip  # noqa: E402


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


def main():
    f, h, w, C, G = 16, 256, 256, 64, 4
    M = f * h * w
    hp, wp = h // 2, w // 2
    x = torch.randn(M, C, device="cuda").to(torch.bfloat16)
    mean, rstd = torch.randn(G, C, device="cuda") * 0.1, torch.rand(G, C, device="cuda") + 0.5
    gamma, beta = torch.rand(C, device="cuda") + 0.5, torch.randn(C, device="cuda") * 0.1
    unit = h * w
    y = torch.empty_like(x)
    p = torch.empty(f * hp * wp, C, dtype=torch.bfloat16, device="cuda")
    arg = torch.empty(f * hp * wp, C, dtype=torch.uint8, device="cuda")
    dz = torch.empty_like(x)
    dx = torch.empty_like(x)
    dp = torch.randn_like(p)
    print(f"bn_apply                     {timeit(lambda: hip.bn_apply(x, mean, rstd, gamma, beta, y, groups=G, relu=True, unit=unit)):8.1f} us")
    print(f"maxpool fwd                  {timeit(lambda: hip.maxpool3x3s2(y, p, arg, f, h, w, hp, wp)):8.1f} us")
    print(f"fused bn_relu_pool           {timeit(lambda: hip.bn_relu_pool(x, mean, rstd, gamma, beta, f, h, w, groups=G, unit=unit)):8.1f} us")
    hip.maxpool3x3s2(y, p, arg, f, h, w, hp, wp)
    print(f"maxpool bwd                  {timeit(lambda: hip.maxpool3x3s2(dp, dz, arg, f, h, w, hp, wp, backward=True)):8.1f} us")
    print(f"bn_bwd (reduce + dx)         {timeit(lambda: hip.bn_bwd(dz, x, None, mean, rstd, gamma, dx, groups=G, relu=True, beta=beta, unit=unit)):8.1f} us")


if __name__ == "__main__":
    main()
