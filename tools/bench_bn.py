#!/usr/bin/env python3
"""BatchNorm kernels on the shapes of the training step: us and effective GB/s (algorithmic bytes) per kernel."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from stswincl_amd import hip
from bench_gemm import timeit


def main():
    dev, dt = "cuda", torch.bfloat16
    shapes = [(1048576, 64, 0, "stem bn1"), (262144, 64, 0, "layer1 bn1"), (262144, 64, 1, "layer1 bn2 (+resid)"),
              (65536, 128, 1, "layer2 bn2"), (65536, 256, 0, "layer4 bn1"), (65536, 256, 1, "layer4 bn2"),
              (65536, 512, 0, "layer5 bn1"), (65536, 512, 1, "layer5 bn2"), (4096, 512, 0, "aspp"), (16384, 256, 0, "decoder")]
    G, unit_of = 4, lambda M: M // 16
    print(f"{'M':>8s} {'C':>4s} res | {'colstats':>14s} {'apply':>14s} {'bwd reduce':>14s} {'bwd dx':>14s}   note   (us, GB/s)")
    cold = os.environ.get("STSWIN_BN_COLD", "1") == "1"     # rotate over > 1 GB of operands: the 256 MB infinity cache must not serve them
    print(f"# operands {'rotated over > 1 GB (cold)' if cold else 'reused (warm: served by the infinity cache up to 256 MB)'}")
    for M, C, res, note in shapes:
        nb = M * C * 2
        nset = max(2, int(1.5e9 // (4 * nb))) if cold else 1
        nset = min(nset, 24)
        xs = [torch.randn(M, C, device=dev).to(dt) for _ in range(nset)]
        gs_ = [torch.randn(M, C, device=dev).to(dt) for _ in range(nset)]
        rs_ = [torch.randn(M, C, device=dev).to(dt) if res else None for _ in range(nset)]
        ys = [torch.empty(M, C, device=dev, dtype=dt) for _ in range(nset)]
        dxs = [torch.empty(M, C, device=dev, dtype=dt) for _ in range(nset)]
        drs = [torch.empty(M, C, device=dev, dtype=dt) if res else None for _ in range(nset)]
        x = xs[0]
        gamma, beta = torch.rand(C, device=dev) + 0.5, torch.randn(C, device=dev)
        unit = unit_of(M)
        s, ss = hip.colstats(x, groups=G, unit=unit)
        mean, rstd = hip.bn_finalize(x, s, ss, None, None, G, 1e-5, 0.1, unit=unit)
        for i in range(nset):
            hip.bn_apply(xs[i], mean, rstd, gamma, beta, ys[i], resid=rs_[i], groups=G, relu=True, unit=unit)
        it = [0]

        def nxt():
            it[0] = (it[0] + 1) % nset
            return it[0]
        t_cs = timeit(lambda: hip.colstats(xs[nxt()], groups=G, unit=unit), iters=40) * 1e3

        def ap():
            i = nxt()
            hip.bn_apply(xs[i], mean, rstd, gamma, beta, dxs[i], resid=rs_[i], groups=G, relu=True, unit=unit)
        t_ap = timeit(ap, iters=40) * 1e3
        sums = hip.zeros(2, G, C, device=dev)
        gs = hip.zeros(2, C, device=dev)

        def red():
            i = nxt()
            hip.bn_bwd(gs_[i], xs[i], ys[i] if res else None, mean, rstd, gamma, dxs[i], dresid=drs[i], groups=G, relu=True, phase=1,
                       sums=(sums[0], sums[1]), beta=beta, unit=unit)

        def dxk():
            i = nxt()
            hip.bn_bwd(gs_[i], xs[i], ys[i] if res else None, mean, rstd, gamma, dxs[i], dresid=drs[i], groups=G, relu=True, phase=2,
                       sums=(sums[0], sums[1]), beta=beta, unit=unit, group_sums=gs)
        t_r = timeit(red, iters=40) * 1e3
        t_d = timeit(dxk, iters=40) * 1e3
        by = {"cs": nb, "ap": nb * (2 + res), "r": nb * (2 + res), "d": nb * (3 + 2 * res)}
        cell = lambda t, b: f"{t:7.1f} {b / t / 1e3:6.0f}"                                   # noqa: E731
        print(f"{M:8d} {C:4d} {res:3d} | {cell(t_cs, by['cs'])} {cell(t_ap, by['ap'])} {cell(t_r, by['r'])} {cell(t_d, by['d'])}   {note}",
              flush=True)


if __name__ == "__main__":
    main()
