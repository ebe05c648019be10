#!/usr/bin/env python3
"""Epilogue-heavy gemm_nt shapes of the Swin MLP (GELU + pre-activation copy, residual, GELU' product) timed per
kernel variant: 8w = 128x128 8-wave, stream = persistent 256x256 ping-pong ring, ring = one tile per workgroup (register
epilogue), ring-lds = same with the LDS-staged fp32 epilogue."""
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


def main():
    dt, dev = torch.bfloat16, "cuda"
    shapes = [(65536, 2048, 512, "fc1 s1"), (16384, 4096, 1024, "fc1 s2"), (65536, 512, 2048, "fc2 s1"), (65536, 512, 512, "proj s1"),
              (65536, 1536, 512, "qkv s1")]
    variants = (("auto", 0), ("8w", hip.GF_NOBIG), ("stream", hip.GF_BIG | hip.GF_STREAM), ("duo", hip.GF_BIG | hip.GF_DUO), ("ring", hip.GF_BIG | hip.GF_NOSTREAM), ("mid", hip.GF_MID))
    print(f"{'shape':10s} {'epilogue':10s} " + " ".join(f"{n:>8s}" for n, _ in variants) + "   (us)")
    for M, N, K, note in shapes:
        A = torch.randn(M, K, device=dev).to(dt)
        W = (torch.randn(N, K, device=dev) / K ** 0.5).to(dt)
        out = torch.empty(M, N, device=dev, dtype=dt)
        out2 = torch.empty(M, N, device=dev, dtype=dt)
        R = torch.randn(M, N, device=dev).to(dt)
        bias = torch.randn(N, device=dev)
        cs = torch.zeros(N, device=dev)
        cases = [("plain", dict()), ("bias", dict(bias=bias)), ("gelu+dgelu", dict(bias=bias, out2=out2, flags=hip.GF_GELU | hip.GF_C2_DGELU)),
                 ("resid", dict(bias=bias, resid=R, flags=hip.GF_RESID)),
                 ("mul_r+cs", dict(resid=R, flags=hip.GF_MUL_R, colsum_out=cs))]
        for cname, kw in cases:
            cells = []
            for vname, vf in variants:
                k2 = dict(kw)
                k2["flags"] = k2.get("flags", 0) | vf
                try:
                    cells.append(f"{timeit(lambda: hip.gemm_nt(A, W, out, M=M, **k2)):8.1f}")
                except Exception:
                    cells.append(f"{'err':>8s}")
            print(f"{note:10s} {cname:10s} " + " ".join(cells), flush=True)


if __name__ == "__main__":
    main()
