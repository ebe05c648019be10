#!/usr/bin/env python3
"""Prefetch depth of the 128-family gemm_nt kernels (double buffer vs 3/4-stage ring) on the shapes that use them in the step."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from stswincl_amd import hip
from bench_gemm import timeit


def main():
    dev, dt = "cuda", torch.bfloat16
    shapes = [  # M, N, K, S, dilation, note
        (4096, 512, 1024, 9, 2, "aspp 3x3 (128x64)"), (4096, 1024, 512, 9, 1, "decoder 3x3 (128x64)"), (4096, 512, 1024, 1, 0, "aspp 1x1"),
        (65536, 128, 128, 9, 1, "layer2 3x3 (128x128)"), (65536, 128, 64, 9, 1, "layer2 first 3x3"), (65536, 128, 256, 9, 1, "layer4 dgrad"),
        (262144, 64, 64, 9, 1, "layer1 3x3 (256x64)"), (262144, 64, 128, 9, 1, "layer2 dgrad (256x64)"),
        (16384, 256, 448, 9, 1, "decoder 3x3"), (16384, 448, 256, 9, 1, "decoder dgrad"), (16384, 64, 512, 1, 0, "low-level 1x1"),
        (65536, 384, 512, 1, 0, "plain N=384"), (16384, 12, 256, 1, 0, "classifier"),
    ]
    print(f"{'M':>7s} {'N':>5s} {'K':>5s} {'S':>2s} | {'auto':>8s} {'nodeep':>8s} {'deep':>8s}  variant(auto)  note")
    for M, N, K, S, dil, note in shapes:
        A = torch.randn(M, K, device=dev).to(dt)
        W = (torch.randn(N, S * K, device=dev) / (S * K) ** 0.5).to(dt)
        out = torch.empty(M, N, device=dev, dtype=dt)
        rows = None
        if S > 1:
            f = 16
            side = int((M // f) ** 0.5)
            rows = hip.conv3x3_rowmap(f, side, side, dil)
        cells = []
        for fl in (0, hip.GF_NODEEP, hip.GF_DEEP):
            t = timeit(lambda: hip.gemm_nt(A, W, out, M=M, a_rows=rows, S=S, flags=fl)) * 1e3
            if fl == 0:
                var = hip.last_variant(0)["kernel"]
            cells.append(f"{t:8.1f}")
        print(f"{M:7d} {N:5d} {K:5d} {S:2d} | " + " ".join(cells) + f"  {str(var):>14s}  {note}", flush=True)


if __name__ == "__main__":
    main()
