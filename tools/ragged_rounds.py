#!/usr/bin/env python3
"""gemm_nt shapes whose 256x256 tile count does not fill whole rounds of the 256 CUs (the half-batch middle-pair layers of the Swin stack:
128 or 384 tiles): the launcher sends them to the 128x128 kernel (round-1 measurement).  Re-measured: forced ring (GF_BIG) against default,
cold operands (rotated)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from stswincl_amd import hip

dev, dt = "cuda", torch.bfloat16
shapes = [(8192, 1024, 1024, "proj s2 mid"), (8192, 1024, 4096, "fc2 s2 mid"), (8192, 1024, 3072, "qkv dgrad s2 mid"), (8192, 3072, 1024, "qkv s2 mid"),
          (8192, 4096, 1024, "fc1 s2 mid (512 tiles)"), (32768, 512, 512, "proj s1 mid (256 tiles)"), (24576, 1024, 1024, "B = 6 (384 tiles)"),
          (4096, 1024, 4096, "64 tiles")]


def timeit(fn, nset, iters=40):
    for k in range(4):
        fn(k % nset)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for k in range(iters):
        fn(k % nset)
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters * 1e3


print(f"{'M':>6s} {'N':>5s} {'K':>5s} {'tiles':>5s} | {'default us':>10s} {'TF/s':>6s} {'variant':>8s} | {'ring us':>8s} {'TF/s':>6s} | note")
for M, N, K, note in shapes:
    nset = max(2, int(6e8 // (2 * (M * K + N * K + M * N))))
    A = [torch.randn(M, K, device=dev).to(dt) for _ in range(nset)]
    W = [(torch.randn(N, K, device=dev) / K ** 0.5).to(dt) for _ in range(nset)]
    O = [torch.empty(M, N, device=dev, dtype=dt) for _ in range(nset)]
    fl = 2.0 * M * N * K
    td = timeit(lambda i: hip.gemm_nt(A[i], W[i], O[i], M=M), nset)
    vd = hip.load().stswin_last_variant(0)
    tb = timeit(lambda i: hip.gemm_nt(A[i], W[i], O[i], M=M, flags=hip.GF_BIG), nset)
    tiles = ((M + 255) // 256) * ((N + 255) // 256)
    duo = ""
    if hip.tuning_build():          # the 128x256-tile, two-workgroups-per-CU variant of tuning builds (GF_DUO)
        tdu = timeit(lambda i: hip.gemm_nt(A[i], W[i], O[i], M=M, flags=hip.GF_DUO | hip.GF_BIG), nset)
        duo = f" | duo {tdu:7.1f} us {fl / tdu * 1e-6:6.0f} TF/s"
    print(f"{M:6d} {N:5d} {K:5d} {tiles:5d} | {td:10.1f} {fl / td * 1e-6:6.0f} {vd:8x} | {tb:8.1f} {fl / tb * 1e-6:6.0f}{duo} | {note}", flush=True)
