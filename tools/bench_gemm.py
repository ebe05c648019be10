#!/usr/bin/env python3
"""Per-shape timing of the GEMM kernels on the shapes the training step uses (random bf16 data, HIP events)."""
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
    return a.elapsed_time(b) / iters

def main():
    dt = torch.bfloat16
    dev = "cuda"
    print(f"{'kernel':8s} {'M':>7s} {'N':>6s} {'K':>6s} {'S':>2s} {'us':>9s} {'TFLOP/s':>8s}  note")
    extra = [(8192, 1024, 1024, 1, "proj s2"), (8192, 1024, 3072, 1, "dx s2 (qkv dgrad)"), (32768, 512, 1536, 1, "dx s1 (qkv dgrad)")]
    nt_shapes = extra + [(32768, 1536, 512, 1, "qkv s1 (gather)"), (32768, 512, 512, 1, "proj s1"), (32768, 2048, 512, 1, "fc1 s1"),
                 (32768, 512, 2048, 1, "fc2 s1"), (8192, 3072, 1024, 1, "qkv s2"), (8192, 4096, 1024, 1, "fc1 s2"),
                 (8192, 1024, 4096, 1, "fc2 s2"), (16384, 1024, 2048, 1, "patch-merge"),
                 (65536, 512, 512, 9, "resnet layer5 3x3"), (65536, 256, 256, 9, "resnet layer4 3x3"),
                 (262144, 64, 64, 9, "resnet layer1 3x3"), (4096, 512, 1024, 9, "aspp 3x3")]
    for M, N, K, S, note in nt_shapes:
        A = torch.randn(M, K, device=dev).to(dt)
        W = (torch.randn(N, S * K, device=dev) / (S * K) ** 0.5).to(dt)
        out = torch.empty(M, N, device=dev, dtype=dt)
        rows = None
        if S > 1 or "gather" in note:
            rows = torch.randint(0, M, (S, M), device=dev, dtype=torch.int32)
        for fl, tag in ((hip.GF_NOBIG, "8w"), (hip.GF_HALF, "half-pp"), (hip.GF_BIG, "big-pp"), (0, "auto")):
            t = timeit(lambda: hip.gemm_nt(A, W, out, M=M, a_rows=rows, S=S, flags=fl))
            print(f"{'nt-' + tag:8s} {M:7d} {N:6d} {K:6d} {S:2d} {t * 1e3:9.1f} {2.0 * M * N * K * S / t / 1e9:8.1f}  {note}")
    tn_shapes = [(32768, 1536, 512, "dWqkv s1"), (32768, 2048, 512, "dWfc1 s1"), (32768, 512, 2048, "dWfc2 s1"),
                 (32768, 512, 512, "dWproj s1"), (8192, 4096, 1024, "dWfc1 s2"), (8192, 1024, 4096, "dWfc2 s2"),
                 (65536, 512, 4608, "resnet layer5 wgrad (bseg)"), (262144, 64, 576, "resnet layer1 wgrad (bseg)")]
    for Mk, Ni, Nj, note in tn_shapes:
        At = torch.randn(Mk, Ni, device=dev).to(dt)
        bseg = 0
        rows = None
        if "bseg" in note:
            bseg = Nj // 9
            Bt = torch.randn(Mk, bseg, device=dev).to(dt)
            rows = torch.randint(0, Mk, (9, Mk), device=dev, dtype=torch.int32)
        else:
            Bt = torch.randn(Mk, Nj, device=dev).to(dt)
        out = torch.zeros(Ni, Nj, device=dev)
        for splits in (0, 8):
            t = timeit(lambda: hip.gemm_tn(At, Bt, out, Mk=Mk, bt_rows=rows, bseg=bseg, splits=splits))
            print(f"{'tn':8s} {Mk:7d} {Ni:6d} {Nj:6d} {splits:2d} {t * 1e3:9.1f} {2.0 * Mk * Ni * Nj / t / 1e9:8.1f}  {note} splits={splits}")

if __name__ == "__main__":
    main()
