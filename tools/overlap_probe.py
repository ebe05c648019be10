#!/usr/bin/env python3
"""Does a dgrad gemm_nt overlap with the wgrad gemm_tn of the same layer when the two run on separate HIP streams?
Pairs from the training step; sequential (one stream) vs forked (nt on the current stream, tn on a side stream, joined)."""
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
    dev, dt = "cuda", torch.bfloat16
    side = torch.cuda.Stream()
    pairs = [  # (M, N, K, S, nt flags, resid) , (Mk, Ni, Nj, bseg)
        ("fc2 dgrad*gelu' + dWfc2", (65536, 2048, 512, 1, hip.GF_MUL_DGELU, True), (65536, 512, 2048, 0)),
        ("fc1 dgrad + dWfc1", (65536, 512, 2048, 1, 0, False), (65536, 2048, 512, 0)),
        ("qkv dgrad + dWqkv", (65536, 512, 1536, 1, 0, False), (65536, 1536, 512, 0)),
        ("layer5 conv dgrad + wgrad", (65536, 512, 512, 9, 0, False), (65536, 512, 4608, 512)),
        ("layer1 conv dgrad + wgrad", (262144, 64, 64, 9, 0, False), (262144, 64, 576, 64)),
        ("s2 fc2 dgrad*gelu' + dWfc2", (16384, 4096, 1024, 1, hip.GF_MUL_DGELU, True), (16384, 1024, 4096, 0)),
    ]
    print(f"{'pair':34s} {'nt us':>8s} {'tn us':>8s} {'seq us':>8s} {'fork us':>8s}  fork/seq")
    for name, (M, N, K, S, fl, res), (Mk, Ni, Nj, bseg) in pairs:
        A = torch.randn(M, K, device=dev).to(dt)
        W = (torch.randn(N, S * K, device=dev) / (S * K) ** 0.5).to(dt)
        out = torch.empty(M, N, device=dev, dtype=dt)
        R = torch.randn(M, N, device=dev).to(dt) if res else None
        rows = hip.conv3x3_rowmap(16, int((M // 16) ** 0.5), int((M // 16) ** 0.5), 1) if S > 1 else None
        At = torch.randn(Mk, Ni, device=dev).to(dt)
        Bt = torch.randn(Mk, bseg or Nj, device=dev).to(dt)
        dW = torch.zeros(Ni, Nj, device=dev)
        nt = lambda: hip.gemm_nt(A, W, out, M=M, a_rows=rows, S=S, resid=R, flags=fl)          # noqa: E731
        tn = lambda: hip.gemm_tn(At, Bt, dW, Mk=Mk, bt_rows=rows if bseg else None, bseg=bseg)  # noqa: E731

        def seq():
            nt(); tn()

        def fork():
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                tn()
            nt()
            torch.cuda.current_stream().wait_stream(side)

        t_nt, t_tn, t_seq, t_fork = timeit(nt), timeit(tn), timeit(seq), timeit(fork)
        print(f"{name:34s} {t_nt:8.1f} {t_tn:8.1f} {t_seq:8.1f} {t_fork:8.1f}  {t_fork / t_seq:.3f}", flush=True)


if __name__ == "__main__":
    main()
