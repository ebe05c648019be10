#!/usr/bin/env python3
"""Where does gemm_nt stand against the vendor library on the step's plain GEMM shapes?  torch.matmul (hipBLASLt / rocBLAS) vs
hip.gemm_nt, bf16, no epilogue, HIP events.  A yardstick only: the library is not on the product path."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from stswincl_amd import hip
from bench_gemm import timeit


def main():
    dev, dt = "cuda", torch.bfloat16
    shapes = [(65536, 512, 2048), (65536, 2048, 512), (65536, 1536, 512), (65536, 512, 512), (16384, 1024, 4096), (16384, 4096, 1024),
              (65536, 512, 4608), (8192, 8192, 8192), (4096, 4096, 4096)]
    alt = int(os.environ.get("ALT_FLAGS", "0"))
    print(f"{'M':>7s} {'N':>6s} {'K':>6s} | {'gemm_nt us':>10s} {'TF/s':>7s} | {'matmul us':>10s} {'TF/s':>7s} | ratio" + ("  | alt us  TF/s" if alt else ""))
    for M, N, K in shapes:
        A = torch.randn(M, K, device=dev).to(dt)
        W = (torch.randn(N, K, device=dev) / K ** 0.5).to(dt)
        if os.environ.get("ZERO") == "1":          # all-zero operands: the same instruction stream at a fraction of the switching power
            A.zero_(); W.zero_()
        out = torch.empty(M, N, device=dev, dtype=dt)
        t1 = timeit(lambda: hip.gemm_nt(A, W, out, M=M)) * 1e3
        Wt = W.t()
        t2 = timeit(lambda: torch.matmul(A, Wt, out=out)) * 1e3
        fl = 2.0 * M * N * K
        extra = ""
        if alt:
            ref = torch.empty_like(out)
            hip.gemm_nt(A, W, ref, M=M)
            ta = timeit(lambda: hip.gemm_nt(A, W, out, M=M, flags=alt)) * 1e3
            assert torch.equal(out, ref)
            extra = f"  | {ta:7.1f} {fl / ta / 1e6:6.0f}"
        print(f"{M:7d} {N:6d} {K:6d} | {t1:10.1f} {fl / t1 / 1e6:7.0f} | {t2:10.1f} {fl / t2 / 1e6:7.0f} | {t2 / t1:.2f}" + extra, flush=True)


if __name__ == "__main__":
    main()
