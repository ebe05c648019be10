#!/usr/bin/env python3
"""gemm_tn (weight-gradient form, C[Ni][Nj] = At^T Bt, fp32 output) against torch.matmul on the step's shapes.  Yardstick only."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from stswincl_amd import hip
from bench_gemm import timeit


def main():
    dev, dt = "cuda", torch.bfloat16
    shapes = [(65536, 512, 2048), (65536, 2048, 512), (16384, 4096, 1024), (16384, 1024, 4096), (65536, 1536, 512), (65536, 512, 512),
              (16384, 1024, 1024), (262144, 64, 576)]
    print(f"{'Mk':>7s} {'Ni':>6s} {'Nj':>6s} | {'gemm_tn us':>10s} {'TF/s':>7s} | {'matmul bf16 out us':>18s} {'TF/s':>7s} | ratio")
    for Mk, Ni, Nj in shapes:
        At = torch.randn(Mk, Ni, device=dev).to(dt)
        Bt = torch.randn(Mk, Nj, device=dev).to(dt)
        out = torch.empty(Ni, Nj, device=dev)
        t1 = timeit(lambda: hip.gemm_tn(At, Bt, out, Mk=Mk, overwrite=True)) * 1e3
        o2 = torch.empty(Ni, Nj, device=dev, dtype=dt)
        AtT = At.t()
        t2 = timeit(lambda: torch.matmul(AtT, Bt, out=o2)) * 1e3
        fl = 2.0 * Mk * Ni * Nj
        print(f"{Mk:7d} {Ni:6d} {Nj:6d} | {t1:10.1f} {fl / t1 / 1e6:7.0f} | {t2:18.1f} {fl / t2 / 1e6:7.0f} | {t2 / t1:.2f}", flush=True)


if __name__ == "__main__":
    main()
