#!/usr/bin/env python3
"""Start-time stagger of the first round of 256x256 ring workgroups (flag bit 17) on the K = 512 shapes with their production epilogues."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from stswincl_amd import hip
from bench_gemm import timeit


def main():
    dev, dt = "cuda", torch.bfloat16
    ST = 1 << 17
    cases = [("fc1 fwd gelu+gelu'", 65536, 2048, 512, hip.GF_GELU | hip.GF_C2_DGELU, "bias+c2"), ("fc2 dgrad * gelu' + colsum", 65536, 2048, 512, hip.GF_MUL_R, "r+cs"),
             ("qkv fwd", 65536, 1536, 512, 0, "bias"), ("proj + resid", 65536, 512, 512, hip.GF_RESID, "bias+r"), ("fc2 fwd + resid", 65536, 512, 2048, hip.GF_RESID, "bias+r"),
             ("plain", 65536, 2048, 512, 0, ""), ("fc1 fwd s2", 16384, 4096, 1024, hip.GF_GELU | hip.GF_C2_DGELU, "bias+c2"),
             ("fc1 fwd s1 B=8", 131072, 2048, 512, hip.GF_GELU | hip.GF_C2_DGELU, "bias+c2"), ("fc1 fwd s1 half", 32768, 2048, 512, hip.GF_GELU | hip.GF_C2_DGELU, "bias+c2"),
             ("fc1 fwd nograd", 65536, 2048, 512, hip.GF_GELU, "bias")]
    print(f"{'case':28s} {'plain us':>9s} {'stagger us':>10s}")
    for name, M, N, K, fl, opts in cases:
        A = torch.randn(M, K, device=dev).to(dt)
        W = (torch.randn(N, K, device=dev) / K ** 0.5).to(dt)
        out = torch.empty(M, N, device=dev, dtype=dt)
        b = torch.randn(N, device=dev) if "bias" in opts else None
        out2 = torch.empty(M, N, device=dev, dtype=dt) if "c2" in opts else None
        R = torch.randn(M, N, device=dev).to(dt) if "r" in opts.split("+") else None
        cs = torch.zeros(N, device=dev) if "cs" in opts else None
        ts = []
        for extra in (0, ST):
            ts.append(timeit(lambda: hip.gemm_nt(A, W, out, M=M, bias=b, out2=out2, resid=R, colsum_out=cs, flags=fl | extra)) * 1e3)
        print(f"{name:28s} {ts[0]:9.1f} {ts[1]:10.1f}", flush=True)


if __name__ == "__main__":
    main()
