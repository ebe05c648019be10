#!/usr/bin/env python3
"""The shader clock a 256x256 ring GEMM really runs its main loop at: core-clock cycles (s_memtime) between the loop's first and last
stage over the 100 MHz wall stamps of the same interval (debug flag 1 << 19), on random and on all-zero operands, for the 8-wave kernel
and (tuning builds: W4_CHECK=w4 / rs / m32) a variant.  MFMA-bound time per 32-deep stage = 1024 cycles / that clock."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from stswincl_amd import hip
mode = os.environ.get("W4_CHECK", "")
extra = {"w4": hip.GF_W4R, "rs": hip.GF_W4R | hip.GF_M32PP, "m32": hip.GF_M32PP, "rot": hip.GF_ROT, "swp": hip.GF_ROT | hip.GF_W4R}.get(mode, 0)
for M, N, K in ((4096, 4096, 4096), (65536, 512, 2048), (65536, 2048, 512)):
    for zero in (False, True):
        A = torch.randn(M, K, device="cuda").bfloat16()
        W = (torch.randn(N, K, device="cuda") / K ** 0.5).bfloat16()
        if zero:
            A.zero_(); W.zero_()
        out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
        nblk = (M // 256) * (N // 256)
        for _ in range(3):
            ts = torch.zeros(nblk * 16, dtype=torch.int64, device="cuda")
            hip.gemm_nt(A, W, out, M=M, flags=hip.GF_BIG | extra | (1 << 19), colsum_out=ts.view(torch.float32))
        torch.cuda.synchronize()
        t = ts.view(nblk, 16).cpu().double()
        wall_us = (t[:, 3] - t[:, 2]) / 100.0
        cyc = t[:, 7]
        ok = (wall_us > 0) & (cyc > 0)
        mhz = (cyc[ok] / wall_us[ok])
        span = float((t[:, 6].max() - t[:, 0].min()) / 100.0)
        print(f"{mode or '8-wave':7s} M={M:6d} N={N:5d} K={K:5d} {'zeros ' if zero else 'random'}: main loop {float(wall_us[ok].mean()):7.2f} us = "
              f"{float(wall_us[ok].mean()) / (K / 32) * 1e3:6.1f} ns per stage at {float(mhz.mean()):6.0f} MHz (p10 {float(mhz.quantile(0.1)):5.0f}, p90 {float(mhz.quantile(0.9)):5.0f}) "
              f"-> MFMA-bound {1024 / float(mhz.mean()) * 1e3:5.0f} ns per stage; kernel span {span:7.1f} us", flush=True)
